"""Build lib/variants/<name>/libpercepnet_hip.so with tools/experiments/pitch_x.hip in place of csrc/pn_dsp_fe_split_p.hip (the other
objects come from the default build), so that pitch-kernel experiments do not touch the kernels the committed profiles are stamped for.
    cp percepnet_amd/csrc/pn_dsp_fe_split_p.hip tools/experiments/pitch_x.hip      (the working copy is not tracked)
    python tools/experiments/build_x.py <name> [-DFLAG ...]
then PERCEPNET_LIB=percepnet_amd/lib/variants/<name>/libpercepnet_hip.so python tools/fp_variants.py / tools/fe_ab.py on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from percepnet_amd import build as b
name, defs = sys.argv[1], sys.argv[2:]
vdir = os.path.join(b.LIBDIR, "variants", name); os.makedirs(vdir, exist_ok=True)
o = os.path.join(vdir, "pn_dsp_fe_split_p.o")
cmd = [b._hipcc()] + b.FLAGS + defs + ["-I" + b.CSRC, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(ROOT, "tools", "experiments", "pitch_x.hip"), "-o", o]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.stderr.write(r.stderr); sys.exit(1)
print(" ".join(l.split("remark:")[1].strip() for l in r.stderr.splitlines() if "pn_fe_pitch" in l or any(k in l for k in ("    VGPRs:", "ScratchSize", "Occupancy"))))
objs = [o if s == "pn_dsp_fe_split_p.hip" else os.path.join(b.LIBDIR, s.rsplit(".", 1)[0] + ".o") for s in b.SOURCES]
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + b.EXPORT_MAP, "-o", os.path.join(vdir, "libpercepnet_hip.so")] + objs)
print(os.path.join(vdir, "libpercepnet_hip.so"))
