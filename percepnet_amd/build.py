"""Build libpercepnet_hip.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box).

    python -m percepnet_amd.build [--force]

Flags that matter for parity (DESIGN.md "Numerics"):
  -ffp-contract=off   the DSP and STRICT kernels must round every multiply and add separately,
                      like the reference's x86-64 SSE2 build; MFMA is an explicit builtin and is
                      unaffected
  (defaults kept)     correctly rounded fp32 divide/sqrt, fp32 denormals preserved
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpercepnet_hip.so")
RUN = os.path.join(LIBDIR, "percepnet_run")
RELINKED = os.path.join(LIBDIR, "percepNet_run_relinked")    # reference src/main.cpp, untouched, linked against LIB
REFERENCE_SRC = os.environ.get("PERCEPNET_REFERENCE_SRC", "/root/reference/src")
SOURCES = ["pn_tables.cpp", "pn_dsp_fe.hip", "pn_dsp_fe_g2.hip", "pn_dsp_fe_split_s.hip", "pn_dsp_fe_split_p.hip", "pn_dsp.hip", "pn_nn.hip", "pn_nn_small.hip", "pn_nn_f16.hip", "pn_targets.hip", "pn_context.cpp",
           "pn_featgen.cpp", "rnnoise_compat.cpp"]
# percepnet_run.cpp / percepnet_featgen.cpp (the CLIs) are linked separately against the library
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


# The kernels carry two workarounds for codegen hazards of this exact toolchain; a different
# hipcc is allowed but announced, and the context-creation self-test (pn_ctx_create) is what guards the results.
EXPECTED_HIP = "7.2"


def toolchain_info(hipcc):
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError as e:
        return f"hipcc --version failed: {e}"
    ver = next((l.split(":", 1)[1].strip() for l in out.splitlines() if l.startswith("HIP version")), "?")
    if not ver.startswith(EXPECTED_HIP):
        print(f"[percepnet_amd.build] WARNING: hipcc reports HIP {ver}; the kernels were validated with HIP {EXPECTED_HIP}.x "
              f"(MFMA-drain and loop back-edge hazards, DESIGN.md): run the GPU parity tests before trusting this build",
              file=sys.stderr, flush=True)
    return out.strip()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines, verbose=False, only=None):
    """Tuning aid: build lib/variants/<name>/libpercepnet_hip.so with extra -D flags
    (select it at run time with PERCEPNET_LIB=<path>).  only: sources to recompile with the flags;
    the other objects are taken from the default build."""
    vdir = os.path.join(LIBDIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o"))
            continue
        o = os.path.join(vdir, src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + FLAGS + list(defines) + (["-x", "hip"] if src.endswith(".cpp") else []) + \
              ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(o)
    lib = os.path.join(vdir, "libpercepnet_hip.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "..", "include", "percepnet_hip.h"),
        os.path.join(HERE, "..", "include", "percepnet_nnet_data.h")]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o")
        if force or _stale(o, deps):
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    for name in ("percepnet_run", "percepnet_featgen"):
        cli, exe = os.path.join(CSRC, name + ".cpp"), os.path.join(LIBDIR, name)
        if os.path.exists(cli) and (force or _stale(exe, [cli, LIB])):
            cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", cli, "-o", exe, "-L" + LIBDIR,
                   "-lpercepnet_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    # INTEGRATION.md level 1, built for real: the reference's own UNTOUCHED main.cpp (the percepNet_run CLI) compiled
    # where it lies and linked against this library instead of the reference's objects.  Only possible where
    # /root/reference exists (this container); the binary travels to the GPU box like the .so (tests/test_gpu_parity.py).
    ref_main = os.path.join(REFERENCE_SRC, "main.cpp")
    if os.path.exists(ref_main) and (force or _stale(RELINKED, [ref_main, LIB])):
        cmd = ["g++", "-std=c++11", "-O3", "-w", "-I" + REFERENCE_SRC, ref_main, "-o", RELINKED, "-L" + LIBDIR,
               "-lpercepnet_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(os.path.join(LIBDIR, "BUILD_INFO.txt"), "w") as f:
        f.write(toolchain_info(hipcc) + "\nflags: " + " ".join(FLAGS) + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
