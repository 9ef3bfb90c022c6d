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
SOURCES = ["pn_tables.cpp", "pn_dsp_fe.hip", "pn_dsp.hip", "pn_nn.hip", "pn_nn_f16.hip", "pn_targets.hip", "pn_context.cpp",
           "pn_featgen.cpp", "rnnoise_compat.cpp"]
# percepnet_run.cpp / percepnet_featgen.cpp (the CLIs) are linked separately against the library
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


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
                   "-lpercepnet_hip", "-Wl,-rpath,$ORIGIN"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
