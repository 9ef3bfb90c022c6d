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
EXPORT_MAP = os.path.join(CSRC, "libpercepnet_hip.map")    # ld version script: the export list (everything else is local)
RELINKED = os.path.join(LIBDIR, "percepNet_run_relinked")    # reference src/main.cpp, untouched, linked against LIB
REFERENCE_SRC = os.environ.get("PERCEPNET_REFERENCE_SRC", "/root/reference/src")
SOURCES = ["pn_tables.cpp", "pn_model.cpp", "pn_pack.cpp", "pn_dsp_fe.hip", "pn_dsp_fe_g2.hip", "pn_dsp_fe_split_s.hip", "pn_dsp_fe_split_p.hip", "pn_dsp.hip", "pn_nn.hip", "pn_nn_small.hip", "pn_nn_x3.hip", "pn_nn_d.hip", "pn_nn_n48.hip", "pn_targets.hip", "pn_state.hip", "pn_active.hip", "pn_context.cpp",
           "pn_featgen.cpp", "rnnoise_compat.cpp"]
# percepnet_run.cpp / percepnet_featgen.cpp (the CLIs) are linked separately against the library
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


# per-source extra flags.  pn_nn_x3.hip: no SLP vectorisation — packed f32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32)
# issued beside a wave that keeps the SIMD's matrix pipe busy are 4x slower than beside an idle one (NOTES_history.md, round-4 DESIGN 4.2f;
# tools/probes/mfma_valu_pair_probe.hip), and the gating epilogues of these kernels run exactly there
# -pragma-unroll-threshold: the paired-phase GRU kernel's epilogue phase is ONE fully unrolled loop over its 32 / 36 barrier steps
# (every step then has compile-time register sets, ring slots and tile numbers); the default limit of `#pragma unroll` (16 K
# IR instructions, counted before the per-step branches fold) silently leaves it rolled — with every array in scratch
EXTRA_FLAGS = {"pn_nn_x3.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"],
               "pn_nn_d.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


# The kernels were validated with this toolchain (NOTES_history.md: the MFMA wait-state rule is met by the compiler's own
# padding, with zero margin); a different hipcc is allowed but announced, and the create-time self-tests of the network
# and DSP kernels (pn_ctx_create) are what guard the results.
EXPECTED_HIP = "7.2"

# Register hygiene gate for the PRODUCTION DSP kernels: the single-launch front end of rounds 1-2 ran at 256 VGPRs + 219
# AGPR spill copies, the regime in which a register-allocation-dependent corruption once appeared (NOTES_history.md).  The
# phase-split kernels and the back end must stay free of AGPR spill copies and (almost) free of scratch; the build fails
# otherwise.  kernel-name prefix -> (max AGPRs, max scratch bytes per lane)
RESOURCE_LIMITS = {"pn_fe_spec_in_kernel": (0, 0), "pn_fe_spec_out_kernel": (0, 40), "pn_fe_pitch_kernel": (0, 48),
                   "pn_backend_kernel": (0, 16),
                   # shadow-operand network kernels: everything in registers (a staging array once went to scratch: +30 % time)
                   "pn_gru_x3_kernel": (0, 0), "pn_dense_x3_kernel": (0, 0),
                   "pn_gru_d_kernel": (0, 0)}
RESOURCE_SOURCES = ("pn_dsp_fe_split_s.hip", "pn_dsp_fe_split_p.hip", "pn_dsp.hip", "pn_nn_x3.hip", "pn_nn_d.hip")


def parse_resource_remarks(text):
    """hipcc -Rpass-analysis=kernel-resource-usage stderr -> {demangled-ish kernel name: {"VGPRs": n, "AGPRs": n, ...}}"""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            mm = re.match(r"_Z\d+(pn_[a-z0-9_]+)", name)
            cur = out.setdefault(name, {"kernel": mm.group(1) if mm else name})
            continue
        m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill|SGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def check_resources(remarks):
    """-> list of violations of RESOURCE_LIMITS (empty = clean)."""
    bad = []
    for name, r in remarks.items():
        for prefix, (max_agpr, max_scratch) in RESOURCE_LIMITS.items():
            if r["kernel"].startswith(prefix):
                if r.get("AGPRs", 0) > max_agpr or r.get("ScratchSize", 0) > max_scratch:
                    bad.append(f"{r['kernel']}: {r.get('VGPRs')} VGPRs, {r.get('AGPRs')} AGPRs, {r.get('ScratchSize')} B/lane scratch "
                               f"(limits: {max_agpr} AGPRs, {max_scratch} B) [{name}]")
    return bad


# ---- disassembly gate of the pitch kernel's hand-scheduled sequences (advisor, round 5) -----------------------------------------
# pn_dsp_fe_split_p.hip issues its DPP adds and the LDS reads of the packed coarse loop from inline assembly, so the compiler's
# hazard recogniser and s_waitcnt pass no longer see them.  Two properties the source relies on are checked in the ISA of every
# build instead:
#   (1) gfx9 hazard "VALU writes a VGPR -> a DPP instruction reads it as src0 (the operand routed through the cross-lane network):
#       2 wait states".  Fails on any VALU write of vN followed by fewer than two wait states (an instruction = 1, s_nop k = k + 1)
#       before a *_dpp instruction whose src0 is vN.
#   (2) the packed coarse loop waits for its own LDS reads with hand-counted s_waitcnt lgkmcnt(4): LDS returns in order, SMEM does
#       not — a scalar load inside that region would break the count.  Fails on any s_load / s_buffer_load between the first and
#       the last v_pk_mul_f32 of pn_fe_pitch_kernel.
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
GATED_OBJECTS = ("pn_dsp_fe_split_p",)


def disassemble_device_code(obj_path):
    """gfx950 ISA text of the device code object bundled in a hipcc -c output (llvm-objdump --offloading + -d)."""
    import shutil
    import tempfile
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump")
    if not objdump:
        raise RuntimeError("llvm-objdump not found: the disassembly gate cannot run")
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([objdump, "--offloading", local], cwd=d, capture_output=True, text=True, check=True)
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        if not cos:
            raise RuntimeError(f"no device code object in {obj_path}")
        return subprocess.run([objdump, "-d", os.path.join(d, cos[0])], capture_output=True, text=True, check=True).stdout


def _vregs(op):
    """'v78' -> {78}; 'v[66:67]' -> {66, 67}; anything else -> set()"""
    import re
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def check_dpp_and_waitcnt(disasm):
    """-> list of violations of the two hand-scheduling rules above in an llvm-objdump -d listing (empty = clean)."""
    import re
    bad = []
    func, ins = None, []                              # per function: (mnemonic, [operands], text)
    funcs = {}
    for line in disasm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            func = m.group(1); ins = funcs.setdefault(func, [])
            continue
        t = line.split("//")[0].strip()
        if func is None or not t or t.startswith("."):
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        ins.append((parts[0], ops, t))
    for func, ins in funcs.items():
        for i, (mn, ops, text) in enumerate(ins):
            if "_dpp" not in mn or len(ops) < 2:
                continue
            src0 = _vregs(ops[1].split()[0])          # v_*_dpp vdst, src0 (DPP-routed), [src1] <modifiers glued to the last operand>
            if not src0:
                continue
            states, j = 0, i - 1
            while j >= 0 and states < 2:
                pm, pops, ptext = ins[j]
                if pm.startswith("v_") and pops and (_vregs(pops[0].split()[0]) & src0) and not pm.startswith(("v_cmp", "v_nop")):
                    bad.append(f"{func}: `{ptext}` writes the DPP-routed src0 of `{text}` only {states} wait state(s) earlier (need 2)")
                    break
                states += (int(pops[0], 0) + 1) if (pm == "s_nop" and pops) else 1
                j -= 1
        if "pn_fe_pitch_kernel" in func:
            pk = [i for i, x in enumerate(ins) if x[0] == "v_pk_mul_f32"]
            if pk:
                for mn, ops, text in ins[pk[0]:pk[-1] + 1]:
                    if mn.startswith(("s_load_", "s_buffer_load")):
                        bad.append(f"{func}: scalar load `{text}` inside the packed coarse loop (its hand-counted s_waitcnt lgkmcnt(4) assumes LDS reads only)")
    return bad


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
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + list(defines) + (["-x", "hip"] if src.endswith(".cpp") else []) + \
              ["-c", os.path.join(CSRC, src), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(o)
    lib = os.path.join(vdir, "libpercepnet_hip.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORT_MAP, "-o", lib] + objs)
    return lib


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "..", "include", "percepnet_hip.h"),
        os.path.join(HERE, "..", "include", "percepnet_nnet_data.h")]
    objs = []
    resources = {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o")
        rlog = o + ".resources.txt"
        if force or _stale(o, deps) or (src in RESOURCE_SOURCES and not os.path.exists(rlog)):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
            if src in RESOURCE_SOURCES:
                cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
            if verbose:
                print(" ".join(cmd), flush=True)
            if src in RESOURCE_SOURCES:
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode:
                    sys.stderr.write(r.stderr)
                    raise subprocess.CalledProcessError(r.returncode, cmd)
                open(rlog, "w").write("\n".join(l for l in r.stderr.splitlines() if "remark:" in l) + "\n")
            else:
                subprocess.check_call(cmd)
        if src in RESOURCE_SOURCES and os.path.exists(rlog):
            resources.update(parse_resource_remarks(open(rlog).read()))
        objs.append(o)
    bad = check_resources(resources)
    if bad:
        raise RuntimeError("register hygiene gate (build.RESOURCE_LIMITS) failed:\n  " + "\n  ".join(bad))
    for base in GATED_OBJECTS:
        o = os.path.join(LIBDIR, base + ".o")
        stamp = o + ".isa_gate_ok"
        if force or _stale(stamp, [o]):
            viol = check_dpp_and_waitcnt(disassemble_device_code(o))
            if viol:
                raise RuntimeError("disassembly gate (build.check_dpp_and_waitcnt) failed — rebuild with -DPN_FP_DPP_ASM=0 -DPN_FP_COARSE_PK=0 or "
                                   "fix the sequence:\n  " + "\n  ".join(viol[:20]))
            open(stamp, "w").write("ok\n")
    if force or _stale(LIB, objs + [EXPORT_MAP]):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORT_MAP, "-o", LIB] + objs
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
        f.write("DSP kernel resources (hipcc -Rpass-analysis=kernel-resource-usage):\n")
        for name, r in sorted(resources.items(), key=lambda kv: kv[1]["kernel"]):
            f.write(f"  {r['kernel']}: VGPRs {r.get('VGPRs')} AGPRs {r.get('AGPRs')} scratch {r.get('ScratchSize')} B/lane "
                    f"LDS {r.get('LDS Size')} B/block occupancy {r.get('Occupancy')} waves/SIMD  [{name[:60]}]\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
