"""CPU-only hygiene checks of the drop-in library (round-3 verdict item 6): the export list is exactly the C-ABI plus the
reference's nine mangled names; the host-side parsers, packers and table builder survive hostile input under
-fsanitize=address,undefined; rnnoise_model_from_file / rnnoise_model_free (declared, never defined, by the reference:
src/rnnoise.h:62-64) reject what they must."""
import ctypes
import os
import re
import shutil
import struct
import subprocess

import pytest

from percepnet_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANGLED = {"_Z16rnnoise_get_sizev", "_Z12rnnoise_initP12DenoiseStateP8RNNModel", "_Z14rnnoise_createP8RNNModel",
           "_Z15rnnoise_destroyP12DenoiseState", "_Z21rnnoise_process_frameP12DenoiseStatePfPKfP8_IO_FILE",
           "_Z23rnnoise_model_from_fileP8_IO_FILE", "_Z18rnnoise_model_freeP8RNNModel", "_Z11compute_rnnP8RNNStatePfS1_PKf",
           "_Z5trainiPPc"}


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return ctypes.CDLL(api.LIB_PATH)


def test_nothing_but_the_c_abi_and_the_reference_names_is_exported(lib):
    """-fvisibility=hidden + the ld version script csrc/libpercepnet_hip.map: no launcher, packer, kernel stub or libstdc++
    instantiation leaks out of a library that is meant to be dropped into somebody else's process."""
    if not shutil.which("nm"):
        pytest.skip("nm not available")
    out = subprocess.run(["nm", "-D", "--defined-only", api.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    hdr = open(os.path.join(ROOT, "include", "percepnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:pn_|rnnoise_)\w+)\s*\(", hdr))
    assert exported == declared | MANGLED, (sorted(exported - declared - MANGLED), sorted((declared | MANGLED) - exported))


def test_host_pieces_under_address_and_undefined_behaviour_sanitizers(blob, tmp_path):
    """tests/c/host_sanitize.cpp = pn_model.cpp + pn_pack.cpp + pn_tables.cpp + pn_cli_util.h built WITHOUT HIP by plain g++
    with -fsanitize=address,undefined, then fed the valid container, every truncation around the headers, absurd
    dimensions in every header field, random byte flips, wrong RNNModel geometries, and exactly-sized packer outputs."""
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = tmp_path / "host_sanitize"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-DPN_NO_HIP", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(ROOT, "tests", "c", "host_sanitize.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if r.returncode and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    mp = tmp_path / "m.pnw"
    mp.write_bytes(blob)
    r = subprocess.run([str(exe), str(mp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


def _layer_offsets(blob):
    off, offs = 8, []
    for _ in range(10):
        kind, nin, nn, ks, act, ra = struct.unpack_from("<6I", blob, off)
        offs.append(off)
        nb = 6 * nn if kind == 2 else nn
        nw = nin * ks * nn * (3 if kind == 2 else 1)
        nr = nn * 3 * nn if kind == 2 else 0
        off += 24 + 4 * (nb + nw + nr)
    assert off == len(blob)
    return offs


def test_pnw1_parser_refuses_hostile_containers(lib, blob):
    L = api.load_library()
    offs = _layer_offsets(blob)
    cases = {"truncated header": blob[:offs[3] + 10], "truncated arrays": blob[:offs[4] - 4], "trailing byte": blob + b"\0",
             "empty": b"", "magic only": b"PNW1"}
    for fld, val in ((1, 0xFFFFFFFF), (2, 0x80000000), (3, 0x10000), (0, 2), (1, 0)):          # huge / wrong dimensions, wrong kind
        b = bytearray(blob); struct.pack_into("<I", b, offs[1] + 4 * fld, val); cases[f"conv1 field {fld} = {val:#x}"] = bytes(b)
    b = bytearray(blob); struct.pack_into("<I", b, 4, 0xFFFFFFFF); cases["4 G layers"] = bytes(b)
    for name, c in cases.items():
        assert L.pn_model_from_blob(c, len(c)) is None, name
        assert L.pn_last_error(), name
    assert b"topology" in (L.pn_model_from_blob(cases["conv1 field 2 = 0x80000000"], len(blob)) or L.pn_last_error())


def test_model_from_file_and_free_are_safe_on_foreign_input(lib, blob, tmp_path):
    """rnnoise_model_free must only free what rnnoise_model_from_file returned: a foreign RNNModel (the link-time
    percepnet_model_orig, a caller's own struct) is left alone; a double free is a no-op."""
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p; libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    lib.rnnoise_model_from_file_c.restype = ctypes.c_void_p; lib.rnnoise_model_from_file_c.argtypes = [ctypes.c_void_p]
    lib.rnnoise_model_free_c.argtypes = [ctypes.c_void_p]
    good, bad, huge = tmp_path / "good.pnw", tmp_path / "bad.pnw", tmp_path / "huge.pnw"
    good.write_bytes(blob); bad.write_bytes(blob[:100000])
    with open(huge, "wb") as f:                                   # sparse 80 MB of zeros behind a valid magic: refused unread
        f.write(b"PNW1"); f.seek(80 << 20); f.write(b"\0")
    def load(p):
        fp = libc.fopen(str(p).encode(), b"rb"); assert fp
        m = lib.rnnoise_model_from_file_c(fp); libc.fclose(fp)
        return m
    assert load(bad) is None and load(huge) is None
    assert lib.rnnoise_model_from_file_c(None) is None
    m = load(good)
    assert m
    # a struct the library never handed out (here: 80 zero bytes on our side) and NULL are ignored
    foreign = (ctypes.c_char * 80)()
    lib.rnnoise_model_free_c(ctypes.addressof(foreign)); lib.rnnoise_model_free_c(None)
    lib.rnnoise_model_free_c(m); lib.rnnoise_model_free_c(m)     # second call: no longer registered, nothing happens


def test_disassembly_gate_of_the_hand_scheduled_pitch_kernel():
    """Advisor (round 5, medium): the pitch kernel's DPP adds and the LDS reads of its packed coarse loop come from inline assembly, out
    of sight of the compiler's hazard recogniser and s_waitcnt pass.  build.check_dpp_and_waitcnt checks the ISA of every build: no VALU
    write within two wait states of a DPP read of that register as src0, no scalar load inside the packed coarse loop.  Here: the
    checker finds planted violations in a synthetic listing, and the object of THIS build is clean (build() fails otherwise)."""
    from percepnet_amd import build
    fake = """
0000000000001000 <pn_fe_pitch_kernel>:
	v_mul_f32_e32 v5, v1, v2                                // 000000001000: 0A0A0501
	v_add_f32_dpp v7, v5, v7 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1// 000000001004: 020E0EFA FF095005
	v_mul_f32_e32 v9, v1, v2                                // 00000000100C: 0A120501
	s_nop 1                                                    // 000000001010: BF800001
	v_add_f32_dpp v7, v9, v7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1// 000000001014
	v_mul_f32_e32 v12, v1, v2                               // one instruction between = 1 wait state: still too close
	v_mov_b32_e32 v20, v21
	v_subrev_f32_dpp v7, v12, v7 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_pk_mul_f32 v[36:37], v[34:35], v[42:43] op_sel:[1,0]
	s_load_dwordx2 s[4:5], s[0:1], 0x0
	v_pk_mul_f32 v[36:37], v[34:35], v[42:43] op_sel:[1,0]
	v_pk_add_f32 v[10:11], v[36:37], v[42:43]
	v_mov_b32_dpp v3, v11 row_ror:1 row_mask:0xf bank_mask:0xf
	ds_read_b32 v30, v31
	v_add_f32_dpp v7, v30, v7 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1
"""
    bad = build.check_dpp_and_waitcnt(fake)
    assert len(bad) == 4, bad
    assert "v5" in bad[0] and "0 wait state" in bad[0]
    assert "v12" in bad[1] and "1 wait state" in bad[1]
    assert "v[10:11]" in bad[2]
    assert "scalar load" in bad[3] and "s_load_dwordx2" in bad[3]
    build.build(verbose=False)
    obj = os.path.join(build.LIBDIR, "pn_dsp_fe_split_p.o")
    isa = build.disassemble_device_code(obj)
    assert isa.count("_dpp") > 500 and "v_pk_mul_f32" in isa          # the hand-scheduled sequences are really there
    assert build.check_dpp_and_waitcnt(isa) == []
