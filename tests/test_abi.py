"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/percepnet_hip.h declares, plus the reference's Itanium-mangled rnnoise_* names
(reference src/rnnoise.h:49-68 compiled as C++).  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from percepnet_amd import api, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return ctypes.CDLL(api.LIB_PATH)


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "percepnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:pn_|rnnoise_)\w+)\s*\(", hdr))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/percepnet_hip.h but not exported"


def test_reference_mangled_symbols_exported(lib):
    for n in ["_Z16rnnoise_get_sizev", "_Z12rnnoise_initP12DenoiseStateP8RNNModel",
              "_Z14rnnoise_createP8RNNModel", "_Z15rnnoise_destroyP12DenoiseState",
              "_Z21rnnoise_process_frameP12DenoiseStatePfPKfP8_IO_FILE",
              "_Z23rnnoise_model_from_fileP8_IO_FILE", "_Z18rnnoise_model_freeP8RNNModel",
              "_Z11compute_rnnP8RNNStatePfS1_PKf",       # void compute_rnn(RNNState*, float*, float*, const float*), rnnoise.h:68
              "_Z5trainiPPc"]:          # int train(int, char**), rnnoise.h:66
        assert hasattr(lib, n), n


def test_model_parsing_and_error_paths_without_gpu(lib, blob):
    L = api.load_library()
    assert L.pn_model_from_blob(b"nope", 4) is None
    assert b"PNW1" in L.pn_last_error()
    m = api.Model(blob)
    assert m.h
    bad = bytearray(blob); bad[8] = 7  # first layer kind -> invalid
    assert L.pn_model_from_blob(bytes(bad), len(bad)) is None
    import torch
    if not torch.cuda.is_available():
        # the product path must fail loudly without a GPU — never fall back to a CPU path
        with pytest.raises(api.PercepNetError):
            api.Context(m, 4)
        with pytest.raises(api.PercepNetError):
            api.FeatGen(4)
    m.close()


def test_model_digest_is_sha256_of_the_content(lib, blob):
    """The shared-weights cache (pn_context.cpp) is keyed by SHA-256 of the model content — a strong digest instead of round 5's two
    64-bit hashes + retained host copy.  Known answers for the hash itself, and the model digest recomputed with hashlib from the
    PNW1 container: array bytes in layer order, then (activation, reset_after) of every layer."""
    import hashlib
    import struct
    from percepnet_amd import weights
    lib.pn_debug_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    lib.pn_debug_sha256.restype = None
    for msg in (b"", b"abc", b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", b"a" * 1000003, bytes(range(256)) * 257):
        out = ctypes.create_string_buffer(32)
        lib.pn_debug_sha256(msg, len(msg), out)
        assert out.raw == hashlib.sha256(msg).digest(), len(msg)
    L = api.load_library()
    L.pn_model_digest.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.pn_model_digest.restype = None
    m = api.Model(blob)
    out = ctypes.create_string_buffer(32)
    L.pn_model_digest(m.h, out)
    h = hashlib.sha256()
    off, tail = 8, b""
    for _ in range(struct.unpack_from("<I", blob, 4)[0]):
        kind, nin, nn, ks, act, ra = struct.unpack_from("<6I", blob, off); off += 24
        n = (6 * nn if kind == 2 else nn) + nin * ks * nn * (3 if kind == 2 else 1) + (nn * 3 * nn if kind == 2 else 0)
        h.update(blob[off:off + 4 * n]); off += 4 * n
        tail += struct.pack("<2i", act, ra)
    h.update(tail)
    assert off == len(blob) and out.raw == h.digest()
    # one flipped weight bit, or one changed activation, is another model
    for patch_at, val in ((len(blob) - 5, blob[len(blob) - 5] ^ 1), (8 + 16, (blob[8 + 16] + 1) % 4)):
        b2 = bytearray(blob); b2[patch_at] = val
        m2 = api.Model(bytes(b2)); o2 = ctypes.create_string_buffer(32)
        L.pn_model_digest(m2.h, o2); m2.close()
        assert o2.raw != out.raw
    m.close()


def test_header_is_plain_c_and_a_c_caller_links(lib, blob, tmp_path):
    """gcc -std=c99 compiles a caller against include/percepnet_hip.h and links libpercepnet_hip.so; without a GPU it
    must report the missing device through pn_last_error() and exit cleanly (never fall back to a CPU path)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(api.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-L" + libdir,
                           "-lpercepnet_hip", "-Wl,-rpath," + libdir])
    mp = tmp_path / "model.pnw"
    mp.write_bytes(blob)
    r = subprocess.run([str(exe), str(mp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "percepnet_hip" in r.stdout
    import torch
    if not torch.cuda.is_available():
        assert "no context" in r.stdout and "HIP" in r.stdout
    else:
        assert "process rc=0" in r.stdout


def test_relinked_reference_cli_is_never_silently_inert(lib, blob, tmp_path):
    """The reference's untouched main.cpp linked against this library (lib/percepNet_run_relinked): when no model or no
    usable GPU is there, rnnoise_init leaves the state inert — the reference has no error path, main.cpp ignores return
    values — but says so on stderr instead of producing a silent all-zero file without a word."""
    import subprocess
    import numpy as np
    import torch
    if not os.path.exists(build.RELINKED):
        pytest.skip("needs /root/reference at build time")
    (tmp_path / "a.pcm").write_bytes((np.arange(480 * 4) % 1000).astype(np.int16).tobytes())
    env = {k: v for k, v in os.environ.items() if k != "PERCEPNET_MODEL"}
    r = subprocess.run([build.RELINKED, "a.pcm", "o.pcm"], cwd=tmp_path, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "INERT" in r.stderr and "no model" in r.stderr
    if not torch.cuda.is_available():
        (tmp_path / "m.pnw").write_bytes(blob)
        r = subprocess.run([build.RELINKED, "a.pcm", "o.pcm"], cwd=tmp_path, capture_output=True, text=True, timeout=120,
                           env=dict(env, PERCEPNET_MODEL=str(tmp_path / "m.pnw")))
        assert r.returncode == 0 and "INERT" in r.stderr and "no CPU fallback" in r.stderr
        assert not np.fromfile(tmp_path / "o.pcm", np.int16).any()


def test_launch_geometry_refusals_without_gpu(lib):
    """Round-4 verdict item 8, CPU side: the network launchers' geometry predicates (csrc/pn_launch_check.h, reached
    through pn_debug_check_launch) accept every layer of the PercepNet topology and refuse, with pn_last_error() set, what
    their software pipelines cannot run — the GPU test (test_a_refused_launch_fails_the_frame) shows that a refusal
    fails pn_process_*."""
    import ctypes
    lib.pn_debug_check_launch.argtypes = [ctypes.c_int] * 4
    lib.pn_last_error.restype = ctypes.c_char_p
    # (kind, panels, width, n_out) of the layers as pn_context.cpp launches them
    topology = [(0, 1, 128, 128), (0, 5, 128, 512), (0, 3, 512, 512), (0, 5, 512, 34), (0, 1, 128, 34),
                (1, 5, 128, 512), (1, 3, 512, 512), (1, 5, 512, 34), (2, 1, 512, 512), (2, 2, 512, 128),
                (3, 5, 512, 34), (3, 1, 128, 34)]
    for g in topology:
        assert lib.pn_debug_check_launch(*g) == 0, (g, lib.pn_last_error())
    refused = [((0, 3, 32, 34), b"even"), ((0, 1, 32, 34), b"even"), ((0, 6, 128, 34), b"panels"), ((1, 3, 500, 34), b"whole"),
               ((2, 1, 512, 96), b"neurons"), ((2, 1, 32, 512), b"even"), ((3, 1, 48, 34), b"power-of-two"), ((3, 3, 16, 34), b"multiple"),
               ((9, 1, 128, 34), b"unknown kind")]
    for g, word in refused:
        assert lib.pn_debug_check_launch(*g) == -1, g
        assert word in lib.pn_last_error(), (g, lib.pn_last_error())
