"""ctypes binding of libpercepnet_hip.so (include/percepnet_hip.h) — the host-side mirror of the
reference's frame-engine interface for Python callers (tests, bench).

There is deliberately no fallback: if the HIP library is missing or no GPU is usable, loading /
context creation raises.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpercepnet_hip.so")

NN_MFMA, NN_STRICT, NN_MFMA_F16, NN_MFMA_X3 = 0, 1, 2, 3
FRAME = 480

_vp = ctypes.c_void_p
_lib = None


class PercepNetError(RuntimeError):
    pass


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64; if torch is going
    # to be used in this process (tests, bench) it must be loaded BEFORE this library so that
    # both bind to the same runtime copy (loading order reversed, torch reports "No HIP GPUs").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("PERCEPNET_LIB", LIB_PATH)   # override: tuning variants (build.build_variant)
    if not os.path.exists(path):
        raise PercepNetError(
            f"{path} is missing: build it with `python -m percepnet_amd.build` (no CPU fallback exists)")
    L = ctypes.CDLL(path)
    L.pn_last_error.restype = ctypes.c_char_p
    L.pn_version.restype = ctypes.c_char_p
    L.pn_model_from_blob.restype = _vp
    L.pn_model_from_blob.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.pn_model_free.argtypes = [_vp]
    L.pn_ctx_create.restype = _vp
    L.pn_ctx_create.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]
    L.pn_ctx_destroy.argtypes = [_vp]
    L.pn_ctx_reset.argtypes = [_vp]
    L.pn_ctx_reset_streams.argtypes = [_vp, _vp, ctypes.c_int]
    L.pn_ctx_n_streams.argtypes = [_vp]
    L.pn_ctx_frames_done.restype = ctypes.c_int64
    L.pn_ctx_frames_done.argtypes = [_vp]
    L.pn_ctx_device_bytes.restype = ctypes.c_size_t
    L.pn_ctx_device_bytes.argtypes = [_vp]
    L.pn_ctx_describe.argtypes = [_vp, ctypes.c_char_p, ctypes.c_size_t]
    if hasattr(L, "pn_ctx_weight_bytes"):
        L.pn_ctx_weight_bytes.restype = ctypes.c_size_t
        L.pn_ctx_weight_bytes.argtypes = [_vp]
    for name in ("pn_process_f32", "pn_process_i16", "pn_process_host_f32", "pn_process_host_i16"):
        getattr(L, name).argtypes = [_vp, _vp, _vp, _vp]
    L.pn_process_i16_multi.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int]
    if hasattr(L, "pn_process_i16_active"):                 # (a round-4 library loaded through PERCEPNET_LIB for A/B timing has none of these)
        for name in ("pn_process_f32_active", "pn_process_i16_active", "pn_submit_host_f32_active", "pn_submit_host_i16_active"):
            if hasattr(L, name):
                getattr(L, name).argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_int]
        L.pn_debug_check_launch.argtypes = [ctypes.c_int] * 4
        L.pn_ctx_debug_inject_launch_failure.argtypes = [_vp, ctypes.c_int]
    for name in ("pn_submit_host_f32", "pn_submit_host_i16"):
        getattr(L, name).argtypes = [_vp, _vp, _vp, _vp]
    L.pn_host_wait.argtypes = [_vp]
    if hasattr(L, "pn_ctx_pipe_streams"):
        L.pn_ctx_pipe_streams.argtypes = [_vp]
        L.pn_ctx_pipe_streams.restype = ctypes.c_char_p
    if hasattr(L, "pn_host_frames_delivered"):
        L.pn_host_frames_delivered.argtypes = [_vp]
        L.pn_host_frames_delivered.restype = ctypes.c_int64
    L.pn_host_alloc.argtypes = [ctypes.c_size_t]
    L.pn_host_alloc.restype = _vp
    L.pn_host_free.argtypes = [_vp]
    L.pn_host_free.restype = None
    L.pn_ctx_synchronize.argtypes = [_vp]
    L.pn_ctx_set_postfilter.argtypes = [_vp, ctypes.c_int]
    L.pn_ctx_read_features.argtypes = [_vp, _vp, _vp]
    L.pn_ctx_read_features_dev.argtypes = [_vp, _vp, _vp]
    L.pn_ctx_compute_rnn_host.argtypes = [_vp, _vp, _vp]
    L.pn_ctx_set_rnn_state_host.argtypes = [_vp] * 8
    L.pn_ctx_get_rnn_state_host.argtypes = [_vp] * 8
    L.pn_ctx_debug_copy.restype = ctypes.c_longlong
    L.pn_ctx_debug_copy.argtypes = [_vp, ctypes.c_int, _vp, ctypes.c_longlong]
    L.pn_ctx_set_profiling.argtypes = [_vp, ctypes.c_int]
    L.pn_kernel_name.restype = ctypes.c_char_p
    L.pn_kernel_name.argtypes = [ctypes.c_int]
    L.pn_ctx_kernel_time.argtypes = [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_int64)]
    L.pn_ctx_reset_profile.argtypes = [_vp]
    L.pn_featgen_create.restype = _vp
    L.pn_featgen_create.argtypes = [ctypes.c_int, ctypes.c_int, _vp]
    L.pn_featgen_destroy.argtypes = [_vp]
    L.pn_featgen_reset.argtypes = [_vp]
    L.pn_featgen_synchronize.argtypes = [_vp]
    L.pn_featgen_device_bytes.restype = ctypes.c_size_t
    L.pn_featgen_device_bytes.argtypes = [_vp]
    L.pn_featgen_process_i16.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.pn_featgen_process_i16_files.argtypes = [_vp, _vp, _vp, ctypes.c_int, _vp, _vp]
    L.pn_featgen_process_host_i16_files.argtypes = [_vp, _vp, _vp, ctypes.c_int, _vp, _vp]
    L.pn_featgen_run_files.argtypes = [ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    _lib = L
    return L


def _err(L):
    return L.pn_last_error().decode(errors="replace")


class Model:
    def __init__(self, blob):
        self.L = load_library()
        self.h = self.L.pn_model_from_blob(blob, len(blob))
        if not self.h:
            raise PercepNetError(_err(self.L))

    def close(self):
        if self.h:
            self.L.pn_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """B independent streams advanced one 10 ms frame per call (pn_ctx)."""

    def __init__(self, model, n_streams, device=0, nn_mode=NN_MFMA, stream=None):
        self.L = model.L
        self.model = model
        self.n_streams = int(n_streams)
        self.h = self.L.pn_ctx_create(model.h, device, self.n_streams, nn_mode, stream)
        if not self.h:
            raise PercepNetError(_err(self.L))

    def close(self):
        if self.h:
            self.L.pn_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PercepNetError(_err(self.L))

    def reset(self):
        self._chk(self.L.pn_ctx_reset(self.h))

    def reset_streams(self, ids):
        """rnnoise_init for the streams `ids` only (the others keep running): include/percepnet_hip.h pn_ctx_reset_streams."""
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).ravel())
        self._chk(self.L.pn_ctx_reset_streams(self.h, a.ctypes.data, int(a.size)))

    def synchronize(self):
        self._chk(self.L.pn_ctx_synchronize(self.h))

    def set_postfilter(self, enable):
        """Optional envelope post-filter on the gains (reference post_filtering, denoise.cpp:216-250)."""
        self._chk(self.L.pn_ctx_set_postfilter(self.h, int(bool(enable))))

    def device_bytes(self):
        return self.L.pn_ctx_device_bytes(self.h)

    def weight_bytes(self):
        return self.L.pn_ctx_weight_bytes(self.h)

    def describe(self):
        """{"nn": ..., "dense": "small"|"batch", "gru": ..., "gru_rb": ..., "frontend": ...}: the kernel families in use."""
        buf = ctypes.create_string_buffer(256)
        if self.L.pn_ctx_describe(self.h, buf, len(buf)) < 0:
            raise PercepNetError("pn_ctx_describe failed")
        return dict(kv.split("=", 1) for kv in buf.value.decode().split())

    # device-pointer entry points (ints, e.g. torch.Tensor.data_ptr())
    def process_i16_dev(self, d_in, d_out, d_gr=None):
        self._chk(self.L.pn_process_i16(self.h, d_in, d_out, d_gr))

    def process_f32_dev(self, d_in, d_out, d_gr=None):
        self._chk(self.L.pn_process_f32(self.h, d_in, d_out, d_gr))

    def process_i16_active_dev(self, d_in, d_out, d_gr, ids):
        """One frame for the streams `ids` only; every other stream keeps all of its state and its output rows
        (include/percepnet_hip.h pn_process_i16_active; reference contract: src/denoise.cpp:508-547, one call per stream)."""
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).ravel())
        self._chk(self.L.pn_process_i16_active(self.h, d_in, d_out, d_gr, a.ctypes.data, int(a.size)))

    def process_f32_active_dev(self, d_in, d_out, d_gr, ids):
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).ravel())
        self._chk(self.L.pn_process_f32_active(self.h, d_in, d_out, d_gr, a.ctypes.data, int(a.size)))

    def debug_inject_launch_failure(self, enable):
        self._chk(self.L.pn_ctx_debug_inject_launch_failure(self.h, int(bool(enable))))

    # pipelined host-buffer entry points (raw host pointers; the buffers should be pinned and must outlive delivery)
    def submit_host_i16(self, h_in, h_out, h_gr=None):
        self._chk(self.L.pn_submit_host_i16(self.h, h_in, h_out, h_gr))

    def submit_host_i16_active(self, h_in, h_out, h_gr, ids):
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).ravel())
        self._chk(self.L.pn_submit_host_i16_active(self.h, h_in, h_out, h_gr, a.ctypes.data, int(a.size)))

    def pipe_streams(self):
        """'nn' / 'hl' ...: how the copy streams of the pipelined host path were obtained ('' before the first submit)"""
        return self.L.pn_ctx_pipe_streams(self.h).decode() if hasattr(self.L, "pn_ctx_pipe_streams") else ""

    def host_wait(self):
        self._chk(self.L.pn_host_wait(self.h))

    # host numpy entry points
    def process_i16(self, frame, want_gr=True):
        frame = np.ascontiguousarray(frame, dtype=np.int16).reshape(self.n_streams, FRAME)
        out = np.empty_like(frame)
        gr = np.empty((self.n_streams, 68), np.float32) if want_gr else None
        self._chk(self.L.pn_process_host_i16(self.h, frame.ctypes.data, out.ctypes.data,
                                             gr.ctypes.data if want_gr else None))
        return out, gr

    def process_f32(self, frame, want_gr=True):
        frame = np.ascontiguousarray(frame, dtype=np.float32).reshape(self.n_streams, FRAME)
        out = np.empty_like(frame)
        gr = np.empty((self.n_streams, 68), np.float32) if want_gr else None
        self._chk(self.L.pn_process_host_f32(self.h, frame.ctypes.data, out.ctypes.data,
                                             gr.ctypes.data if want_gr else None))
        return out, gr

    def run_pcm(self, pcm):
        """percepNet_run semantics (main.cpp:30-39) for a batch: pcm int16 [B, n_frames*480] ->
        (out int16 [B, (n_frames-1)*480] with the first output frame dropped, gr [B, n_frames, 68])."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16).reshape(self.n_streams, -1)
        n = pcm.shape[1] // FRAME
        out = np.zeros((self.n_streams, max(n - 1, 0) * FRAME), np.int16)
        gr = np.zeros((self.n_streams, n, 68), np.float32)
        for t in range(n):
            o, g = self.process_i16(pcm[:, t * FRAME:(t + 1) * FRAME])
            gr[:, t] = g
            if t > 0:
                out[:, (t - 1) * FRAME:t * FRAME] = o
        return out, gr

    def read_features(self):
        feat = np.empty((self.n_streams, 70), np.float32)
        sil = np.empty(self.n_streams, np.int32)
        self._chk(self.L.pn_ctx_read_features(self.h, feat.ctypes.data, sil.ctypes.data))
        return feat, sil

    def read_features_dev(self, d_feat, d_sil=None):
        """Device-pointer twin of read_features (async on the context's stream)."""
        self._chk(self.L.pn_ctx_read_features_dev(self.h, d_feat, d_sil))

    def compute_rnn(self, feat):
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(self.n_streams, 70)
        gr = np.empty((self.n_streams, 68), np.float32)
        self._chk(self.L.pn_ctx_compute_rnn_host(self.h, feat.ctypes.data, gr.ctypes.data))
        return gr

    RNN_STATE_SHAPES = (("conv1", 4 * 128), ("conv2", 2 * 512), ("gru1", 512), ("gru2", 512), ("gru3", 512),
                        ("gru_gb", 512), ("gru_rb", 128))

    def get_rnn_state(self):
        """-> {name: float32 [n_streams, n]} in the reference's RNNState layout (nnet_data.h:28-38)."""
        st = {k: np.empty((self.n_streams, n), np.float32) for k, n in self.RNN_STATE_SHAPES}
        self._chk(self.L.pn_ctx_get_rnn_state_host(self.h, *[st[k].ctypes.data for k, _ in self.RNN_STATE_SHAPES]))
        return st

    def set_rnn_state(self, st):
        arrs = [np.ascontiguousarray(st[k], dtype=np.float32).reshape(self.n_streams, n) if k in st else None
                for k, n in self.RNN_STATE_SHAPES]
        self._chk(self.L.pn_ctx_set_rnn_state_host(self.h, *[a.ctypes.data if a is not None else None for a in arrs]))

    def debug_copy(self, which, n_floats):
        """Internal device buffer `which` (see pn_ctx_debug_copy in percepnet_hip.h) -> float32[n_floats] (tests/tools)."""
        buf = np.empty(n_floats, np.float32)
        n = self.L.pn_ctx_debug_copy(self.h, which, buf.ctypes.data, buf.nbytes)
        if n < 0:
            raise PercepNetError(_err(self.L))
        return buf[:n // 4]

    def set_profiling(self, on):
        self._chk(self.L.pn_ctx_set_profiling(self.h, 1 if on else 0))

    def reset_profile(self):
        self._chk(self.L.pn_ctx_reset_profile(self.h))

    def kernel_times(self):
        """-> {family: (total_ms, launches)} from HIP events on the context's stream."""
        out = {}
        for i in range(self.L.pn_kernel_count()):
            name = self.L.pn_kernel_name(i)
            ms = ctypes.c_double()
            n = ctypes.c_int64()
            self._chk(self.L.pn_ctx_kernel_time(self.h, name, ctypes.byref(ms), ctypes.byref(n)))
            out[name.decode()] = (ms.value, n.value)
        return out


class FeatGen:
    """Batched training-feature generator (pn_featgen): the reference's `percepNet <speech> <noisy>
    <count> <output>` binary (train(), denoise.cpp:603-787) for n_pairs pairs in lock-step."""

    def __init__(self, n_pairs, device=0, stream=None):
        self.L = load_library()
        self.n_pairs = int(n_pairs)
        self.h = self.L.pn_featgen_create(device, self.n_pairs, stream)
        if not self.h:
            raise PercepNetError(_err(self.L))

    def close(self):
        if self.h:
            self.L.pn_featgen_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PercepNetError(_err(self.L))

    def reset(self):
        self._chk(self.L.pn_featgen_reset(self.h))

    def synchronize(self):
        self._chk(self.L.pn_featgen_synchronize(self.h))

    def device_bytes(self):
        return self.L.pn_featgen_device_bytes(self.h)

    def process_dev(self, d_speech, d_noisy, d_records, d_test_pcm=None):
        """One frame; device pointers (ints): [n_pairs][480] i16 x2 -> [n_pairs][138] f32 (+ [n_pairs][480] i16)."""
        self._chk(self.L.pn_featgen_process_i16(self.h, d_speech, d_noisy, d_records, d_test_pcm))

    def process_files_dev(self, d_speech, d_noisy, n_frames, d_records, d_test_pcm=None):
        self._chk(self.L.pn_featgen_process_i16_files(self.h, d_speech, d_noisy, n_frames, d_records, d_test_pcm))

    def run(self, speech, noisy, want_test_pcm=True):
        """speech, noisy: int16 [n_pairs, n_frames*480] -> (records [n_pairs, n_frames, 138] f32,
        test_output [n_pairs, n_frames, 480] i16 or None).  Continues from the current state."""
        speech = np.ascontiguousarray(speech, dtype=np.int16).reshape(self.n_pairs, -1)
        noisy = np.ascontiguousarray(noisy, dtype=np.int16).reshape(self.n_pairs, -1)
        n = min(speech.shape[1], noisy.shape[1]) // 480
        speech = np.ascontiguousarray(speech[:, :n * 480]); noisy = np.ascontiguousarray(noisy[:, :n * 480])
        rec = np.empty((self.n_pairs, n, 138), np.float32)
        pcm = np.empty((self.n_pairs, n, 480), np.int16) if want_test_pcm else None
        self._chk(self.L.pn_featgen_process_host_i16_files(self.h, speech.ctypes.data, noisy.ctypes.data, n,
                                                           rec.ctypes.data, pcm.ctypes.data if want_test_pcm else None))
        return rec, pcm


def featgen_run_files(jobs, device=0, test_pcm=False):
    """jobs: [(speech_path, noisy_path, count, output_path), ...] — pn_featgen_run_files."""
    L = load_library()
    n = len(jobs)
    arr = lambda xs: (ctypes.c_char_p * n)(*[x.encode() if x is not None else None for x in xs])
    sp, no, out = arr([j[0] for j in jobs]), arr([j[1] for j in jobs]), arr([j[3] for j in jobs])
    cnt = (ctypes.c_int * n)(*[int(j[2]) for j in jobs])
    to = arr([j[3] + ".test_output.pcm" for j in jobs]) if test_pcm else None
    ti = arr([j[3] + ".test_input.pcm" for j in jobs]) if test_pcm else None
    if L.pn_featgen_run_files(device, n, sp, no, cnt, out, to, ti) != 0:
        raise PercepNetError(_err(L))
