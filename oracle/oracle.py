"""TEST INFRASTRUCTURE — ctypes loaders for the CPU oracle (oracle/liboracle.so, the C
restatement) and for the compiled reference (oracle/_ref/libpercepnet_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (percepnet_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libpercepnet_ref.so")
REF_AVX2_SO = os.path.join(HERE, "_ref", "libpercepnet_ref_avx2.so")   # "best-effort CPU" footnote build, timing only

c_f = ctypes.POINTER(ctypes.c_float)
c_s = ctypes.POINTER(ctypes.c_short)
c_i = ctypes.POINTER(ctypes.c_int)


def _fp(a):
    return a.ctypes.data_as(c_f)


def build(force=False):
    """Compile the C restatement, and the reference itself when /root/reference is present."""
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(HERE, "percepnet_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(REF_SO) or not os.path.exists(REF_AVX2_SO)):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class Oracle:
    """The C restatement (percepnet_oracle.c)."""

    def __init__(self, blob):
        build()
        self.lib = ctypes.CDLL(ORACLE_SO)
        L = self.lib
        L.pno_model_from_blob.restype = ctypes.c_void_p
        L.pno_model_from_blob.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.pno_create.restype = ctypes.c_void_p
        L.pno_create.argtypes = [ctypes.c_void_p]
        L.pno_destroy.argtypes = [ctypes.c_void_p]
        L.pno_process_frame.argtypes = [ctypes.c_void_p, c_f, c_f, c_f]
        L.pno_run_pcm.argtypes = [ctypes.c_void_p, c_s, ctypes.c_int, c_s, c_f]
        L.pno_run_pcm_pf.argtypes = [ctypes.c_void_p, c_s, ctypes.c_int, c_s, c_f, ctypes.c_int]
        L.pno_run_float.argtypes = [ctypes.c_void_p, c_f, ctypes.c_int, c_f, c_f]
        L.pno_run_pcm_batch.argtypes = [ctypes.c_void_p, c_s, ctypes.c_int, ctypes.c_int, c_s, c_f, c_f, c_i,
                                        ctypes.c_int, ctypes.c_int]
        L.pno_frame_features.argtypes = [ctypes.c_void_p, c_f, c_f]
        L.pno_frame_features.restype = ctypes.c_int
        L.pno_compute_rnn.argtypes = [ctypes.c_void_p, c_f, c_f, c_f]
        L.pno_remove_doubling.restype = ctypes.c_float
        L.pno_remove_doubling.argtypes = [c_f, c_i, ctypes.c_int, ctypes.c_float]
        L.pno_tansig.restype = ctypes.c_float
        L.pno_tansig.argtypes = [ctypes.c_float]
        L.pno_sigmoid.restype = ctypes.c_float
        L.pno_sigmoid.argtypes = [ctypes.c_float]
        L.pno_tansig_table.restype = c_f
        self._blob = blob  # must outlive the model (borrowed pointers)
        self.model = L.pno_model_from_blob(blob, len(blob)) if blob is not None else None
        if blob is not None:
            assert self.model, "bad PNW1 blob"

    def run_pcm(self, pcm, want_gr=True, postfilter=False):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        n = pcm.size // 480
        out = np.zeros(max(n - 1, 0) * 480, np.int16)
        gr = np.zeros((n, 68), np.float32)
        self.lib.pno_run_pcm_pf(self.model, pcm.ctypes.data_as(c_s), n, out.ctypes.data_as(c_s),
                                _fp(gr) if want_gr else None, int(bool(postfilter)))
        return out, gr

    def run_float(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size // 480
        out = np.zeros(n * 480, np.float32)
        gr = np.zeros((n, 68), np.float32)
        self.lib.pno_run_float(self.model, _fp(x), n, _fp(out), _fp(gr))
        return out, gr

    def run_batch(self, pcm, want_feat=True, group=16, threads=None):
        """run_pcm for many streams on all host cores (pno_run_pcm_batch: the same per-stream arithmetic, groups of
        streams sharing each sweep over the weights).  pcm int16 [S, T*480] ->
        (out int16 [S, (T-1)*480], gr [S, T, 68], feat [S, T, 70] | None, silence int32 [S, T] | None)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        S = pcm.shape[0]; T = pcm.shape[1] // 480
        assert pcm.shape[1] == T * 480
        if threads is None:
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count() or 1
        out = np.zeros((S, max(T - 1, 0) * 480), np.int16)
        gr = np.zeros((S, T, 68), np.float32)
        feat = np.zeros((S, T, 70), np.float32) if want_feat else None
        sil = np.zeros((S, T), np.int32) if want_feat else None
        self.lib.pno_run_pcm_batch(self.model, pcm.ctypes.data_as(c_s), S, T, out.ctypes.data_as(c_s), _fp(gr),
                                   _fp(feat) if want_feat else None, sil.ctypes.data_as(c_i) if want_feat else None,
                                   int(group), int(threads))
        return out, gr, feat, sil

    def features(self, x):
        """Per-frame 70 features + silence flag, DSP only (no NN)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size // 480
        st = self.lib.pno_create(self.model)
        feat = np.zeros((n, 70), np.float32)
        sil = np.zeros(n, np.int32)
        for t in range(n):
            sil[t] = self.lib.pno_frame_features(st, _fp(x[t * 480:]), _fp(feat[t]))
        self.lib.pno_destroy(st)
        return feat, sil

    def stages(self, x):
        """Per-frame DSP stage taps of one stream (float samples, the /32768 convention of the CLI): dict of X, P, Y [n, 481] complex64,
        Ex, Ep, Exp, Ey [n, 34], period [n], feat [n, 70], silence [n], comb_buf [n, 5760] (the history AFTER each frame)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size // 480
        L = self.lib
        L.pno_frame_stages.argtypes = [ctypes.c_void_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f]
        L.pno_frame_stages.restype = ctypes.c_int
        L.pno_state_comb_buf.argtypes = [ctypes.c_void_p, c_f]
        st = L.pno_create(self.model)
        o = {k: np.zeros((n, 481, 2), np.float32) for k in ("X", "P", "Y")}
        o.update({k: np.zeros((n, 34), np.float32) for k in ("Ex", "Ep", "Exp", "Ey")})
        o["period"] = np.zeros(n, np.int32); o["feat"] = np.zeros((n, 70), np.float32); o["silence"] = np.zeros(n, np.int32)
        o["comb_buf"] = np.zeros((n, 5760), np.float32)
        for t in range(n):
            o["silence"][t] = L.pno_frame_stages(st, _fp(x[t * 480:]), _fp(o["X"][t]), _fp(o["P"][t]), _fp(o["Y"][t]), _fp(o["Ex"][t]),
                                                 _fp(o["Ep"][t]), _fp(o["Exp"][t]), _fp(o["Ey"][t]),
                                                 o["period"][t:].ctypes.data_as(c_i), _fp(o["feat"][t]))
            L.pno_state_comb_buf(st, _fp(o["comb_buf"][t]))
        L.pno_destroy(st)
        for k in ("X", "P", "Y"):
            o[k] = o[k].view(np.complex64)[..., 0]
        return o

    def train_run(self, speech, noisy, want_test_pcm=True):
        """The `percepNet` training binary on in-memory PCM -> (records [count,138], test_output [count,480])."""
        speech = np.ascontiguousarray(speech, dtype=np.int16); noisy = np.ascontiguousarray(noisy, dtype=np.int16)
        n = min(speech.size, noisy.size) // 480
        out = np.zeros((n, 138), np.float32)
        tst = np.zeros((n, 480), np.int16)
        self.lib.pno_train_run.argtypes = [c_s, c_s, ctypes.c_int, c_f, c_s]
        self.lib.pno_train_run(speech.ctypes.data_as(c_s), noisy.ctypes.data_as(c_s), n, _fp(out),
                               tst.ctypes.data_as(c_s) if want_test_pcm else None)
        return out, tst

    def tables(self):
        tw = c_f(); br = c_s(); hw = c_f(); ch = c_f(); bd = c_i()
        self.lib.pno_tables(ctypes.byref(tw), ctypes.byref(br), ctypes.byref(hw), ctypes.byref(ch),
                            ctypes.byref(bd))
        return (np.ctypeslib.as_array(tw, (960, 2)).copy(), np.ctypeslib.as_array(br, (960,)).copy(),
                np.ctypeslib.as_array(hw, (480,)).copy(), np.ctypeslib.as_array(ch, (7,)).copy(),
                np.ctypeslib.as_array(bd, (34,)).copy())

    def tansig_table(self):
        return np.ctypeslib.as_array(self.lib.pno_tansig_table(), (201,)).copy()


def ref_available():
    if not os.path.exists(REF_SO) and os.path.isdir("/root/reference/src"):
        build()
    return os.path.exists(REF_SO)


class Reference:
    """The reference itself (untouched sources compiled by oracle/Makefile `ref`)."""

    def __init__(self, blob, so=None):
        assert ref_available(), "oracle/_ref/libpercepnet_ref.so not built"
        self.lib = ctypes.CDLL(so or REF_SO)
        L = self.lib
        L.ref_load_weights.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.ref_create.restype = ctypes.c_void_p
        L.ref_destroy.argtypes = [ctypes.c_void_p]
        L.ref_process_frame.argtypes = [ctypes.c_void_p, c_f, c_f, c_f]
        L.ref_run_pcm.argtypes = [c_s, ctypes.c_int, c_s, c_f]
        L.ref_run_float.argtypes = [c_f, ctypes.c_int, c_f, c_f]
        L.ref_compute_rnn.argtypes = [ctypes.c_void_p, c_f, c_f, c_f]
        L.ref_remove_doubling.restype = ctypes.c_float
        L.ref_remove_doubling.argtypes = [c_f, c_i, ctypes.c_int, ctypes.c_float]
        if blob is not None:
            rc = L.ref_load_weights(blob, len(blob))
            assert rc == 0, rc

    def run_pcm(self, pcm, want_gr=True):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        n = pcm.size // 480
        out = np.zeros(max(n - 1, 0) * 480, np.int16)
        gr = np.zeros((n, 68), np.float32)
        self.lib.ref_run_pcm(pcm.ctypes.data_as(c_s), n, out.ctypes.data_as(c_s), _fp(gr) if want_gr else None)
        return out, gr

    def run_float(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size // 480
        out = np.zeros(n * 480, np.float32)
        gr = np.zeros((n, 68), np.float32)
        self.lib.ref_run_float(_fp(x), n, _fp(out), _fp(gr))
        return out, gr

    def train(self, speech, noisy, workdir):
        """Run the reference's train() (denoise.cpp:603-787) on two PCM arrays through real files in
        workdir, as the binary does: returns (records [count,138], test_output.pcm [count,480])."""
        speech = np.ascontiguousarray(speech, dtype=np.int16); noisy = np.ascontiguousarray(noisy, dtype=np.int16)
        n = min(speech.size, noisy.size) // 480
        sp, no, ou = (os.path.join(workdir, f) for f in ("speech.pcm", "noisy.pcm", "features.f32"))
        speech.tofile(sp); noisy.tofile(no)
        self.lib.ref_train.argtypes = [ctypes.c_char_p] * 4
        cwd = os.getcwd()
        os.chdir(workdir)           # train() drops test_input.pcm / test_output.pcm into the cwd
        try:
            rc = self.lib.ref_train(sp.encode(), no.encode(), str(n).encode(), ou.encode())
        finally:
            os.chdir(cwd)
        assert rc == 0, rc
        rec = np.fromfile(ou, np.float32).reshape(n, 138)
        tst = np.fromfile(os.path.join(workdir, "test_output.pcm"), np.int16).reshape(n, 480)
        return rec, tst
