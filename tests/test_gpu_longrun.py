"""GPU parity at the BENCHMARKED configurations and at the horizon SURVEY §8(d) asks for (>= 1000 frames = 10 s):

  * configs[1]: 1024 distinct streams x 1000 frames, MFMA network mode, EVERY stream against the CPU oracle;
  * configs[2]: 65 536 concurrent streams x 1000 frames, a fixed sample of 256 distinct streams (every stream kind)
    scattered through the batch against the oracle, the other slots carrying rotated replicas as in bench.py.

Checked per frame, for all 1000 frames: the 70 features and the silence flag BIT-EQUAL (they do not depend on the
network, so any flip of a discontinuous decision — remove_doubling's threshold ladder pitch.cpp:485-504, the +-1
offsets pitch.cpp:369-385,514-519, silence denoise.cpp:433 — shows here), PCM within +-1 LSB and the g/r tap within
2e-5 (north_star tolerance; the MFMA network is a fused-multiply-add chain, so its recurrent state may drift from the
CPU's separately rounded one — this is the test that bounds that drift over 10 s of audio).

The oracle side runs on all host cores through pno_run_pcm_batch (bit-identical to the single-stream oracle, pinned
in tests/test_oracle.py).  Measured statistics are written to gpurun_out/parity_*.json for profiles/.
"""
import json
import os
import time

import numpy as np
import pytest

from percepnet_amd import api, synth

pytestmark = pytest.mark.gpu

PCM_TOL_LSB = 1
GR_TOL = 2e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(blob):
    m = api.Model(blob)
    yield m
    m.close()


def _record(name, stats):
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"parity_{name}.json"), "w") as f:
            json.dump(stats, f, indent=1)
    except OSError:
        pass
    print(json.dumps(stats))


def shared_stream(dev):
    """A torch stream to run BOTH torch's kernels and the context's launches on (so they are ordered).  torch's default
    stream has handle 0, which pn_ctx_create reads as "create your own stream": never pass that."""
    import torch
    ts = torch.cuda.Stream(dev)
    assert ts.cuda_stream != 0
    return ts


def run_long(ctx, frame_of, T, rows, dev):
    """Advance ctx by T frames; frame_of(t) -> int16 device tensor [B,480].  Returns, for the batch rows `rows`
    (device index tensor or None = all): out int16 [R,(T-1)*480], gr [R,T,68], feat [R,T,70], sil [R,T] (numpy).
    Must be called inside `with torch.cuda.stream(ts)` of the stream the context was created on."""
    import torch
    B = ctx.n_streams
    R = B if rows is None else int(rows.numel())
    o = torch.empty((B, 480), dtype=torch.int16, device=dev)
    g = torch.empty((B, 68), dtype=torch.float32, device=dev)
    f = torch.empty((B, 70), dtype=torch.float32, device=dev)
    s = torch.empty((B,), dtype=torch.int32, device=dev)
    out = torch.empty((T, R, 480), dtype=torch.int16, device=dev)
    gr = torch.empty((T, R, 68), dtype=torch.float32, device=dev)
    feat = torch.empty((T, R, 70), dtype=torch.float32, device=dev)
    sil = torch.empty((T, R), dtype=torch.int32, device=dev)
    for t in range(T):
        fr = frame_of(t)
        ctx.process_i16_dev(fr.data_ptr(), o.data_ptr(), g.data_ptr())      # same stream as torch's: ordered
        ctx.read_features_dev(f.data_ptr(), s.data_ptr())
        if rows is None:
            out[t].copy_(o); gr[t].copy_(g); feat[t].copy_(f); sil[t].copy_(s)
        else:
            torch.index_select(o, 0, rows, out=out[t]); torch.index_select(g, 0, rows, out=gr[t])
            torch.index_select(f, 0, rows, out=feat[t]); torch.index_select(s, 0, rows, out=sil[t])
    torch.cuda.synchronize()
    out = out[1:].permute(1, 0, 2).reshape(R, (T - 1) * 480).cpu().numpy()      # first output frame dropped (main.cpp:37)
    return (out, gr.permute(1, 0, 2).contiguous().cpu().numpy(), feat.permute(1, 0, 2).contiguous().cpu().numpy(),
            sil.permute(1, 0).contiguous().cpu().numpy())


def compare(name, got, ref, extra):
    out, gr, feat, sil = got
    ro, rg, rf, rs = ref
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    dg = np.abs(gr - rg)
    fbits = feat.view(np.uint32) != rf.view(np.uint32)
    stats = dict(extra)
    stats.update({
        "streams_checked": int(out.shape[0]), "frames": int(gr.shape[1]),
        "max_abs_delta_pcm_lsb": int(d.max()), "pcm_samples_differing": int((d != 0).sum()), "pcm_samples": int(d.size),
        "pcm_delta_histogram_lsb": np.bincount(np.minimum(d, 8).ravel().astype(np.int64), minlength=9).tolist(),
        "max_abs_delta_gr": float(dg.max()), "mean_abs_delta_gr": float(dg.mean()),
        "feature_words_differing": int(fbits.sum()), "feature_words": int(fbits.size),
        "silence_flags_differing": int((sil != rs).sum()), "non_silent_frames": int((rs == 0).sum()),
        "max_abs_delta_gr_by_second": [float(dg[:, k:k + 100].max()) for k in range(0, dg.shape[1], 100)],
    })
    _record(name, stats)
    assert not fbits.any(), f"{int(fbits.sum())} feature words differ (first at {np.argwhere(fbits)[0]})"
    assert np.array_equal(sil, rs)
    assert d.max() <= PCM_TOL_LSB, (int(d.max()), np.argwhere(d == d.max())[0])
    assert dg.max() <= GR_TOL, float(dg.max())
    return stats


def test_configs1_1024_streams_1000_frames_every_stream_vs_oracle(model, oracle):
    import torch
    B, T = 1024, 1000
    dev = torch.device("cuda:0")
    t0 = time.time()
    pcm = synth.synth_batch_parallel(B, T)
    t_synth = time.time() - t0
    t0 = time.time()
    ref = oracle.run_batch(pcm, group=8)
    t_oracle = time.time() - t0
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pcm = torch.from_numpy(pcm).to(dev)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
        t0 = time.time()
        got = run_long(ctx, lambda t: d_pcm[:, t * 480:(t + 1) * 480].contiguous(), T, None, dev)
        t_gpu = time.time() - t0
        ctx.close()
    st = compare("configs1_1024x1000", got, ref, {
        "config": "configs[1]: 1024 concurrent streams, fp32 MFMA network, every stream vs the CPU oracle",
        "synth_s": round(t_synth, 1), "oracle_s": round(t_oracle, 1), "gpu_s": round(t_gpu, 1)})
    assert st["non_silent_frames"] > 1000          # the pitch-filter branch really ran (loud streams)


def sample_layout(B, P):
    """Batch slot -> (pool stream, in-frame rotation): P sample slots scattered over the whole batch (first, last,
    tile and grid-stride-round boundaries included) carry the P distinct streams unrotated; every other slot carries
    pool stream (slot % P) with its samples rotated inside each frame, as bench.py fills its batch."""
    rng = np.random.default_rng(7)
    fixed = np.array([0, 1, 127, 128, 255, 256, 4095, 4096, 4097, B - 129, B - 2, B - 1])
    rest = np.setdiff1d(rng.permutation(B)[:2 * P], fixed, assume_unique=False)
    slots = np.sort(np.concatenate([fixed, rng.permutation(rest)[:P - fixed.size]]))
    assert slots.size == P and np.unique(slots).size == P and slots[-1] == B - 1
    idx = np.arange(B) % P
    rot = ((np.arange(B) // P) * 37 + 11) % 480
    idx[slots] = np.arange(P)
    rot[slots] = 0
    return slots, idx, rot


def test_configs2_65536_streams_1000_frames_256_sampled_vs_oracle(model, oracle):
    import torch
    B, T, P = 65536, 1000, 256
    dev = torch.device("cuda:0")
    pool = synth.synth_batch_parallel(P, T)
    kinds = {synth.stream_kind(s) for s in range(P)}
    assert kinds == {"voiced", "loud", "bursts", "twotone"}
    t0 = time.time()
    ref = oracle.run_batch(pool, group=8)
    t_oracle = time.time() - t0
    slots, idx, rot = sample_layout(B, P)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pool = torch.from_numpy(pool).to(dev)
        d_idx = torch.from_numpy(idx).to(dev)
        ar = (torch.arange(480, device=dev)[None, :] + torch.from_numpy(rot).to(dev)[:, None]) % 480
        rows = torch.from_numpy(slots).to(dev)

        def frame_of(t):
            return torch.gather(d_pool[:, t * 480:(t + 1) * 480][d_idx], 1, ar).contiguous()

        ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
        t0 = time.time()
        got = run_long(ctx, frame_of, T, rows, dev)
        t_gpu = time.time() - t0
        ctx.close()
    compare("configs2_65536x1000_sample256", got, ref, {
        "config": "configs[2]: 65536 concurrent streams, fp32 MFMA network, 256 distinct sampled streams vs the CPU oracle",
        "sample_slots_first_last": [int(slots[0]), int(slots[-1])], "oracle_s": round(t_oracle, 1), "gpu_s": round(t_gpu, 1)})


def test_fp16_variant_1000_frames_tolerance_holds(model, oracle):
    """configs[4] (fp16 GEMM operands, fp32 accumulate/state/DSP) over the same 10 s horizon: the re-stated tolerance
    (PCM <= 6 LSB stated, 4 measured; g/r <= 1e-3) must hold for all 1000 frames — the recurrent state sees rounded
    operands every step, so this is where a slow drift would show — and the features, which never touch the
    network, stay bit-equal."""
    import torch
    B, T = 256, 1000
    dev = torch.device("cuda:0")
    pcm = synth.synth_batch_parallel(B, T, first_stream=300)
    ref = oracle.run_batch(pcm, group=8)
    ts = shared_stream(dev)
    with torch.cuda.stream(ts):
        d_pcm = torch.from_numpy(pcm).to(dev)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA_F16, stream=ts.cuda_stream)
        out, gr, feat, sil = run_long(ctx, lambda t: d_pcm[:, t * 480:(t + 1) * 480].contiguous(), T, None, dev)
        ctx.close()
    ro, rg, rf, rs = ref
    d = np.abs(out.astype(np.int32) - ro.astype(np.int32))
    dg = np.abs(gr - rg)
    _record("configs4_fp16_256x1000", {
        "config": "configs[4]: fp16 operands, 256 streams x 1000 frames vs the CPU oracle",
        "max_abs_delta_pcm_lsb": int(d.max()), "pcm_delta_histogram_lsb": np.bincount(np.minimum(d, 8).ravel().astype(np.int64), minlength=9).tolist(),
        "max_abs_delta_gr": float(dg.max()), "mean_abs_delta_gr": float(dg.mean()),
        "max_abs_delta_gr_by_second": [float(dg[:, k:k + 100].max()) for k in range(0, T, 100)]})
    assert np.array_equal(feat.view(np.uint32), rf.view(np.uint32)) and np.array_equal(sil, rs)
    assert d.max() <= 6, int(d.max())            # stated bound: 6 LSB (measured max 4; tests/test_gpu_stress_weights.py has the larger sample)
    assert dg.max() <= 1e-3, float(dg.max())
