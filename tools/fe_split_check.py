"""GPU check of the phase-split front end against the single-launch kernel (run on the GPU box):
bit equality of every front-end output (features, silence, look-ahead spectra ring, comb-filtered spectrum, history
ring) and of PCM / g,r, frame by frame over a ring wrap, at a ragged multi-round batch and at 65 536 streams; then the
per-kernel times of both at 65 536 streams.   python tools/fe_split_check.py [--streams 8195,65536] [--frames 14]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from percepnet_amd import api, synth, weights

ap = argparse.ArgumentParser()
ap.add_argument("--streams", default="8195,65536")
ap.add_argument("--frames", type=int, default=14)
ap.add_argument("--modes", default="mono,split")
ap.add_argument("--time-frames", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
ts = torch.cuda.Stream(dev)
torch.cuda.set_stream(ts)
modes = a.modes.split(",")
ok = True
for B in [int(x) for x in a.streams.split(",")]:
    T = a.frames
    P = min(B, 96)
    pool = torch.from_numpy(synth.synth_batch(P, max(T, a.time_frames))).to(dev)
    idx = torch.arange(B, device=dev) % P
    rot = (torch.arange(B, device=dev) // P) * 37
    ar = (torch.arange(480, device=dev)[None, :] + rot[:, None]) % 480
    frames = [torch.gather(pool[:, t * 480:(t + 1) * 480][idx], 1, ar).contiguous() for t in range(max(T, a.time_frames))]
    ctxs = {}
    for m in modes:
        os.environ["PERCEPNET_FE"] = m
        ctxs[m] = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
        print(B, m, ctxs[m].describe(), flush=True)
    outs = {m: torch.empty((B, 480), dtype=torch.int16, device=dev) for m in modes}
    grs = {m: torch.empty((B, 68), dtype=torch.float32, device=dev) for m in modes}
    nfl = {0: ctxs[modes[0]].L.pn_ctx_debug_copy}
    for t in range(T):
        taps = {}
        for m in modes:
            c = ctxs[m]
            c.process_i16_dev(frames[t].data_ptr(), outs[m].data_ptr(), grs[m].data_ptr())
            torch.cuda.synchronize()
            Bp = (B + 255) // 256 * 256
            feat = c.debug_copy(0, Bp * 128).reshape(Bp, 128)[:B]
            sil = np.empty(B, np.int32); fe = np.empty((B, 70), np.float32)
            c.L.pn_ctx_read_features(c.h, fe.ctypes.data, sil.ctypes.data)
            taps[m] = {"feat": feat.view(np.uint32), "sil": sil, "yring": c.debug_copy(10, 6 * B * 800).view(np.uint32),
                       "Ps": c.debug_copy(11, B * 800).view(np.uint32), "hist": c.debug_copy(12, B * 5768).view(np.uint32),
                       "pcm": outs[m].cpu().numpy(), "gr": grs[m].cpu().numpy().view(np.uint32)}
        ref = taps[modes[0]]
        for m in modes[1:]:
            for k, v in taps[m].items():
                if not np.array_equal(v, ref[k]):
                    bad = np.argwhere(v.reshape(B, -1) != ref[k].reshape(B, -1)) if v.size % B == 0 else np.argwhere(v != ref[k])
                    print(f"MISMATCH B={B} frame {t} mode {m} buffer {k}: {len(bad)} words differ, first {bad[:6].tolist()}", flush=True)
                    ok = False
        if not ok:
            break
    print(f"B={B}: {T} frames {'bit-identical' if ok else 'DIFFER'} across {modes}", flush=True)
    if ok and B >= 65536:
        res = {}
        for m in modes:
            c = ctxs[m]
            for t in range(3):
                c.process_i16_dev(frames[t].data_ptr(), outs[m].data_ptr(), None)
            c.reset_profile(); c.set_profiling(True)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for t in range(a.time_frames):
                c.process_i16_dev(frames[t].data_ptr(), outs[m].data_ptr(), None)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c.set_profiling(False)
            kt = c.kernel_times()
            res[m] = {"ms_per_frame": round(1e3 * dt / a.time_frames, 4), "kernels_ms": {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items() if v[1]}}
        print(json.dumps(res), flush=True)
    for c in ctxs.values():
        c.close()
sys.exit(0 if ok else 1)
