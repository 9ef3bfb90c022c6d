"""Round 6: do two independent half-batch pipelines on two HIP streams overlap (one context's DSP kernels beside the other's
MFMA-bound network)?  usage: python tools/two_ctx_overlap.py [streams_total] [frames]
Prints ms per frame of ONE context of B streams and of TWO contexts of B / 2 streams whose frames are enqueued alternately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
P, T = 64, 32 + K
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)


def make(n, offset):
    st = torch.cuda.Stream(dev)
    ctx = api.Context(model, n, nn_mode=api.NN_MFMA, stream=st.cuda_stream)
    idx = (torch.arange(n, device=dev) + offset) % P
    frames = [pool[:, t * 480:(t + 1) * 480][idx].contiguous() for t in range(T)]
    out = torch.empty((n, 480), dtype=torch.int16, device=dev)
    return st, ctx, frames, out


def run(ctxs, reps=3):
    torch.cuda.synchronize()
    for t in range(32):
        for st, ctx, frames, out in ctxs:
            ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for i in range(K):
            for st, ctx, frames, out in ctxs:
                ctx.process_i16_dev(frames[32 + i].data_ptr(), out.data_ptr(), None)
        torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t0) / K)
    return min(ms)


one = [make(B, 0)]
print(f"one context of {B} streams: {run(one):.4f} ms per frame  [{one[0][1].describe()}]", flush=True)
one[0][1].close()
two = [make(B // 2, 0), make(B // 2, 17)]
print(f"two contexts of {B // 2} streams on two streams, frames enqueued alternately: {run(two):.4f} ms per {B} stream-frames", flush=True)
# half a frame out of phase: context B starts its frame while context A is in its network
st, ctx, frames, out = two[0]
ctx.process_i16_dev(frames[31].data_ptr(), out.data_ptr(), None)          # one extra frame on A
print(f"... after one extra frame on the first context (out of phase): {run(two):.4f}", flush=True)
