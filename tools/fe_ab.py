"""Front-end kernels in STEADY STATE (round 5): 65 536 streams (64 pool streams tiled, as bench.py does), WARM frames from the
zero state first — the history ring holds 12 frames, and until it has filled the pitch search returns degenerate periods
(T = 768: the comb window is then at its largest and nothing overlaps) — then HIP-event times of the next frames and the
periods they filtered at.  PERCEPNET_LIB selects a variant library.  Under rocprofv3 the LAST `frames` dispatches of each
kernel are the steady-state ones (tools/gpu_fe_ab.sh keeps only those)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = api.Context(model, B, stream=stream.cuda_stream)
P = min(B, 64); T = WARM + K
pool = torch.from_numpy(synth.synth_batch(P, T, base_seed=synth.BASE_SEED)).to(dev)
idx = torch.arange(B, device=dev) % P
rot = (torch.arange(B, device=dev) // P) * 37
ar = (torch.arange(480, device=dev)[None, :] + rot[:, None]) % 480
frames = [torch.gather(pool[:, t * 480:(t + 1) * 480][idx], 1, ar).contiguous() for t in range(T)]
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for t in range(WARM):
    ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize(); ctx.reset_profile(); ctx.set_profiling(True)
per = []
for t in range(WARM, T):
    ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
    ctx.synchronize()
    per.append(np.frombuffer(ctx.debug_copy(13, B).tobytes(), dtype=np.int32).astype(np.int64))
kt = ctx.kernel_times()
per = np.stack(per)
name = os.environ.get("PERCEPNET_LIB", "default/x").split("/")[-2]
print(f"{name:12s} B={B} warm={WARM} frames={K}: " + " ".join(f"{k}={v[0]/max(v[1],1):.4f}" for k, v in kt.items() if k.startswith("fe_") or k == "backend")
      + f" | period mean {per.mean():.1f} min {per.min()} max {per.max()}, comb window mean {((960 + 6 * per) * 4).mean():.0f} B/stream", flush=True)
