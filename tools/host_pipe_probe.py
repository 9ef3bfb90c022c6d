"""Where the pipelined host path (pn_submit_host_i16) loses time against the device-resident loop: ms per frame of both in
a tight loop, and the per-kernel HIP-event times inside each.  usage: python tools/host_pipe_probe.py [streams] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B)
pool = synth.synth_batch(64, 4)
NS = 3
h_in = [torch.empty((B, 480), dtype=torch.int16).pin_memory() for _ in range(NS)]
h_out = [torch.empty((B, 480), dtype=torch.int16).pin_memory() for _ in range(NS)]
for k in range(NS): h_in[k].numpy()[:] = pool[np.arange(B) % 64, k * 480:(k + 1) * 480]
d_in = h_in[0].cuda(); d_out = torch.empty_like(d_in); torch.cuda.synchronize()
tag = " ".join(f"{k}={os.environ[k]}" for k in ("HSA_ENABLE_SDMA", "GPU_MAX_HW_QUEUES", "PN_PROBE_SMI") if k in os.environ) or "default env"
def kt_sum():
    kt = ctx.kernel_times(); return sum(v[0] / max(v[1], 1) for v in kt.values()), kt
import ctypes
n = B * 480
a_in = [ctx.L.pn_host_alloc(n * 2) for _ in range(NS)]; a_out = [ctx.L.pn_host_alloc(n * 2) for _ in range(NS)]
for k in range(NS): ctypes.memmove(a_in[k], h_in[k].data_ptr(), n * 2)
sampler = None
if os.environ.get("PN_PROBE_SMI"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sampler = bench.ClockSampler(); sampler.__enter__()
for mode in ("device", "host", "host_pn_alloc"):
    for prof in (False,):
        ctx.reset(); ctx.set_profiling(prof); ctx.reset_profile()
        for t in range(T + 6):
            if t == 6:
                ctx.host_wait() if mode != "device" else ctx.synchronize(); ctx.reset_profile(); t0 = time.perf_counter()
            if mode == "host": ctx.submit_host_i16(h_in[t % NS].data_ptr(), h_out[t % NS].data_ptr(), None)
            elif mode == "host_pn_alloc": ctx.submit_host_i16(a_in[t % NS], a_out[t % NS], None)
            else: ctx.process_i16_dev(d_in.data_ptr(), d_out.data_ptr(), None)
        ctx.host_wait() if mode != "device" else ctx.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / T
        line = f"[{tag}] B={B} {mode:6s} profiling={int(prof)} ms/frame={ms:.3f}"
        if prof:
            s, kt = kt_sum()
            line += f" kernels_sum={s:.3f} | " + " ".join(f"{k}={v[0]/max(v[1],1):.3f}" for k, v in kt.items())
        print(line, flush=True)
if sampler: sampler.__exit__(None, None, None); print(sampler.summary())
