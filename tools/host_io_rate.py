"""PCIe-inclusive rate of the host-buffer boundary (pn_process_host_i16): H2D copy, the 13 launches, D2H copy and a
stream synchronise per frame, serial.  Never the bench's `value` (inputs there are resident in HBM); DESIGN.md §6
quotes this next to it.  usage: python tools/host_io_rate.py [streams] [frames]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B)
pool = synth.synth_batch(64, T)
res = {"streams": B, "frames_timed": T - 4}
for kind in ("pinned", "pageable"):
    mk = (lambda: torch.empty((B, 480), dtype=torch.int16).pin_memory()) if kind == "pinned" else \
         (lambda: torch.empty((B, 480), dtype=torch.int16))
    h_in, h_out = mk(), mk()
    ctx.reset()
    dt = 0.0
    for t in range(T):
        h_in.numpy()[:] = pool[np.arange(B) % 64, t * 480:(t + 1) * 480]
        t0 = time.perf_counter()
        rc = ctx.L.pn_process_host_i16(ctx.h, h_in.data_ptr(), h_out.data_ptr(), None)
        t1 = time.perf_counter()
        assert rc == 0
        if t >= 4: dt += t1 - t0
    ms = 1e3 * dt / (T - 4)
    res[kind] = {"ms_per_frame": round(ms, 3), "streams_real_time": round(B / (ms / 10.0), 1),
                 "pcm_GB_per_s_each_way_over_the_whole_frame": round(B * 960 / 1e9 / (ms / 1e3), 2)}
# pipelined entry point: three rotating pinned buffer sets (a frame's buffers stay live for two more submits)
NS = 3
h_in = [torch.empty((B, 480), dtype=torch.int16).pin_memory() for _ in range(NS)]
h_out = [torch.empty((B, 480), dtype=torch.int16).pin_memory() for _ in range(NS)]
ctx.reset()
for t in range(T + 8):
    if t == 4:
        ctx.host_wait(); t0 = time.perf_counter()
    k = t % NS
    if t < NS: h_in[k].numpy()[:] = pool[np.arange(B) % 64, (t % T) * 480:(t % T + 1) * 480]
    ctx.submit_host_i16(h_in[k].data_ptr(), h_out[k].data_ptr(), None)
ctx.host_wait()
ms = 1e3 * (time.perf_counter() - t0) / (T + 4)
res["pipelined_pinned"] = {"ms_per_frame": round(ms, 3), "streams_real_time": round(B / (ms / 10.0), 1)}
print(json.dumps(res))
