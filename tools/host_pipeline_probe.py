"""Where does the pipelined host path (pn_submit_host_i16) lose time when it runs BACK TO BACK (behind its clock)?

    python tools/host_pipeline_probe.py [streams] [frames]

  A  device-resident frames back to back (pn_process_i16 on device buffers): the kernels alone
  B  pn_submit_host_i16 back to back (the call blocks in hipEventSynchronize on the delivery of frame t - 2)
  C  the same, but the host polls pn_host_frames_delivered and submits only when the call will not block
each with the per-family HIP-event kernel times off and on (on: which kernels stretch when the copies run beside them).
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from percepnet_amd import api, synth, weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
FRAME = 480
busy = []
if os.environ.get("PN_PROBE_BUSY"):                      # a process that already owns streams (bench.py: torch's side streams, the bench context)
    for _ in range(6):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            busy.append((st, torch.zeros(16, device="cuda") + 1))
    torch.cuda.synchronize()
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, device=0, nn_mode=api.NN_MFMA)
L = ctx.L
src = synth.synth_batch(64, 3, base_seed=synth.BASE_SEED + 31337)
n = B * FRAME
bufs = []
for k in range(3):
    hin, hout = L.pn_host_alloc(n * 2), L.pn_host_alloc(n * 2)
    fr = np.ascontiguousarray(src[np.arange(B) % 64, k * FRAME:(k + 1) * FRAME])
    ctypes.memmove(hin, fr.ctypes.data, n * 2)
    bufs.append((hin, hout, torch.from_numpy(fr).cuda()))
dout = torch.empty((B, FRAME), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()


def kt_line(kt):
    return " ".join(f"{k}={ms / max(c, 1):.4f}x{c // N}" for k, (ms, c) in sorted(kt.items()) if c)


def run(tag, fn, prof):
    ctx.host_wait(); ctx.reset()
    for k in range(12):
        fn(k)
    ctx.host_wait(); torch.cuda.synchronize()
    ctx.set_profiling(prof); ctx.reset_profile()
    t = time.perf_counter()
    for k in range(N):
        fn(k)
    ctx.host_wait(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / N
    line = f"{tag:34s} prof={int(prof)} {ms:8.4f} ms/frame"
    if prof:
        kt = ctx.kernel_times()
        line += f"  sum of kernel events {sum(v[0] for v in kt.values()) / N:.4f} | " + kt_line(kt)
    ctx.set_profiling(False)
    print(line, flush=True)


def dev(k):
    ctx.process_i16_dev(bufs[k % 3][2].data_ptr(), dout.data_ptr(), 0)


def sub(k):
    ctx.submit_host_i16(bufs[k % 3][0], bufs[k % 3][1])


state = {"sub": 0, "base": 0}


def sub_poll(k):
    while state["sub"] - (L.pn_host_frames_delivered(ctx.h) - state["base"]) >= 2:
        pass
    ctx.submit_host_i16(bufs[k % 3][0], bufs[k % 3][1]); state["sub"] += 1


print(f"streams {B}, {N} frames per run; env HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA')} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} PN_PIPE_PRIO={os.environ.get('PN_PIPE_PRIO')} busy={len(busy)}")
only = os.environ.get("PN_PROBE_ONLY")
for prof in ((False,) if os.environ.get('PN_PROBE_QUICK') else (False, True)):
    if only in (None, "A"):
        run("A device-resident back to back", dev, prof)
    if only in (None, "B"):
        run("B submit_host back to back", sub, prof)
    if only in (None, "C"):
        ctx.host_wait(); state["sub"] = 0; state["base"] = L.pn_host_frames_delivered(ctx.h)
        # warm-up frames of run() count too: keep the counters in step
        run("C submit_host, poll then submit", sub_poll, prof)
print("copy streams:", ctx.pipe_streams())
ctx.close()
