"""Per-phase shader-clock breakdown of pn_frontend_kernel (variant build with -DPN_FE_CLOCKS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
NAMES = ["hist write", "window+scatter", "look-ahead FFT", "Y store + Ey bands", "downsample", "autocorr+LPC", "FIR",
         "coarse xcorr", "best-pitch coarse", "fine xcorr", "best-pitch fine", "rd: 28 dot chains", "rd: yy_lookup",
         "rd: decide + 3 chains", "comb+window+scatter", "P FFT", "P store + X.P", "Ep bands", "Exp bands + features"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, stream=torch.cuda.current_stream().cuda_stream)
L = ctypes.CDLL(api.LIB_PATH) if not os.environ.get("PERCEPNET_LIB") else ctypes.CDLL(os.environ["PERCEPNET_LIB"])
P = min(B, 64); T = 8
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(T)]
torch.cuda.synchronize()      # the context has its own stream: inputs must be complete before it reads them
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
buf = (ctypes.c_ulonglong * 24)()
for t in range(T):
    if t == T - 3:
        torch.cuda.synchronize(); L.pn_fe_clocks_read(buf, 1)
    ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
n = L.pn_fe_clocks_read(buf, 0)
v = np.array(list(buf)[:len(NAMES)], dtype=np.float64)
tot = v.sum()
print(f"B={B}: total marked cycles/frame/block-wave0 = {tot/3:.0f}")
for nm, c in zip(NAMES, v):
    print(f"  {nm:24s} {100*c/tot:5.1f} %")
