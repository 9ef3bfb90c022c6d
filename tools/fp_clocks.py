"""Per-phase shader-clock shares of pn_fe_pitch_kernel (variant build of pn_dsp_fe_split_p.hip with -DPN_FE_CLOCKS;
PERCEPNET_LIB selects it).  Wave 0 of every block accumulates s_memtime ticks between phase marks."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
NAMES = ["downsample", "LPC", "whitening FIR", "coarse xcorr (+ energies)", "coarse best-pitch", "fine xcorr (+ energy scan)",
         "fine best-pitch", "rd: 29 dot chains", "rd: yy_lookup", "rd: refinement + outputs", "autocorr chain",
         "autocorr tail + gather", "rd: decision + 3 chains"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, stream=torch.cuda.current_stream().cuda_stream)
L = ctypes.CDLL(os.environ.get("PERCEPNET_LIB", api.LIB_PATH))
P = min(B, 64); T = 8
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(T)]
torch.cuda.synchronize()
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
buf = (ctypes.c_ulonglong * 24)()
for t in range(T):
    if t == T - 3:
        torch.cuda.synchronize(); L.pn_fe_clocks_read(buf, 1)
    ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
L.pn_fe_clocks_read(buf, 0)
v = np.array(list(buf)[:len(NAMES)], dtype=np.float64)
tot = v.sum()
nblk = min((B + 15) // 16, 512); rounds = ((B + 15) // 16 + nblk - 1) // nblk
print(f"{os.environ.get('PERCEPNET_LIB', 'default').split('/')[-2]} B={B}: ticks per stream group (wave 0 of each block) = {tot/3/nblk/rounds:.0f}")
for nm, c in zip(NAMES, v):
    print(f"  {nm:24s} {100*c/tot:5.1f} %  {c/3/nblk/rounds:9.0f}")
