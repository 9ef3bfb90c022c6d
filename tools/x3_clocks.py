"""Phase breakdown (shader cycles, wave 0 of each block) of pn_gru_x3_kernel at N=512: variant built with -DPN_X3_CLOCKS."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = 65536
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, nn_mode=api.NN_MFMA_X3, stream=torch.cuda.current_stream().cuda_stream)
L = ctypes.CDLL(os.environ["PERCEPNET_LIB"])
pool = torch.from_numpy(synth.synth_batch(64, 6)).to(dev)
idx = torch.arange(B, device=dev) % 64
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for t in range(6):
    fr = pool[:, t*480:(t+1)*480][idx].contiguous(); torch.cuda.synchronize()
    ctx.process_i16_dev(fr.data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
tr = (ctypes.c_ulonglong * (4096 * 8))()
assert L.pn_x3_trace_read(tr) == 0
a = np.array(list(tr), dtype=np.int64).reshape(4096, 8)
d = np.diff(a[:, :5], axis=1)
names = ["prologue", "x-phase (16 tiles)", "h-phase (16 tiles)", "epilogue"]
for i, n in enumerate(names):
    print(f"{n:22s} mean {d[:, i].mean():9.0f} cycles  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}")
tot = a[:, 4] - a[:, 0]
print(f"block total            mean {tot.mean():9.0f}; MFMA-only K loop would be {32 * 36 * 32} cycles per wave, x2 waves per SIMD = {2 * 32 * 36 * 32}")
wall = (a[:, 5] - a[:, 7]) / 100.0          # microseconds per block (100 MHz counter)
print(f"block wall time mean {wall.mean():.1f} us -> shader clock {tot.mean() / wall.mean():.0f} MHz; launch span {(a[:, 5].max() - a[:, 7].min()) / 100.0:.1f} us")
hw = a[:, 6]; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("blocks per (se, sh, cu) id:", np.bincount((se * 32 + sh * 16 + cu).astype(int)).max(), "(max; XCDs not distinguished)")
