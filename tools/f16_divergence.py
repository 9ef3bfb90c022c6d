"""Where does the fp16 variant first diverge between two identical runs?  Two contexts are created BEFORE the inputs (as
bench.py does), fed the same frames, and every internal activation buffer is compared after each frame."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = api.NN_MFMA_F16 if (len(sys.argv) < 4 or sys.argv[3] == "f16") else api.NN_MFMA
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
ctxs = [api.Context(model, B, nn_mode=mode, stream=torch.cuda.current_stream().cuda_stream) for _ in range(2)]
L = ctxs[0].L
L.pn_ctx_debug_copy.restype = ctypes.c_longlong
L.pn_ctx_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong]
P = 64
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
rot = (torch.arange(B, device=dev) // P) * 37
ar = (torch.arange(480, device=dev)[None, :] + rot[:, None]) % 480
frames = [torch.gather(pool[:, t*480:(t+1)*480][idx], 1, ar).contiguous() for t in range(T)]
torch.cuda.synchronize()      # the context has its own stream: inputs must be complete before it reads them
outs = [torch.empty((B, 480), dtype=torch.int16, device=dev) for _ in range(2)]
Bp = (B + 255) // 256 * 256
names = {0: ("feat", Bp * 128), 1: ("c1ring", 5 * Bp * 128), 2: ("c2ring", 3 * Bp * 512), 3: ("c2out", Bp * 512),
         4: ("gru1", 2 * Bp * 512), 5: ("gru2", 2 * Bp * 512), 6: ("gru3", 2 * Bp * 512), 7: ("gru_gb", 2 * Bp * 512),
         8: ("rb", 2 * Bp * 128), 9: ("gr", B * 68)}
found = False
for t in range(T):
    for c, o in zip(ctxs, outs):
        c.process_i16_dev(frames[t].data_ptr(), o.data_ptr(), None)
    torch.cuda.synchronize()
    for w, (nm, n) in names.items():
        a = np.empty(n, np.float32); b = np.empty(n, np.float32)
        assert L.pn_ctx_debug_copy(ctxs[0].h, w, a.ctypes.data, a.nbytes) > 0
        assert L.pn_ctx_debug_copy(ctxs[1].h, w, b.ctypes.data, b.nbytes) > 0
        d = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
        if len(d):
            width = {"feat": 128, "c1ring": 128, "rb": 128, "gr": 68}.get(nm, 512)
            rows = (d // width) % (Bp if nm != "gr" else B)
            print(f"frame {t}: {nm} differs in {len(d)} elements; rows {rows.min()}..{rows.max()} ({len(np.unique(rows))} rows); "
                  f"cols {np.unique(d % width)[:12].tolist()}; max|d| {np.abs(a[d]-b[d]).max():.3g}; row%128 {np.unique(rows % 128)[:10].tolist()}")
            found = True
    if found:
        break
print("no divergence" if not found else "done")
