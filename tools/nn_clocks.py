"""Effective shader clock and cycles per block of the pipelined GRU kernel (variant built with -DPN_NN_CLOCKS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
ctx = api.Context(model, B, stream=torch.cuda.current_stream().cuda_stream)
L = ctypes.CDLL(os.environ.get("PERCEPNET_LIB", api.LIB_PATH))
P = min(B, 64); T = 10
pool = torch.from_numpy(synth.synth_batch(P, T)).to(dev)
idx = torch.arange(B, device=dev) % P
frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(T)]
torch.cuda.synchronize()      # the context has its own stream: inputs must be complete before it reads them
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
buf = (ctypes.c_ulonglong * 4)()
for t in range(T):
    if t == 4:
        torch.cuda.synchronize(); L.pn_nn_clocks_read(buf, 1)
    ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
L.pn_nn_clocks_read(buf, 0)
cyc, real, n = buf[0], buf[1], buf[2]
print(f"blocks={n} shader cycles/block={cyc/n:.0f} realtime ticks/block={real/n:.1f} -> shader clock = {cyc/real*100:.0f} MHz (100 MHz reference)")
print(f"  MFMA cycles per block-wave if back-to-back: gru512 {32*48*64} ; gru_rb {36*48*64}")

# ---- per-block timeline of the last N=512 GRU launch -------------------------------------------------
tr = (ctypes.c_ulonglong * (8192 * 4))()
if hasattr(L, "pn_nn_trace_read") and L.pn_nn_trace_read(tr) == 0:
    a = np.array(list(tr), dtype=np.int64).reshape(8192, 4)
    st, en, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3] & 0xf
    t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0        # microseconds
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    print(f"launch span {en.max():.1f} us; block duration mean {np.mean(en-st):.1f} us (min {np.min(en-st):.1f}, max {np.max(en-st):.1f}); "
          f"distinct CUs {len(np.unique(key))}; first-start spread {np.sort(st)[511]:.1f} us; last start {st.max():.1f} us")
    busy = []; gaps = []
    for k in np.unique(key):
        m = key == k
        order = np.argsort(st[m]); s_, e_ = st[m][order], en[m][order]
        busy.append(np.sum(e_ - s_))
        # gap between a block's end and the start of the next block that begins after it (slot refill latency)
        for e in e_:
            nxt = s_[s_ >= e - 1e-9]
            if len(nxt): gaps.append(nxt.min() - e)
    busy = np.array(busy); gaps = np.array(gaps)
    print(f"blocks per CU: mean {8192/len(np.unique(key)):.1f}; per-CU block-time sum / (2 x span): mean {np.mean(busy)/(2*en.max()):.3f} min {np.min(busy)/(2*en.max()):.3f}")
    print(f"slot refill gap (end -> next start on the same CU): median {np.median(gaps):.2f} us, mean {np.mean(gaps):.2f}, p90 {np.percentile(gaps,90):.2f}")
    hist, edges = np.histogram(en, bins=10, range=(0, en.max()))
    print("block END histogram over the launch (10 bins):", hist.tolist())
    np.save("gpurun_out/nn_trace.npy", a)
