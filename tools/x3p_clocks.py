"""Phase breakdown (shader cycles, wave 0 of each wave group of every block) of pn_gru_x3p_kernel at N=512: variant built
with -DPN_X3_CLOCKS (percepnet_amd.build.build_variant("x3clk", ["-DPN_X3_CLOCKS"], only=["pn_nn_x3.hip"])).
   PERCEPNET_LIB=.../variants/x3clk/libpercepnet_hip.so PN_MODE=f16|x3 python tools/x3p_clocks.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
mode = {"f16": api.NN_MFMA_F16, "x3": api.NN_MFMA_X3}[os.environ.get("PN_MODE", "f16")]
ctx = api.Context(model, B, nn_mode=mode, stream=torch.cuda.current_stream().cuda_stream)
print(ctx.describe())
L = ctypes.CDLL(os.environ["PERCEPNET_LIB"])
pool = torch.from_numpy(synth.synth_batch(64, 6)).to(dev)
idx = torch.arange(B, device=dev) % 64
out = torch.empty((B, 480), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for t in range(6):
    fr = pool[:, t*480:(t+1)*480][idx].contiguous(); torch.cuda.synchronize()
    ctx.process_i16_dev(fr.data_ptr(), out.data_ptr(), None)
torch.cuda.synchronize()
tr = (ctypes.c_ulonglong * (256 * 2 * 8))()
assert L.pn_x3p_trace_read(tr) == 0
a = np.array(list(tr), dtype=np.int64).reshape(256, 2, 8)
n_it, TT = int(a[0, 0, 3]), int(a[0, 0, 4])
np_ = 2 if mode == api.NN_MFMA_X3 else 1
mfma = TT * 2 * 3 * 2 * (3 if np_ == 2 else 1) * 32      # MFMA cycles of one K phase: tiles x k-steps x gates x row groups x products x 32
for g in (0, 1):
    k, e, tot = a[:, g, 0] / n_it, a[:, g, 1] / n_it, a[:, g, 2]
    print(f"group {g}: K phase mean {k.mean():8.0f} cycles (p10 {np.percentile(k, 10):.0f} p90 {np.percentile(k, 90):.0f}); "
          f"E phase mean {e.mean():8.0f} (p10 {np.percentile(e, 10):.0f} p90 {np.percentile(e, 90):.0f}); kernel {tot.mean():.0f}")
print(f"tiles per group {n_it}, K tiles per phase {TT}; matrix-pipe time of one K phase = {mfma} cycles; "
      f"ideal kernel = {(2 * n_it + 1) * mfma} cycles")

hw = (ctypes.c_uint * (256 * 8))()
if hasattr(L, "pn_x3p_hwid_read") and L.pn_x3p_hwid_read(hw) == 0:
    h = np.array(list(hw), dtype=np.int64).reshape(256, 8)
    simd = (h >> 4) & 3; cu = (h >> 8) & 15; wid = h & 15
    for b in (0, 1, 100, 255):
        print(f"block {b}: SIMD of waves 0..7 = {simd[b].tolist()}  CU {cu[b].tolist()}  wave slot {wid[b].tolist()}")
    same = (simd[:, :4] == simd[:, 4:]).all(axis=1).mean()
    print(f"blocks whose waves w and w + 4 share a SIMD: {100 * same:.0f} %; blocks whose waves 0..3 sit on four different SIMDs: "
          f"{100 * np.mean([len(set(r[:4])) == 4 for r in simd]):.0f} %")
