"""Throughput of the batched training-feature generator with inputs resident in
HBM, next to the CPU reference's train() (oracle/_ref) on one host core.

    python tools/featgen_bench.py [n_pairs] [n_frames]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from percepnet_amd import api, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
K = 16
bsp, bno = synth.synth_pairs(K, T)
idx = np.arange(B) % K
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    d_sp = torch.from_numpy(bsp[idx]).cuda(); d_no = torch.from_numpy(bno[idx]).cuda()       # [B][T*480] file images
    d_rec = torch.empty((B, T, 138), dtype=torch.float32, device="cuda")
    d_pcm = torch.empty((B, T, 480), dtype=torch.int16, device="cuda")
    fg = api.FeatGen(B, stream=st.cuda_stream)
    fg.process_files_dev(d_sp.data_ptr(), d_no.data_ptr(), T, d_rec.data_ptr(), d_pcm.data_ptr())    # warm-up
    fg.synchronize(); fg.reset(); fg.synchronize()
    t0 = time.perf_counter()
    fg.process_files_dev(d_sp.data_ptr(), d_no.data_ptr(), T, d_rec.data_ptr(), d_pcm.data_ptr())
    fg.synchronize()
    dt = time.perf_counter() - t0
res = {"n_pairs": B, "n_frames": T, "gpu_pair_frames_per_s": B * T / dt, "ms_per_frame_of_batch": dt / T * 1e3,
       "device_bytes": fg.device_bytes()}
try:
    from oracle.oracle import Reference, ref_available
    if ref_available():
        ref = Reference(None)
        sp, no = synth.synth_pair(0, 300)
        with tempfile.TemporaryDirectory() as d:
            t0 = time.perf_counter(); ref.train(sp, no, d); c = time.perf_counter() - t0
        res["cpu_reference_train_frames_per_s_one_core"] = 300 / c
        res["speedup_vs_one_core"] = res["gpu_pair_frames_per_s"] / (300 / c)
except Exception as e:  # noqa
    res["cpu_reference"] = f"unavailable: {e}"
print(json.dumps(res))
