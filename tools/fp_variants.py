"""Pitch-kernel variants (PERCEPNET_LIB = a build.build_variant library): a hash of the pitch features / PCM of 512 distinct
streams over 60 frames (compare with the default library run with PERCEPNET_FE=mono, an independent kernel) and the
HIP-event time of pn_fe_pitch_kernel at 65 536 streams."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from percepnet_amd import api, synth, weights
dev = torch.device("cuda:0")
model = api.Model(weights.default_blob(1234))
name = os.environ.get("PERCEPNET_LIB", "default/x").split("/")[-2] + ("+" + os.environ["PERCEPNET_FE"] if "PERCEPNET_FE" in os.environ else "")
B, T = 512, 60
pcm = synth.synth_batch_parallel(B, T)
ctx = api.Context(model, B, nn_mode=api.NN_STRICT, stream=torch.cuda.current_stream().cuda_stream)
h = hashlib.sha256()
d = torch.from_numpy(pcm).to(dev); out = torch.empty((B, 480), dtype=torch.int16, device=dev)
for t in range(T):
    fr = d[:, t*480:(t+1)*480].contiguous(); torch.cuda.synchronize()
    ctx.process_i16_dev(fr.data_ptr(), out.data_ptr(), None); torch.cuda.synchronize()
    f, sil = ctx.read_features()
    h.update(f[:, 68:70].tobytes()); h.update(out.cpu().numpy().tobytes())
ctx.close()
line = f"{name:28s} hash {h.hexdigest()[:16]}"
if "PERCEPNET_FE" not in os.environ:
    B = 65536; K = 6
    ctx = api.Context(model, B, stream=torch.cuda.current_stream().cuda_stream)
    pool = torch.from_numpy(pcm[:64, :480 * (K + 2)]).to(dev); idx = torch.arange(B, device=dev) % 64
    frames = [pool[:, t*480:(t+1)*480][idx].contiguous() for t in range(K + 2)]
    out = torch.empty((B, 480), dtype=torch.int16, device=dev); torch.cuda.synchronize()
    for t in range(2): ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize(); ctx.reset_profile(); ctx.set_profiling(True)
    for t in range(2, K + 2): ctx.process_i16_dev(frames[t].data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize(); kt = ctx.kernel_times()
    line += "  " + " ".join(f"{k}={v[0]/max(v[1],1):.4f}" for k, v in kt.items() if k.startswith("fe_"))
print(line, flush=True)
