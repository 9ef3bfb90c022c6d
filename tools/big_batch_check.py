"""Maximum-size check (tools/big_batch_check.py <streams> [frames]): a batch far beyond the benchmarked 65 536 built from 16 distinct
streams — every replica must be bit-identical wherever it sits (64-bit indexing, grid limits, ring arithmetic) and the first 16
slots must match the CPU oracle (features bit-equal, g/r, PCM)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from percepnet_amd import api, synth, weights
from oracle.oracle import Oracle
B, T, P = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 9, 16
blob = weights.default_blob(1234); model = api.Model(blob); orc = Oracle(blob)
base = synth.synth_batch(P, T)
ro, rg, rf, rs = orc.run_batch(base)
dev = torch.device("cuda:0"); ts = torch.cuda.Stream(dev)
with torch.cuda.stream(ts):
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA, stream=ts.cuda_stream)
    print("state GiB", ctx.device_bytes() / 2**30)
    idx = torch.arange(B, device=dev) % P
    dbase = torch.from_numpy(base).to(dev)
    o = torch.empty((B, 480), dtype=torch.int16, device=dev); g = torch.empty((B, 68), dtype=torch.float32, device=dev)
    f = torch.empty((B, 70), dtype=torch.float32, device=dev); s = torch.empty((B,), dtype=torch.int32, device=dev)
    for t in range(T):
        fr = dbase[:, t*480:(t+1)*480][idx].contiguous()
        ctx.process_i16_dev(fr.data_ptr(), o.data_ptr(), g.data_ptr()); ctx.read_features_dev(f.data_ptr(), s.data_ptr())
        torch.cuda.synchronize()
        ok_o = bool((o == o[:P][idx]).all()); ok_g = bool((g.view(torch.int32) == g[:P][idx].view(torch.int32)).all())
        ok_f = bool((f.view(torch.int32) == f[:P][idx].view(torch.int32)).all())
        fo = np.array_equal(f[:P].cpu().numpy().view(np.uint32), rf[:, t].view(np.uint32))
        dg = np.abs(g[:P].cpu().numpy() - rg[:, t]).max()
        dp = 0 if t == 0 else np.abs(o[:P].cpu().numpy().astype(np.int32) - ro[:, (t-1)*480:t*480].astype(np.int32)).max()
        print(t, "replicas identical pcm/gr/feat:", ok_o, ok_g, ok_f, "| first P vs oracle: feat bit-equal", fo, "dgr %.2e" % dg, "dpcm", dp)
    ctx.close()
