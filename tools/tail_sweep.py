"""Frame time against the batch size around the round boundary of the batch-GEMM kernels (round 6, verdict item 3):
B = 65536 + 512 k, k = 0..8, with the network as ONE chain (PN_NN_CHAINS=1: the staircase of whole rounds per layer) and as row-range
chains on streams of their own (PN_NN_CHAINS=2, the default above 16 384 streams).  One process per point (tools/frame_time.py): ms per
frame without events.    python tools/tail_sweep.py [chains ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
chains = sys.argv[1:] or ["1", "2"]
for k in range(9):
    B = 65536 + 512 * k
    for n in chains:
        env = dict(os.environ, PN_NN_CHAINS=n)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "frame_time.py"), str(B), "30", "3"], env=env, capture_output=True, text=True, timeout=600)
        line = (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]
        print(f"chains={n} linear_from_65536={B / 65536:.4f}x | {line}", flush=True)
