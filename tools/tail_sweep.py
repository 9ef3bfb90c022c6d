"""Frame time against the batch size around the round boundary of the batch-GEMM kernels (round 6, verdict item 3):
B = 65536 + 512 k, k = 0..8, with the tail rows on the batch kernels (PERCEPNET_TAIL_ROWS=0: the staircase) and on the
small-batch kernels (PERCEPNET_TAIL_ROWS=<limit>).  One process per point (tools/frame_time.py): ms per frame without events."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
limits = sys.argv[1:] or ["0", "4096"]
for k in range(9):
    B = 65536 + 512 * k
    for lim in limits:
        env = dict(os.environ, PERCEPNET_TAIL_ROWS=lim)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "frame_time.py"), str(B), "30", "3"], env=env, capture_output=True, text=True, timeout=600)
        line = (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]
        print(f"tail_rows_limit={lim:>5s} linear_from_65536={B / 65536:.4f}x | {line}", flush=True)
