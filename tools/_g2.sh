cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stages.py "tests/test_gpu_parity.py::test_tail_rows_on_the_small_kernels_bit_identical" tests/test_gpu_bench.py -m gpu -x -q > $O/pytest_r06b.log 2>&1; echo "rc=$?" >> $O/pytest_r06b.log; tail -5 $O/pytest_r06b.log
timeout 1200 python tools/tail_sweep.py 0 4096 > $O/r06_tail_sweep.log 2>&1; cat $O/r06_tail_sweep.log
