#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> $OUT/pmc_sq.err
find $OUT -name "*.csv" | head -40
ls -la $OUT/*
# keep only small summaries (kernel traces of full runs can be large)
find $OUT -name "*kernel_trace.csv" -size +8M -delete
