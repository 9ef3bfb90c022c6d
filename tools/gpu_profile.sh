#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench command, then the summaries.
#   tools/gpu_profile.sh <tag> [suffix] [extra bench args]     suffix: "" (headline), _1024, _fp16, _x3 -> <tag>_kernel_stats<suffix>.csv ...
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}; shift
SUF=""; case "$1" in _*) SUF=$1; shift;; esac
OUT=$R/gpurun_out/prof$SUF
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# STEADY STATE: every pass runs W warm frames from the zero state before the K frames whose launches are summarised (the history
# ring holds 12 frames; until it has filled the pitch search returns T = 768 and the comb filter reads its largest window)
W=${PN_PROF_WARMUP:-24}; K=${PN_PROF_STEPS:-5}
B="python $R/bench.py --no-cpu-baseline --no-parity --no-sustained --no-other-configs --no-realtime $@"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B --steps $K --warmup $W > $OUT/stats_bench.json 2> $OUT/stats.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B --steps $K --warmup $W --no-profile > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B --steps $K --warmup $W --no-profile > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- $B --steps $K --warmup $W --no-profile > /dev/null 2> $OUT/pmc_sq.err
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -o bench -- $B --steps $K --warmup $W --no-profile > /dev/null 2> $OUT/pmc_sq2.err
python $R/tools/summarize_prof.py $TAG --suffix "$SUF" --src $OUT --out $R/gpurun_out/prof_summary --frames-total $((${PN_PROF_PRIME:-12} + W + K)) --frames-keep $K
# the raw kernel trace of a full run is large: keep only the summaries
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*.db" -delete
