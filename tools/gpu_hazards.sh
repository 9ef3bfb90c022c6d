#!/bin/bash
# NOTES_history.md (round-4 DESIGN §4.3/§4.4) evidence: (1) how many wait states gfx950 really needs after v_mfma_f32_32x32x2_f32, (2) the loop-exit
# reproducer as hipcc compiles it, (3) the parity / placement tests on builds WITHOUT the manual MFMA drain and WITHOUT the
# DSP loop back-edge padding.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
{ echo "== mfma_waitstate_probe"; timeout 120 tools/probes/mfma_waitstate_probe
  echo "== mfma_exit_hazard (hipcc's own padding, no sched_barrier)"; timeout 120 tools/probes/mfma_exit_hazard 64 2048
  echo "== mfma_exit_hazard_sb (loop body pinned with sched_barrier)"; timeout 120 tools/probes/mfma_exit_hazard_sb 64 2048
  echo "== mfma_exit_hazard nk=1"; timeout 120 tools/probes/mfma_exit_hazard 1 2048; timeout 120 tools/probes/mfma_exit_hazard_sb 1 2048
} > $O/hazard_probes.log 2>&1
for v in nopad nodrain nopad_nodrain; do
  echo "== variant $v" >> $O/hazard_variants.log
  PERCEPNET_LIB=$R/percepnet_amd/lib/variants/$v/libpercepnet_hip.so timeout 900 python -m pytest -q -x -m gpu \
     tests/test_gpu_longrun.py "tests/test_gpu_parity.py::test_placement_invariance_across_grid_stride_rounds" \
     "tests/test_gpu_parity.py::test_full_size_properties_65536_streams" "tests/test_gpu_parity.py::test_strict_mode_bit_exact_pcm_and_taps" \
     "tests/test_gpu_parity.py::test_ragged_batch_sizes" 2>&1 | tail -4 >> $O/hazard_variants.log
done
cat $O/hazard_probes.log; cat $O/hazard_variants.log
