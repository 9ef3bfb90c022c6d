#!/bin/bash
# front-end tuning: parity subset first, then kernel times for each variant library
python -m pytest tests/test_gpu_parity.py -x -q -k "strict_mode_bit_exact or features_and_silence or golden_vectors_strict" 2>&1 | tail -2
python tools/kernel_times.py 65536 8 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  PERCEPNET_LIB=$PWD/percepnet_amd/lib/variants/$v/libpercepnet_hip.so python tools/kernel_times.py 65536 8 2>&1 | grep -v amdgpu.ids
done
