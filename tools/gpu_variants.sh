#!/bin/bash
# kernel times for the default library and each named variant (lib/variants/<name>)
python tools/kernel_times.py 65536 8 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  PERCEPNET_LIB=$PWD/percepnet_amd/lib/variants/$v/libpercepnet_hip.so python tools/kernel_times.py 65536 8 2>&1 | grep -v amdgpu.ids
done
