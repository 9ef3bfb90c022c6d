#!/bin/bash
# feasibility of running the front end of one frame beside the network of another on the same CUs
V=$PWD/percepnet_amd/lib/variants
tools/gpu_variants.sh oneblk fe8 2>&1 | cut -c1-200
python tools/two_ctx.py 65536 1 8 2>&1 | grep -v amdgpu.ids
python tools/two_ctx.py 32768 2 8 2>&1 | grep -v amdgpu.ids
for lib in co co_nopad; do
  for skew in 0 5000000 12000000; do
    PERCEPNET_LIB=$V/$lib/libpercepnet_hip.so PN_SKEW_CYCLES=$skew python tools/two_ctx.py 32768 2 8 2>&1 | grep -v amdgpu.ids
  done
done
