#!/bin/bash
# Can the DSP kernels of one half-batch run BESIDE the network GEMMs of the other on the same CUs?  
# default library: two 4-wave GEMM blocks fill every CU (486 of 512 registers per lane): nothing can co-reside.
# variant oneblk:   GEMM kernels capped at one block per CU (LDS pad): 81 KB of LDS + 264 registers per lane stay free, enough
#                   for one pitch-kernel block (80.9 KB, 256 registers) or two spectral blocks (37.8 KB, 128 registers).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$R/percepnet_amd/lib/variants
python tools/two_ctx.py 65536 1 8 2>&1 | grep -v amdgpu.ids
PERCEPNET_LIB=$V/oneblk/libpercepnet_hip.so python tools/two_ctx.py 65536 1 8 2>&1 | grep -v amdgpu.ids
for lib in default oneblk; do
  for skew in 0 4000000 9000000; do
    if [ $lib = default ]; then PN_SKEW_CYCLES=$skew python tools/two_ctx.py 32768 2 8 2>&1 | grep -v amdgpu.ids
    else PERCEPNET_LIB=$V/$lib/libpercepnet_hip.so PN_SKEW_CYCLES=$skew python tools/two_ctx.py 32768 2 8 2>&1 | grep -v amdgpu.ids; fi
  done
done
