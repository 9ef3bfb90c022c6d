// Per-stream lifecycle of a batched context (round-3 verdict item 4): what rnnoise_init does for ONE DenoiseState
// (reference denoise.cpp:259-280: memset of the struct, calloc of the RNN state) for a chosen subset of the B streams of a
// context, on the device, while the other streams keep running.  Every per-stream buffer of a context is a ring or a
// ping-pong pair indexed by the context's GLOBAL frame counter (hist slot t % 12, look-ahead rings t % 6, conv rings
// tn % 5 / tn % 3, GRU buffers tn & 1); a fresh stream is all-zero in every slot, so zeroing the rows of stream s in every
// slot puts that stream at its own frame 0 whatever the phase of the rings is (the look-ahead spectrum of an all-zero
// window is zero, its band energies are zero: the same values a fresh context holds).
#include "pn_common.h"

// rows ids[0..n) of a [n_slots][n_rows][row_floats] array (slot stride in floats): one block per (row, slot), float4 stores
__global__ __launch_bounds__(256) void pn_zero_rows_kernel(float *__restrict__ base, int row_floats, long long row_stride,
                                                           long long slot_stride, const int *__restrict__ ids) {
  float *row = base + (size_t)blockIdx.y * slot_stride + (size_t)ids[blockIdx.x] * row_stride;
  if ((row_floats & 3) == 0 && (row_stride & 3) == 0 && (slot_stride & 3) == 0) {
    for (int i = threadIdx.x; i < (row_floats >> 2); i += blockDim.x) reinterpret_cast<float4 *>(row)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int i = threadIdx.x; i < row_floats; i += blockDim.x) row[i] = 0.f;
  }
}
void pn_launch_zero_rows(hipStream_t st, void *base, int row_floats, long long row_stride, int n_slots, long long slot_stride,
                         const int *d_ids, int n) {
  if (n <= 0 || !base) return;
  hipLaunchKernelGGL(pn_zero_rows_kernel, dim3(n, n_slots), dim3(row_floats >= 1024 ? 256 : 64), 0, st, (float *)base, row_floats,
                     row_stride, slot_stride, d_ids);
}

// the same rows of a fragment-order operand shadow (pn_nn_x3.hip): [M tile of 128][column tile of 32][plane][k-group of 4]
// [row 0..127][8 halfs]; width = logical columns, np = planes; one thread per (column tile, plane, k-group) of a row
__global__ __launch_bounds__(64) void pn_zero_shadow_rows_kernel(uint4 *__restrict__ S, int n_ct, int np, long long slot_stride_u4,
                                                                const int *__restrict__ ids) {
  const int r = ids[blockIdx.x], per_row = n_ct * np * 4;
  uint4 *slot = S + (size_t)blockIdx.y * slot_stride_u4 + (size_t)(r >> 7) * n_ct * np * 512;
  for (int i = threadIdx.x; i < per_row; i += blockDim.x) {
    const int kg = i & 3, pl = (i >> 2) % np, ct = (i >> 2) / np;
    slot[((size_t)ct * np + pl) * 512 + kg * 128 + (r & 127)] = make_uint4(0, 0, 0, 0);
  }
}
void pn_launch_zero_shadow_rows(hipStream_t st, void *S, int width, int np, int n_slots, long long slot_stride_halfs,
                                const int *d_ids, int n) {
  if (n <= 0 || !S) return;
  hipLaunchKernelGGL(pn_zero_shadow_rows_kernel, dim3(n, n_slots), dim3(64), 0, st, (uint4 *)S, width / 32, np, slot_stride_halfs / 8, d_ids);
}
