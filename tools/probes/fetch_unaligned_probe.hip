// Calibration of rocprofv3's FETCH_SIZE for the comb filter's access pattern (round 5): one wave per "stream" reads 7 taps x
// 4 quarters of 960 bytes with one UNALIGNED global_load_dwordx4 per lane (60 lanes), exactly like pn_fe_spec_out_kernel.
//   mode 0: the 28 windows are disjoint (28 x 960 B of distinct data per stream)         -> bytes known: 26 880 per stream
//   mode 1: the comb's overlapping windows at period T (union = (960 + 6 T) x 4 B), tap-major order
//   mode 2: the same windows, quarter-major halves (the round-3/4 order)
//   mode 3: aligned dwordx4 loads of 26 880 contiguous bytes per stream (the guide's calibrated case: FETCH_SIZE x 2 = bytes)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_unaligned_probe fetch_unaligned_probe.hip
// Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc TCC_HIT_sum TCC_MISS_sum`; the kernel name carries the mode.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define ROW 8192            // floats per stream (32 KB: > the largest union window, 5568)
template <int MODE>
__global__ __launch_bounds__(256) void fetch_probe(const float *__restrict__ buf, float *__restrict__ out, int n_streams, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lc = lane < 60 ? lane : 59;
  float acc = 0;
  for (int s = blockIdx.x * 4 + wave; s < n_streams; s += gridDim.x * 4) {
    const float *h = buf + (size_t)s * ROW + 1;          // + 1: every window is 4-byte but not 16-byte aligned
    if (MODE == 3) {
      const float *a = buf + (size_t)s * ROW;
      for (int i = 0; i < 28; i++) { const float4 v = *reinterpret_cast<const float4 *>(a + 240 * i + 4 * lc); acc += v.x + v.w; }
    } else if (MODE == 0) {
      for (int i = 0; i < 28; i++) { const f4u v = *reinterpret_cast<const f4u *>(h + 240 * i + 4 * lc); acc += v.x + v.w; }
    } else if (MODE == 1) {
      for (int k = 0; k < 7; k++)
        for (int q = 0; q < 4; q++) { const f4u v = *reinterpret_cast<const f4u *>(h + T * (6 - k) + 240 * q + 4 * lc); acc += v.x + v.w; }
    } else {
      for (int k0 = 0; k0 < 4; k0 += 2)
        for (int q = 0; q < 2; q++)
          for (int k = 0; k < 7; k++) { const f4u v = *reinterpret_cast<const f4u *>(h + T * (6 - k) + 240 * (k0 + q) + 4 * lc); acc += v.x + v.w; }
    }
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 65536, T = argc > 2 ? atoi(argv[2]) : 418;
  float *buf, *out;
  if (hipMalloc(&buf, (size_t)n * ROW * 4 + 65536) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, (size_t)n * ROW * 4 + 65536);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(fetch_probe<3>, dim3(768), dim3(256), 0, 0, buf, out, n, T);
    hipLaunchKernelGGL(fetch_probe<0>, dim3(768), dim3(256), 0, 0, buf, out, n, T);
    hipLaunchKernelGGL(fetch_probe<1>, dim3(768), dim3(256), 0, 0, buf, out, n, T);
    hipLaunchKernelGGL(fetch_probe<2>, dim3(768), dim3(256), 0, 0, buf, out, n, T);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  printf("fetch_unaligned_probe: %d streams, T = %d: distinct bytes per stream: mode 3 / 0: %d, modes 1 / 2 (union window): %d\n", n, T, 28 * 960, (960 + 6 * T) * 4);
  return 0;
}
