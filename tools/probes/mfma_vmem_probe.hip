// Micro-probe (tuning aid, not part of the library): does a wave issuing global loads slow down the MFMA stream
// of ANOTHER wave on the same SIMD?  Block = 8 waves on one CU (LDS-padded to one block per CU): waves 0-3 issue
// v_mfma_f32_32x32x2_f32 back to back (3 independent accumulators), waves 4-7 (same SIMDs) do `mode`:
//   0 nothing, 1 global_load_dwordx4 at ~7 per 3072 cycles, 2 the same number of ds_read_b128,
//   3 loads issued by the MFMA waves themselves (7 per 48 MFMAs, consumed after the 48), 4 loads as fast as possible,
//   5 wave-specialised GEMM skeleton: MFMA waves also issue 16 ds_read_b128 per 48 MFMAs and one s_barrier per 48
//     MFMAs; companions issue 7 loads + 7 ds_write_b128 per barrier
//   6 as 5 without the barrier
// Prints MFMA cycles per instruction as seen by the MFMA waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define SB() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(512) void probe(const float *__restrict__ src, float *__restrict__ out, long long *cyc,
                                             int iters, int mode) {
  __shared__ float pad[30000];            // 120 KB: one block per CU
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  pad[tid] = tid;
  __syncthreads();
  const float4 *p = reinterpret_cast<const float4 *>(src) + (size_t)(blockIdx.x & 7) * 65536 + tid;   // small, L2-resident footprint (the real kernel re-uses panels across blocks)
  float4 sink = make_float4(0, 0, 0, 0);
  if (wave < 4) {
    floatx16 a0, a1, a2;
    for (int i = 0; i < 16; i++) { a0[i] = lane; a1[i] = 1; a2[i] = 2; }
    const float x = lane * 0.001f, y = 1.0f;
    float4 ld[7], rd[16];
    for (int q = 0; q < 7; q++) ld[q] = make_float4(0, 0, 0, 0);
    for (int q = 0; q < 16; q++) rd[q] = make_float4(0, 0, 0, 0);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        SB();
        if (mode == 3 && q < 7) { ld[q] = p[(size_t)((it * 8 + q) & 63) * 512]; SB(); }
        if (mode == 5 || mode == 6 || mode == 8) { rd[2 * q] = *reinterpret_cast<const float4 *>(&pad[(lane * 4 + q * 512) % 29000]);
                         rd[2 * q + 1] = *reinterpret_cast<const float4 *>(&pad[(lane * 4 + q * 512 + 256) % 29000]); SB(); }
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        SB();
      }
      if (mode == 3) { for (int q = 0; q < 7; q++) sink.x += ld[q].x; }
      if (mode == 5 || mode == 6 || mode == 8) { for (int q = 0; q < 16; q++) sink.y += rd[q].x; }
      if (mode == 5 || mode == 7 || mode == 8) { SB(); __syncthreads(); SB(); }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    float s = sink.x;
    for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
  } else {
    float4 va[7], vb[7], vc[7];
    for (int q = 0; q < 7; q++) vc[q] = make_float4(0, 0, 0, 0);
    // companion waves.  mode >= 5: loads run two barriers ahead of their ds_write (three register sets, rotated
    // by unrolling so that no copy waits for data in flight)
    if (mode == 7 || mode == 8) {
      for (int it = 0; it < iters; it++) __syncthreads();
    } else if (mode >= 5) {
#define WS_STEP(LD, ST, itx) do {                                                                          \
        _Pragma("unroll") for (int q = 0; q < 7; q++) LD[q] = p[(size_t)(((itx) * 7 + 14 + q) & 63) * 512]; \
        SB();                                                                                              \
        _Pragma("unroll") for (int q = 0; q < 7; q++)                                                      \
          *reinterpret_cast<float4 *>(&pad[1024 + (tid - 256) * 4 + q * 1024]) = ST[q];                    \
        SB();                                                                                              \
        if (mode == 5) __syncthreads(); else __builtin_amdgcn_s_sleep(30);                                 \
      } while (0)
#pragma unroll
      for (int q = 0; q < 7; q++) { va[q] = p[(size_t)(q & 63) * 512]; vb[q] = p[(size_t)((7 + q) & 63) * 512]; }
      for (int it = 0; it + 2 < iters; it += 3) {
        WS_STEP(vc, va, it); WS_STEP(va, vb, it + 1); WS_STEP(vb, vc, it + 2);
      }
      sink.x += va[0].x + vb[0].x + vc[0].x;
    } else
    for (int it = 0; it < iters; it++) {
      if (mode == 1) {
#pragma unroll
        for (int q = 0; q < 7; q++) {
          const float4 v = p[(size_t)((it * 7 + q) & 63) * 512]; sink.x += v.x; SB();
          __builtin_amdgcn_s_sleep(6);     // ~384 cycles
        }
      } else if (mode == 2) {
#pragma unroll
        for (int q = 0; q < 7; q++) {
          const float4 v = *reinterpret_cast<const float4 *>(&pad[(lane * 4 + q * 256) % 29000]); sink.x += v.x; SB();
          __builtin_amdgcn_s_sleep(6);
        }
      } else if (mode == 4) {
#pragma unroll
        for (int q = 0; q < 28; q++) { const float4 v = p[(size_t)((it * 28 + q) & 63) * 512]; sink.x += v.x; SB(); }
      } else {
        __builtin_amdgcn_s_sleep(40);
      }
    }
    out[(size_t)blockIdx.x * 512 + tid] = sink.x;
  }
}

int main(int argc, char **argv) {
  const int iters = (argc > 1 ? atoi(argv[1]) : 2001) / 3 * 3, nblk = 256;
  float *src, *out; long long *cyc;
  hipMalloc(&src, (size_t)nblk * 65536 * 16 + (1 << 20)); hipMemset(src, 0, (size_t)nblk * 65536 * 16 + (1 << 20));
  hipMalloc(&out, (size_t)nblk * 512 * 4); hipMalloc(&cyc, nblk * 4 * 8);
  long long *h = (long long *)malloc(nblk * 4 * 8);
  const char *names[9] = {"companion idle", "companion: 7 global loads / 3072 cyc", "companion: 7 ds_reads / 3072 cyc",
                          "loads issued by the MFMA waves", "companion: loads back to back",
                          "WS skeleton: ds_reads + barrier in MFMA waves, loads + ds_writes in companions", "WS skeleton without barrier",
                          "MFMA + barrier per 48, companions barrier only", "MFMA + ds_reads + barrier per 48, companions barrier only"};
  for (int mode = 0; mode < 9; mode++) {
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), 0, 0, src, out, cyc, iters, mode);   // warm-up
    hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), 0, 0, src, out, cyc, iters, mode);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, nblk * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nblk * 4; i++) s += (double)h[i];
    printf("mode %d (%s): %.1f cycles per MFMA\n", mode, names[mode], s / (nblk * 4) / ((double)iters * 48));
  }
  return 0;
}
