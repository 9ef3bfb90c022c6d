// Reproducer for DESIGN.md §4.3 ("MFMA result hazard"): a K loop of v_mfma_f32_32x32x2_f32 whose exit block reads the
// accumulators at once.  Build:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DPAD=n] [-DSB] -o mfma_exit_hazard
//                                tools/probes/mfma_exit_hazard.hip        (add -S --cuda-device-only for the ISA)
// Run: prints, per accumulator element, how many of the 64x... outputs differ from the exact fmaf chain computed by a
// VALU loop in the same kernel (MFMA fp32 == fmaf chain bit for bit, so any difference is a lost update).
//   -DPAD=n   n explicit wait states (s_nop) between the loop and the first accumulator read
//   -DSB      pin the loop body order with sched_barrier(0) as the production kernels do
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#ifndef PAD
#define PAD 0
#endif
#define STR2(x) #x
#define STR(x) STR2(x)

__global__ __launch_bounds__(256) void probe(const float *__restrict__ A, const float *__restrict__ B,
                                             float *__restrict__ out, float *__restrict__ ref, int nk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *a = A + ((size_t)blockIdx.x * 4 + wave) * nk * 64, *b = B + (size_t)wave * nk * 64 * 3;
  floatx16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = 0.25f * t;
#pragma unroll 1
  for (int k = 0; k < nk; k++) {
    const float av = a[k * 64 + lane];
    const float b0 = b[(k * 3 + 0) * 64 + lane], b1 = b[(k * 3 + 1) * 64 + lane], b2 = b[(k * 3 + 2) * 64 + lane];
#ifdef SB
    __builtin_amdgcn_sched_barrier(0);
#endif
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2, acc[2], 0, 0, 0);
#ifdef SB
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
#if PAD > 0
  asm volatile("s_nop " STR(PAD - 1) ::: "memory");
#endif
  // exit block: read every accumulator element at once (element 15 of the LAST MFMA's tuple first, like hipcc's copies)
  const size_t o = (((size_t)blockIdx.x * 4 + wave) * 64 + lane) * 48;
#pragma unroll
  for (int t = 2; t >= 0; t--)
#pragma unroll
    for (int i = 15; i >= 0; i--) out[o + t * 16 + i] = acc[t][i];
  // reference: the same k-ordered fmaf chains on the VALU.  D[i][j] lives in lane (j, i-half): row = (r&3)+8(r>>2)+4(lane>>5)
#pragma unroll 1
  for (int t = 0; t < 3; t++)
#pragma unroll 1
    for (int r = 0; r < 16; r++) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
      float s = 0.25f * t;
      for (int k = 0; k < nk; k++)
        for (int kk = 0; kk < 2; kk++)
          s = __builtin_fmaf(a[k * 64 + kk * 32 + row], b[(k * 3 + t) * 64 + kk * 32 + col], s);
      ref[o + t * 16 + r] = s;
    }
}

int main(int argc, char **argv) {
  const int nk = argc > 1 ? atoi(argv[1]) : 64, nb = argc > 2 ? atoi(argv[2]) : 2048;
  std::vector<float> hA((size_t)nb * 4 * nk * 64), hB((size_t)4 * nk * 64 * 3);
  srand(1);
  for (auto &v : hA) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto &v : hB) v = (rand() % 2001 - 1000) / 1000.f;
  float *A, *B, *out, *ref;
  const size_t no = (size_t)nb * 4 * 64 * 48;
  hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&out, no * 4); hipMalloc(&ref, no * 4);
  hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, A, B, out, ref, nk);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  std::vector<float> ho(no), hr(no);
  hipMemcpy(ho.data(), out, no * 4, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), ref, no * 4, hipMemcpyDeviceToHost);
  long bad[48] = {0}, total = 0;
  for (size_t i = 0; i < no; i++) if (ho[i] != hr[i]) { bad[i % 48]++; total++; }
  printf("PAD=%d nk=%d blocks=%d: %ld of %zu outputs differ from the fmaf chain\n", PAD, nk, nb, total, no);
  for (int t = 0; t < 3; t++) { printf("  acc[%d] per element:", t); for (int i = 0; i < 16; i++) printf(" %ld", bad[t * 16 + i]); printf("\n"); }
  return total ? 1 : 0;
}
