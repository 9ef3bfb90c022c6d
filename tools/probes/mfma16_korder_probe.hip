// Micro-probe: is v_mfma_f32_16x16x4_f32 the k-ASCENDING fused-multiply-add chain  d = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c))))
// bit for bit, as v_mfma_f32_32x32x2_f32 is for its two k values?  The network kernels may only use it for long-K, narrow-N
// layers (fc_gb: K = 2560, N = 34) if every output stays the reference's k-ordered chain (nnet.cpp:59-72).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *A, const float *B, const float *C, float *D, int ksteps) {
  const int l = threadIdx.x;
  floatx4 acc;
  for (int r = 0; r < 4; r++) acc[r] = C[(4 * (l / 16) + r) * 16 + (l % 16)];
  for (int s = 0; s < ksteps; s++) {
    const float a = A[(l % 16) * (4 * ksteps) + 4 * s + l / 16];      // A[i][k], i = l % 16, k = 4s + l / 16
    const float b = B[(4 * s + l / 16) * 16 + (l % 16)];               // B[k][j], j = l % 16
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) D[(4 * (l / 16) + r) * 16 + (l % 16)] = acc[r];
}
int main() {
  const int KS = 8, K = 4 * KS;
  float hA[16 * K], hB[K * 16], hC[256], hD[256];
  unsigned x = 12345;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((int)(x >> 8) % 20001 - 10000) * 1.37e-4f; };
  for (float &v : hA) v = rnd(); for (float &v : hB) v = rnd(); for (float &v : hC) v = rnd() * 3;
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, KS);
  if (hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
  int asc = 0, desc = 0, pair = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float c1 = hC[i * 16 + j], c2 = c1, c3 = c1;
      for (int s = 0; s < KS; s++) {
        for (int q = 0; q < 4; q++) c1 = fmaf(hA[i * K + 4 * s + q], hB[(4 * s + q) * 16 + j], c1);
        for (int q = 3; q >= 0; q--) c2 = fmaf(hA[i * K + 4 * s + q], hB[(4 * s + q) * 16 + j], c2);
        // pairwise: (a0b0 + a1b1) + (a2b2 + a3b3) + c in some fused form — only reported as "neither"
        c3 = fmaf(hA[i * K + 4 * s + 3], hB[(4 * s + 3) * 16 + j], fmaf(hA[i * K + 4 * s + 1], hB[(4 * s + 1) * 16 + j], fmaf(hA[i * K + 4 * s + 2], hB[(4 * s + 2) * 16 + j], fmaf(hA[i * K + 4 * s], hB[(4 * s) * 16 + j], c3))));
      }
      asc += memcmp(&c1, &hD[i * 16 + j], 4) == 0; desc += memcmp(&c2, &hD[i * 16 + j], 4) == 0; pair += memcmp(&c3, &hD[i * 16 + j], 4) == 0;
    }
  printf("v_mfma_f32_16x16x4_f32 over %d k-steps: %d/256 outputs equal the k-ASCENDING fmaf chain, %d/256 the descending one, %d/256 the 0,2,1,3 order\n", KS, asc, desc, pair);
  return 0;
}
