// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 subnormal inputs?  (The split-precision GEMM mode stores the low
// halves of its operands unscaled; most of them are fp16 subnormals.)  Prints the accumulated value for inputs of
// 2^-20 (subnormal) x 1.0 summed over K = 16: 1.52587890625e-05 if honoured, 0 if flushed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float av, float bv) {
  half8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
  floatx16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
}
int main() {
  float *d; hipMalloc(&d, 64 * 4);
  const float cases[][2] = {{9.5367431640625e-07f, 1.f}, {1.f, 9.5367431640625e-07f}, {9.5367431640625e-07f, 9.5367431640625e-07f},
                            {5.9604644775390625e-08f, 1.f}, {6.103515625e-05f, 6.103515625e-05f}};
  for (auto &cs : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, cs[0], cs[1]);
    float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a=%.10g b=%.10g -> %.10g (expected %.10g)\n", cs[0], cs[1], h[0], 16.0 * (double)cs[0] * (double)cs[1]);
  }
  return 0;
}
