// How many wait states does gfx950 REALLY need between v_mfma_f32_32x32x2_f32 and a VALU read of its last-written
// result register (element 15 of the tuple)?  LLVM's hazard recogniser inserts 18 for a 16-pass XDL op; DESIGN.md §4.3
// once saw a read only 2 states after the final MFMA return the pre-MFMA value.  This probe issues, entirely inside one
// asm statement (hipcc pads nothing there), a chain of 4 dependent MFMAs, then N wait states (s_nop), then reads a15
// (early), then drains the pipe and reads a15 again (final).  For each N it reports in how many lanes early != final.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_waitstate_probe tools/probes/mfma_waitstate_probe.hip && ./mfma_waitstate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define STR2(x) #x
#define STR(x) STR2(x)

// NOPS: literal asm lines producing exactly N wait states (empty for N = 0)
#define PROBE(NAME, NOPS)                                                                                   \
  __global__ void NAME(const float *__restrict__ in, float *__restrict__ out) {                             \
    const float a = in[threadIdx.x], b = in[64 + threadIdx.x];                                              \
    float early, fin;                                                                                       \
    asm volatile(                                                                                           \
        "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n"   \
        "v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"   \
        "v_accvgpr_write_b32 a8, 0\n v_accvgpr_write_b32 a9, 0\n v_accvgpr_write_b32 a10, 0\n v_accvgpr_write_b32 a11, 0\n" \
        "v_accvgpr_write_b32 a12, 0\n v_accvgpr_write_b32 a13, 0\n v_accvgpr_write_b32 a14, 0\n v_accvgpr_write_b32 a15, 0\n" \
        "s_nop 7\n"                                                                                         \
        "v_mfma_f32_32x32x2_f32 a[0:15], %2, %3, a[0:15]\n"                                                 \
        "v_mfma_f32_32x32x2_f32 a[0:15], %2, %3, a[0:15]\n"                                                 \
        "v_mfma_f32_32x32x2_f32 a[0:15], %2, %3, a[0:15]\n"                                                 \
        "v_mfma_f32_32x32x2_f32 a[0:15], %2, %3, a[0:15]\n"                                                 \
        NOPS                                                                                                \
        "v_accvgpr_read_b32 %0, a15\n"                                                                      \
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"                                            \
        "v_accvgpr_read_b32 %1, a15\n"                                                                      \
        : "=&v"(early), "=&v"(fin) : "v"(a), "v"(b)                                                         \
        : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "memory"); \
    out[threadIdx.x] = early; out[64 + threadIdx.x] = fin;                                                  \
  }
PROBE(p0, "")
PROBE(p1, "s_nop 0\n")
PROBE(p2, "s_nop 1\n")
PROBE(p4, "s_nop 3\n")
PROBE(p8, "s_nop 7\n")
PROBE(p10, "s_nop 9\n")
PROBE(p12, "s_nop 11\n")
PROBE(p13, "s_nop 12\n")
PROBE(p14, "s_nop 13\n")
PROBE(p15, "s_nop 14\n")
PROBE(p16, "s_nop 15\n")
PROBE(p17, "s_nop 15\n s_nop 0\n")
PROBE(p18, "s_nop 15\n s_nop 1\n")
PROBE(p20, "s_nop 15\n s_nop 3\n")

int main() {
  float h[128], *in, *out, r[128];
  for (int i = 0; i < 128; i++) h[i] = 0.5f + 0.01f * i;
  if (hipMalloc(&in, sizeof(h)) != hipSuccess || hipMalloc(&out, sizeof(h)) != hipSuccess) return 2;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  struct { int n; void (*k)(const float *, float *); } P[] = {{0, p0}, {1, p1}, {2, p2}, {4, p4}, {8, p8}, {10, p10}, {12, p12},
      {13, p13}, {14, p14}, {15, p15}, {16, p16}, {17, p17}, {18, p18}, {20, p20}};
  for (auto &p : P) {
    int bad = 0;
    for (int rep = 0; rep < 50; rep++) {
      hipLaunchKernelGGL(p.k, dim3(1), dim3(64), 0, 0, in, out);
      if (hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) return 2;
      for (int i = 0; i < 64; i++) bad += r[i] != r[64 + i];
    }
    printf("wait states %2d: early != final in %4d of 3200 lane-reads%s\n", p.n, bad, bad ? "" : "   <- safe");
  }
  return 0;
}
