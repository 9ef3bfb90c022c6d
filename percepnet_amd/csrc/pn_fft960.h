// 960-point forward FFT of the DSP kernels as THREE register-fused passes (gfx950, one stream per wavefront) — shared by
// the spectral front-end kernels (pn_dsp_fe_split_s.hip) and the back end (pn_dsp.hip).
//   P1  lane l < 60 forms its inputs 4l + c + 240k (c, k < 4) in registers and runs the first radix-4 stage
//       (kiss_fft.cpp:112-131) on them; there is no digit-reversal scatter: butterfly n = 4l + c lands on b(n)
//   P2  lane l < 60: radix-4 stages m=4 and m=16 (139-166) on the 16 elements 64*blk + 16a + 4a' + j', in place
//   P3  lane u < 64: radix-3 (196-227) and radix-5 (259-304) on the 15 elements u + 64b + 192c; results stay in registers
// LDS layout phi(i) = i + 4*(i >> 6) (float2 units, 1020 float2 per stream): P2 and P3 are bank-conflict-free.
// tools/fft960_model.py proves bit for bit that the three passes equal the five in-place stages of kiss_fft / fe_fft960.
// Same butterfly arithmetic, same order, separate IEEE roundings (-ffp-contract=off).
#pragma once
#include "pn_common.h"
#ifndef PN_WAVE_SYNC
#define PN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#ifndef CMUL
#define CMUL(m, a, b) do { (m).x = (a).x*(b).x - (a).y*(b).y; (m).y = (a).x*(b).y + (a).y*(b).x; } while (0)
#endif
#define FS_NF 1020                      // float2 per stream: 960 + 4 per 64 (layout phi)
#define FS_PHI(i) ((i) + 4 * ((i) >> 6))

__device__ __forceinline__ float2 fs_tw(const PnTables *__restrict__ T, int i) { return make_float2(T->tw[2 * i], T->tw[2 * i + 1]); }

// radix-4 butterfly without twiddles (m = 1, kiss_fft.cpp:112-131), in place on f[0..3]
__device__ __forceinline__ void fs_bfly4_m1(float2 *f) {
  float2 f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], s0, s1;
  s0.x = f0.x - f2.x; s0.y = f0.y - f2.y;
  f0.x += f2.x; f0.y += f2.y;
  s1.x = f1.x + f3.x; s1.y = f1.y + f3.y;
  f2.x = f0.x - s1.x; f2.y = f0.y - s1.y;
  f0.x += s1.x; f0.y += s1.y;
  s1.x = f1.x - f3.x; s1.y = f1.y - f3.y;
  f1.x = s0.x + s1.y; f1.y = s0.y - s1.x;
  f3.x = s0.x - s1.y; f3.y = s0.y + s1.x;
  f[0] = f0; f[1] = f1; f[2] = f2; f[3] = f3;
}
// radix-4 butterfly with twiddles (kiss_fft.cpp:139-166) on four elements held with stride `st` in a register array
template <int st>
__device__ __forceinline__ void fs_bfly4(float2 *f, float2 t1, float2 t2, float2 t3) {
  float2 f0 = f[0], fm = f[st], f2m = f[2 * st], f3m = f[3 * st];
  float2 s0, s1, s2, s3, s4, s5;
  CMUL(s0, fm, t1); CMUL(s1, f2m, t2); CMUL(s2, f3m, t3);
  s5.x = f0.x - s1.x; s5.y = f0.y - s1.y;
  f0.x += s1.x; f0.y += s1.y;
  s3.x = s0.x + s2.x; s3.y = s0.y + s2.y;
  s4.x = s0.x - s2.x; s4.y = s0.y - s2.y;
  f2m.x = f0.x - s3.x; f2m.y = f0.y - s3.y;
  f0.x += s3.x; f0.y += s3.y;
  fm.x = s5.x + s4.y; fm.y = s5.y - s4.x;
  f3m.x = s5.x - s4.y; f3m.y = s5.y + s4.x;
  f[0] = f0; f[st] = fm; f[2 * st] = f2m; f[3 * st] = f3m;
}

// Per-lane constants of the FFT that are worth a register each for the whole kernel (the 29 twiddles of P2/P3 are
// fetched per transform from the 7.7 KB L1-resident table instead).
struct FsLane {
  int p1off[4];         // float2 index phi(4 b(n)) of stage-1 butterfly n = 4l + c  (b(n): digit reversal, kiss_fft.cpp:315-345)
};
__device__ __forceinline__ void fs_lane_init(FsLane &Z, const PnTables *__restrict__ T, int l) {
  const int lc = l < 60 ? l : 59;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int n = 4 * lc + c;
    const int n0 = n % 5, n1 = (n / 5) % 3, n2 = (n / 15) % 4, n3 = n / 60;
    const int b = 48 * n0 + 16 * n1 + 4 * n2 + n3;
    Z.p1off[c] = FS_PHI(4 * b);
  }
}

// P1: x[k] = the four float4 of samples 4l + 240k (k = 0..3) of the 960-sample frame, lane l < 60.  Window, 1/960 scale,
// first radix-4 stage, results to LDS (two 16-byte stores per butterfly).
__device__ __forceinline__ void fs_fft_p1(float2 *F, const float *win, const FsLane &Z, const float4 *x, int l) {
  const float scale = 1.f / PN_NFFT;
  if (l < 60) {
    // window weight of sample ii = 4l + c + 240k (apply_window, denoise.cpp:282-289): win[ii] for ii < 480, else
    // win[959 - ii] — for k = 2, 3 the four weights are the float4 at 476 - 4l - 240(k-2), components reversed
    float wq[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float4 q = *reinterpret_cast<const float4 *>(win + (k < 2 ? 4 * l + 240 * k : 476 - 4 * l - 240 * (k - 2)));
      if (k < 2) { wq[k][0] = q.x; wq[k][1] = q.y; wq[k][2] = q.z; wq[k][3] = q.w; }
      else { wq[k][0] = q.w; wq[k][1] = q.z; wq[k][2] = q.y; wq[k][3] = q.x; }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float2 f[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float v = c == 0 ? x[k].x : (c == 1 ? x[k].y : (c == 2 ? x[k].z : x[k].w));
        f[k] = make_float2(scale * (v * wq[k][c]), scale * 0.f);
      }
      fs_bfly4_m1(f);
      float4 *dst = reinterpret_cast<float4 *>(F + Z.p1off[c]);
      dst[0] = make_float4(f[0].x, f[0].y, f[1].x, f[1].y);
      dst[1] = make_float4(f[2].x, f[2].y, f[3].x, f[3].y);
    }
  }
}

// P2 + P3.  On return w[b][c] = output u + 64b + 192c (u = lane).  ALL = false (the analysis side: bins >= 400 are never
// used, denoise.cpp:89-182): only c = 0, 1 (all b) and w[0][2] (bin 384 + u, meaningful for u < 16) are formed.
template <bool ALL>
__device__ __forceinline__ void fs_fft_p23(float2 *F, const PnTables *__restrict__ T, int l_in, float2 (&w)[3][5]) {
  // The twiddle indices depend on the lane only, i.e. the 29 loads are invariant in the caller's stream loop: without
  // this opaque copy they are hoisted out of it and held in 58 registers for the whole kernel (spills at 128).
  int l = l_in;
  asm volatile("" : "+v"(l));
  // twiddles of P2: issued first, the table is L1/L2-resident
  const int jp = l & 3, blk = (l < 60 ? l : 59) >> 2;
  float2 t2[3], t3[4][3];
#pragma unroll
  for (int q = 1; q <= 3; q++) t2[q - 1] = fs_tw(T, jp * 60 * q);
#pragma unroll
  for (int ap = 0; ap < 4; ap++)
#pragma unroll
    for (int q = 1; q <= 3; q++) t3[ap][q - 1] = fs_tw(T, 15 * (4 * ap + jp) * q);
  PN_WAVE_SYNC();
  if (l < 60) {
    float2 *f = F + 68 * blk + jp;                 // phi(64 blk + jp)
    float2 v[16];                                  // v[4a + a'] = element 64 blk + 16a + 4a' + jp
#pragma unroll
    for (int e = 0; e < 16; e++) v[e] = f[4 * e];
#pragma unroll
    for (int a = 0; a < 4; a++) fs_bfly4<1>(v + 4 * a, t2[0], t2[1], t2[2]);          // m = 4: over a'
#pragma unroll
    for (int ap = 0; ap < 4; ap++) fs_bfly4<4>(v + ap, t3[ap][0], t3[ap][1], t3[ap][2]);   // m = 16: over a, j = 4a' + jp
#pragma unroll
    for (int e = 0; e < 16; e++) f[4 * e] = v[e];
  }
  float2 r3a = fs_tw(T, 5 * l), r3b = fs_tw(T, 10 * l), t5[3][4];
#pragma unroll
  for (int b = 0; b < 3; b++)
#pragma unroll
    for (int q = 1; q <= 4; q++) t5[b][q - 1] = fs_tw(T, q * (l + 64 * b));
  const float2 ya = fs_tw(T, 192), yb = fs_tw(T, 384);
  const float epi3 = T->tw[2 * 320 + 1];
  PN_WAVE_SYNC();
#pragma unroll
  for (int b = 0; b < 3; b++)
#pragma unroll
    for (int c = 0; c < 5; c++) w[b][c] = F[l + 68 * b + 204 * c];      // phi(u + 64b + 192c) = u + 68b + 204c
#pragma unroll
  for (int c = 0; c < 5; c++) {                    // radix-3, m = 64 (kiss_fft.cpp:196-227): over b
    float2 f0 = w[0][c], fm = w[1][c], f2m = w[2][c], s0, s1, s2, s3;
    CMUL(s1, fm, r3a); CMUL(s2, f2m, r3b);
    s3.x = s1.x + s2.x; s3.y = s1.y + s2.y;
    s0.x = s1.x - s2.x; s0.y = s1.y - s2.y;
    fm.x = f0.x - s3.x * .5f; fm.y = f0.y - s3.y * .5f;
    s0.x *= epi3; s0.y *= epi3;
    f0.x += s3.x; f0.y += s3.y;
    f2m.x = fm.x + s0.y; f2m.y = fm.y - s0.x;
    fm.x = fm.x - s0.y; fm.y = fm.y + s0.x;
    w[0][c] = f0; w[1][c] = fm; w[2][c] = f2m;
  }
#pragma unroll
  for (int b = 0; b < 3; b++) {                    // radix-5, m = 192 (kiss_fft.cpp:259-304): over c; outputs 0, 1 (and 2 for b = 0)
    float2 f0 = w[b][0], f1 = w[b][1], f2 = w[b][2], f3 = w[b][3], f4 = w[b][4];
    float2 s0 = f0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10;
    CMUL(s1, f1, t5[b][0]); CMUL(s2, f2, t5[b][1]); CMUL(s3, f3, t5[b][2]); CMUL(s4, f4, t5[b][3]);
    s7.x = s1.x + s4.x; s7.y = s1.y + s4.y;
    s10.x = s1.x - s4.x; s10.y = s1.y - s4.y;
    s8.x = s2.x + s3.x; s8.y = s2.y + s3.y;
    s9.x = s2.x - s3.x; s9.y = s2.y - s3.y;
    f0.x = f0.x + (s7.x + s8.x);
    f0.y = f0.y + (s7.y + s8.y);
    s5.x = s0.x + (s7.x * ya.x + s8.x * yb.x);
    s5.y = s0.y + (s7.y * ya.x + s8.y * yb.x);
    s6.x = s10.y * ya.y + s9.y * yb.y;
    s6.y = -(s10.x * ya.y + s9.x * yb.y);
    f1.x = s5.x - s6.x; f1.y = s5.y - s6.y;
    if (ALL) { f4.x = s5.x + s6.x; f4.y = s5.y + s6.y; }
    if (ALL || b == 0) {
      float2 s11, s12;
      s11.x = s0.x + (s7.x * yb.x + s8.x * ya.x);
      s11.y = s0.y + (s7.y * yb.x + s8.y * ya.x);
      s12.x = s9.y * ya.y - s10.y * yb.y;
      s12.y = s10.x * yb.y - s9.x * ya.y;
      f2.x = s11.x + s12.x; f2.y = s11.y + s12.y;
      if (ALL) { f3.x = s11.x - s12.x; f3.y = s11.y - s12.y; }
    }
    w[b][0] = f0; w[b][1] = f1;
    if (ALL || b == 0) w[b][2] = f2;
    if (ALL) { w[b][3] = f3; w[b][4] = f4; }
  }
}
