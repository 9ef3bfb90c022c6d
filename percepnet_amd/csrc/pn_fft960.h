// 960-point complex FFT in LDS, shared by the front end (16 lanes per stream) and the back end (64 lanes per
// stream): opus_fft_impl (kiss_fft.cpp:518-564) for the factors 5,3,4,4,4, input already scaled by 1/960 and
// digit-reverse scattered (opus_fft_c 578-585).  Butterfly arithmetic is the reference's, operation for
// operation; the butterflies of a pass are independent, so they are spread over the LN lanes.
//
// Each lane handles its butterflies in batches of U: all LDS reads of the batch (operands and twiddles) are issued
// before the first butterfly is computed and all results are written afterwards.  Written butterfly by
// butterfly, the compiler must assume that the stores of one may alias the loads of the next (in-place FFT) and
// emits load -> wait -> compute -> store per butterfly, i.e. one exposed LDS latency each.
#pragma once
#include <hip/hip_runtime.h>

#ifndef PN_WAVE_SYNC
#define PN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#define PN_CMUL(m, a, b) do { (m).x = (a).x*(b).x - (a).y*(b).y; (m).y = (a).x*(b).y + (a).y*(b).x; } while (0)

// kf_bfly4, m == 1 (kiss_fft.cpp:112-131)
__device__ __forceinline__ void pn_bfly4_m1(float2 &f0, float2 &f1, float2 &f2, float2 &f3) {
  float2 s0, s1;
  s0.x = f0.x - f2.x; s0.y = f0.y - f2.y;
  f0.x += f2.x; f0.y += f2.y;
  s1.x = f1.x + f3.x; s1.y = f1.y + f3.y;
  f2.x = f0.x - s1.x; f2.y = f0.y - s1.y;
  f0.x += s1.x; f0.y += s1.y;
  s1.x = f1.x - f3.x; s1.y = f1.y - f3.y;
  f1.x = s0.x + s1.y; f1.y = s0.y - s1.x;
  f3.x = s0.x - s1.y; f3.y = s0.y + s1.x;
}
// kf_bfly4 (139-166)
__device__ __forceinline__ void pn_bfly4(float2 &f0, float2 &fm, float2 &f2m, float2 &f3m, const float2 &t1,
                                         const float2 &t2, const float2 &t3) {
  float2 s0, s1, s2, s3, s4, s5;
  PN_CMUL(s0, fm, t1); PN_CMUL(s1, f2m, t2); PN_CMUL(s2, f3m, t3);
  s5.x = f0.x - s1.x; s5.y = f0.y - s1.y;
  f0.x += s1.x; f0.y += s1.y;
  s3.x = s0.x + s2.x; s3.y = s0.y + s2.y;
  s4.x = s0.x - s2.x; s4.y = s0.y - s2.y;
  f2m.x = f0.x - s3.x; f2m.y = f0.y - s3.y;
  f0.x += s3.x; f0.y += s3.y;
  fm.x = s5.x + s4.y; fm.y = s5.y - s4.x;
  f3m.x = s5.x - s4.y; f3m.y = s5.y + s4.x;
}
// kf_bfly3 (196-227)
__device__ __forceinline__ void pn_bfly3(float2 &f0, float2 &fm, float2 &f2m, const float2 &t1, const float2 &t2,
                                         float epi3) {
  float2 s0, s1, s2, s3;
  PN_CMUL(s1, fm, t1); PN_CMUL(s2, f2m, t2);
  s3.x = s1.x + s2.x; s3.y = s1.y + s2.y;
  s0.x = s1.x - s2.x; s0.y = s1.y - s2.y;
  fm.x = f0.x - s3.x * .5f; fm.y = f0.y - s3.y * .5f;
  s0.x *= epi3; s0.y *= epi3;
  f0.x += s3.x; f0.y += s3.y;
  f2m.x = fm.x + s0.y; f2m.y = fm.y - s0.x;
  fm.x = fm.x - s0.y; fm.y = fm.y + s0.x;
}
// kf_bfly5 (259-304)
__device__ __forceinline__ void pn_bfly5(float2 &f0, float2 &f1, float2 &f2, float2 &f3, float2 &f4, const float2 &t1,
                                         const float2 &t2, const float2 &t3, const float2 &t4, const float2 &ya,
                                         const float2 &yb) {
  float2 s0 = f0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12;
  PN_CMUL(s1, f1, t1); PN_CMUL(s2, f2, t2); PN_CMUL(s3, f3, t3); PN_CMUL(s4, f4, t4);
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
  f4.x = s5.x + s6.x; f4.y = s5.y + s6.y;
  s11.x = s0.x + (s7.x * yb.x + s8.x * ya.x);
  s11.y = s0.y + (s7.y * yb.x + s8.y * ya.x);
  s12.x = s9.y * ya.y - s10.y * yb.y;
  s12.y = s10.x * yb.y - s9.x * ya.y;
  f2.x = s11.x + s12.x; f2.y = s11.y + s12.y;
  f3.x = s11.x - s12.x; f3.y = s11.y - s12.y;
}

// LN lanes per transform, l = lane index inside the group, U = butterflies per batch
template <int LN, int U>
__device__ __forceinline__ void pn_fft960(float2 *F, const float2 *tw, int l) {
  PN_WAVE_SYNC();
  {                                                   // radix-4, m = 1: 240 butterflies
    constexpr int NIT = (240 + LN - 1) / LN;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += U) {
      float2 a[U][4];
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u), bc = b < 240 ? b : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) a[u][k] = F[4 * bc + k];
      }
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) pn_bfly4_m1(a[u][0], a[u][1], a[u][2], a[u][3]);
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u);
        if (240 % LN == 0 || b < 240) {
#pragma unroll
          for (int k = 0; k < 4; k++) F[4 * b + k] = a[u][k];
        }
      }
    }
  }
  PN_WAVE_SYNC();
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {              // radix-4, m = 4 (fstride 60) then m = 16 (fstride 15)
    const int m = pass ? 16 : 4, fs = pass ? 15 : 60, mm = pass ? 64 : 16;
    constexpr int NIT = (240 + LN - 1) / LN;
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += U) {
      float2 a[U][4], t[U][3];
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u), bc = b < 240 ? b : 0;
        const int i = bc / m, j = bc % m;
        const float2 *f = F + i * mm + j;
#pragma unroll
        for (int k = 0; k < 4; k++) a[u][k] = f[k * m];
        t[u][0] = tw[j * fs]; t[u][1] = tw[2 * j * fs]; t[u][2] = tw[3 * j * fs];
      }
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) pn_bfly4(a[u][0], a[u][1], a[u][2], a[u][3], t[u][0], t[u][1], t[u][2]);
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u);
        if (240 % LN == 0 || b < 240) {
          float2 *f = F + (b / m) * mm + b % m;
#pragma unroll
          for (int k = 0; k < 4; k++) f[k * m] = a[u][k];
        }
      }
    }
    PN_WAVE_SYNC();
  }
  {                                                   // radix-3, m = 64, fstride 5: 320 butterflies; epi3 = tw[fstride*m].y
    const float epi3 = tw[320].y;
    constexpr int NIT = 320 / LN;
    static_assert(320 % LN == 0, "LN");
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += U) {
      float2 a[U][3], t[U][2];
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u);
        const int i = b >> 6, j = b & 63;
        const float2 *f = F + i * 192 + j;
        a[u][0] = f[0]; a[u][1] = f[64]; a[u][2] = f[128];
        t[u][0] = tw[j * 5]; t[u][1] = tw[2 * j * 5];
      }
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) pn_bfly3(a[u][0], a[u][1], a[u][2], t[u][0], t[u][1], epi3);
#pragma unroll
      for (int u = 0; u < U; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u);
        float2 *f = F + (b >> 6) * 192 + (b & 63);
        f[0] = a[u][0]; f[64] = a[u][1]; f[128] = a[u][2];
      }
    }
    PN_WAVE_SYNC();
  }
  {                                                   // radix-5, m = 192, fstride 1: 192 butterflies; ya = tw[m], yb = tw[2m]
    const float2 ya = tw[192], yb = tw[384];
    constexpr int NIT = 192 / LN;
    static_assert(192 % LN == 0, "LN");
    constexpr int U5 = U > 4 ? 4 : U;                 // 18 registers per butterfly
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += U5) {
      float2 a[U5][5], t[U5][4];
#pragma unroll
      for (int u = 0; u < U5; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u);
#pragma unroll
        for (int k = 0; k < 5; k++) a[u][k] = F[b + 192 * k];
#pragma unroll
        for (int k = 0; k < 4; k++) t[u][k] = tw[(k + 1) * b];
      }
#pragma unroll
      for (int u = 0; u < U5; u++) if (it0 + u < NIT)
        pn_bfly5(a[u][0], a[u][1], a[u][2], a[u][3], a[u][4], t[u][0], t[u][1], t[u][2], t[u][3], ya, yb);
#pragma unroll
      for (int u = 0; u < U5; u++) if (it0 + u < NIT) {
        const int b = l + LN * (it0 + u);
#pragma unroll
        for (int k = 0; k < 5; k++) F[b + 192 * k] = a[u][k];
      }
    }
    PN_WAVE_SYNC();
  }
}
