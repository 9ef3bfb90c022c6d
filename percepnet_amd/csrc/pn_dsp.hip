// DSP kernels of the PercepNet frame engine for gfx950 — one wavefront (64 lanes) per stream.
//
//   pn_frontend_kernel : history ring write, window + 960-pt FFT (x3), ERB band energies /
//                        correlation, CELT pitch analysis (downsample + LPC whitening, coarse/fine
//                        xcorr search, octave-error removal), 7-tap comb filter, 70 features
//                        == compute_frame_features + compute_lookahead_band_energy +
//                           create_features (reference denoise.cpp:372-434, 498-506, 487-496)
//   pn_backend_kernel  : pitch-filter mix, band-gain interpolation, inverse transform, window,
//                        overlap-add, PCM conversion
//                        == pitch_filter + gain apply + frame_synthesis (denoise.cpp:436-485,
//                           539-545) + the CLI's float->short (main.cpp:36)
//
// Numerics contract: every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (this file is compiled with -ffp-contract=off; division
// and sqrt are correctly rounded), so the features, the discrete pitch decisions and the
// synthesis are bit-identical to the CPU reference given identical g/r.  Data-parallel work
// (butterflies of one FFT stage, bins, lags) is spread over lanes; every order-sensitive
// reduction (inner products, running energies, band sums, Levinson) runs as the reference's
// sequential chain on one lane — one lane per lag/band, never a shuffle tree.  Only the ADDs of
// such a chain are serially dependent: operands are fetched from LDS 16 at a time and the
// products formed ahead of the chain.
//
// Work distribution: 512-thread blocks = 8 wavefronts = 8 concurrent streams.  The block stages
// the shared read-only tables (twiddles, window, digit-reversal, band map: 13.6 KB) into LDS
// once; each wavefront owns a private 8.3 KB LDS slice (FFT buffer, aliased by the pitch scratch)
// and loops over streams  s = blockIdx.x*8 + wave, += gridDim.x*8.  After the table staging no
// block-level barrier exists: a wavefront only ever synchronises with itself (LDS operations of
// one wave execute in order; PN_WAVE_SYNC is the compiler-level fence), so the 16 waves a CU
// holds (2 blocks, 80 KB LDS each) drift freely and hide each other's chain latency.
#include "pn_common.h"

#define LANES 64
#ifndef PN_DSP_WPB
#define PN_DSP_WPB 8          // wavefronts (= concurrent streams) per block
#endif
#ifndef PN_DSP_WAVES_PER_SIMD
#define PN_DSP_WAVES_PER_SIMD 4
#endif
#define WPB PN_DSP_WPB
#define DSP_THREADS (LANES * WPB)

// same-wave LDS producer -> consumer ordering
#define PN_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// same-wave global-memory store -> load ordering (drains vmcnt)
#define PN_WAVE_SYNC_GLOBAL() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

struct alignas(16) PnDspTablesLds {
  float2 tw[PN_NFFT];            // 7680 B
  float win[PN_FRAME];           // 1920 B
  float frac[PN_SPEC_BINS];      // 1600 B
  int16_t bitrev[PN_NFFT];       // 1920 B
  int16_t border[PN_NB + 2];
  uint8_t band[PN_SPEC_BINS];
  float comb_w[8];
};
struct alignas(16) PnDspWaveLds {
  float2 fft[PN_NFFT];           // 7680 B  FFT work buffer; pitch scratch / per-bin products alias it
  float e[4][PN_NB + 2];         // 576 B
};
struct PnDspShared {
  PnDspTablesLds t;
  PnDspWaveLds w[WPB];
};

__device__ __forceinline__ void pn_stage_tables(PnDspTablesLds &S, const PnTables *__restrict__ T) {
  const int tid = threadIdx.x;
  for (int i = tid; i < PN_NFFT; i += DSP_THREADS) {
    S.tw[i] = make_float2(T->tw[2 * i], T->tw[2 * i + 1]);
    S.bitrev[i] = T->bitrev[i];
  }
  for (int i = tid; i < PN_FRAME; i += DSP_THREADS) S.win[i] = T->half_window[i];
  for (int i = tid; i < PN_SPEC_BINS; i += DSP_THREADS) { S.frac[i] = T->bin_frac[i]; S.band[i] = T->bin_band[i]; }
  if (tid < PN_NB + 2) S.border[tid] = T->border[tid];
  if (tid < 8) S.comb_w[tid] = T->comb_hann[tid];
  __syncthreads();
}

// ---- 960-point FFT in LDS (opus_fft_impl, kiss_fft.cpp:518-564, factors 5,3,4,4,4) -----------
// Input must already be scaled by 1/960 and digit-reverse scattered (opus_fft_c 578-585).
#define CMUL(m, a, b) do { (m).x = (a).x*(b).x - (a).y*(b).y; (m).y = (a).x*(b).y + (a).y*(b).x; } while (0)

__device__ __forceinline__ void pn_fft960_lds(float2 *F, const float2 *tw, int lane) {
  PN_WAVE_SYNC();
  // radix-4, m=1 (degenerate twiddle-free butterfly, kiss_fft.cpp:112-131)
  for (int b = lane; b < 240; b += LANES) {
    float2 *f = F + 4 * b;
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
  PN_WAVE_SYNC();
  // radix-4, m=4 (fstride 60, mm 16) then m=16 (fstride 15, mm 64)  (kiss_fft.cpp:139-166)
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int m = pass ? 16 : 4, fs = pass ? 15 : 60, mm = pass ? 64 : 16;
    for (int b = lane; b < 240; b += LANES) {
      const int i = b / m, j = b % m;
      float2 *f = F + i * mm + j;
      float2 f0 = f[0], fm = f[m], f2m = f[2 * m], f3m = f[3 * m];
      const float2 t1 = tw[j * fs], t2 = tw[2 * j * fs], t3 = tw[3 * j * fs];
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
      f[0] = f0; f[m] = fm; f[2 * m] = f2m; f[3 * m] = f3m;
    }
    PN_WAVE_SYNC();
  }
  // radix-3, m=64, fstride 5, mm 192 (kiss_fft.cpp:196-227); epi3 = tw[fstride*m]
  {
    const float epi3 = tw[320].y;
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int j = lane;
      float2 *f = F + i * 192 + j;
      float2 f0 = f[0], fm = f[64], f2m = f[128], s0, s1, s2, s3;
      CMUL(s1, fm, tw[j * 5]); CMUL(s2, f2m, tw[2 * j * 5]);
      s3.x = s1.x + s2.x; s3.y = s1.y + s2.y;
      s0.x = s1.x - s2.x; s0.y = s1.y - s2.y;
      fm.x = f0.x - s3.x * .5f; fm.y = f0.y - s3.y * .5f;
      s0.x *= epi3; s0.y *= epi3;
      f0.x += s3.x; f0.y += s3.y;
      f2m.x = fm.x + s0.y; f2m.y = fm.y - s0.x;
      fm.x = fm.x - s0.y; fm.y = fm.y + s0.x;
      f[0] = f0; f[64] = fm; f[128] = f2m;
    }
    PN_WAVE_SYNC();
  }
  // radix-5, m=192, fstride 1 (kiss_fft.cpp:259-304); ya = tw[m], yb = tw[2m]
  {
    const float2 ya = tw[192], yb = tw[384];
#pragma unroll
    for (int it = 0; it < 3; it++) {
      const int u = lane + LANES * it;
      float2 *f = F + u;
      float2 f0 = f[0], f1 = f[192], f2 = f[384], f3 = f[576], f4 = f[768];
      float2 s0 = f0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12;
      CMUL(s1, f1, tw[u]); CMUL(s2, f2, tw[2 * u]); CMUL(s3, f3, tw[3 * u]); CMUL(s4, f4, tw[4 * u]);
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
      f[0] = f0; f[192] = f1; f[384] = f2; f[576] = f3; f[768] = f4;
    }
    PN_WAVE_SYNC();
  }
}

// ---- band reductions (denoise.cpp:89-160): lane b owns band b, sums in the reference's order ---
// PROD = false: tmp = |A[k]|^2 formed on the fly (compute_band_energy);
// PROD = true : tmp[k] precomputed per bin (compute_band_corr: X.r*P.r then += X.i*P.i).
template <bool PROD>
__device__ __forceinline__ float pn_band_reduce(const PnDspTablesLds &S, const float2 *A, const float *prod, int b) {
  float sum = 0;
  if (b < PN_NB) {
    if (b >= 1) {   // contributions `sum[i+1] += frac*tmp` of interval i = b-1
      const int lo = S.border[b - 1], hi = S.border[b];
      for (int k = lo; k < hi; k++) {
        float tmp;
        if (PROD) tmp = prod[k];
        else { tmp = A[k].x * A[k].x; tmp += A[k].y * A[k].y; }
        sum += S.frac[k] * tmp;
      }
    }
    if (b <= PN_NB - 2) {  // contributions `sum[i] += (1-frac)*tmp` of interval i = b
      const int lo = S.border[b], hi = S.border[b + 1];
      for (int k = lo; k < hi; k++) {
        float tmp;
        if (PROD) tmp = prod[k];
        else { tmp = A[k].x * A[k].x; tmp += A[k].y * A[k].y; }
        sum += (1 - S.frac[k]) * tmp;
      }
    }
    if (b == 0 || b == PN_NB - 1) sum *= 2;
  }
  return sum;
}

// logical comb_buf index j in [0,5760) (newest sample at 5759) -> ring offset
__device__ __forceinline__ int pn_ring(int j, int base_slot) {
  const int f = j / PN_FRAME;
  int slot = base_slot + f;
  if (slot >= PN_HIST_FRAMES) slot -= PN_HIST_FRAMES;
  return slot * PN_FRAME + (j - f * PN_FRAME);
}

// window (apply_window, denoise.cpp:282-289) + 1/960 scale + digit-reverse scatter of 960 real
// samples starting at logical history index j0
__device__ __forceinline__ void pn_window_scatter(const PnDspTablesLds &S, float2 *F, const float *__restrict__ h,
                                                  int base_slot, int j0, int lane) {
  const float scale = 1.f / PN_NFFT;
#pragma unroll 5
  for (int it = 0; it < 15; it++) {
    const int i = lane + LANES * it;
    const float w = S.win[i < PN_FRAME ? i : PN_WINDOW - 1 - i];
    const float v = h[pn_ring(j0 + i, base_slot)] * w;
    F[S.bitrev[i]] = make_float2(scale * v, scale * 0.f);
  }
}

// acc + sum_{j<N} a[j]*b[j], adds strictly in j order (celt_inner_prod / xcorr_kernel /
// dual_inner_prod, pitch.h:53-144).  `a` is wave-uniform (every lane correlates the same x against
// its own lag of y): it is read with ds_read_b128 at one address for all lanes (an LDS broadcast,
// one instruction per 4 steps, `a` must be 16-byte aligned); the per-lane operand b costs one
// ds_read_b32 per step.  Operands are fetched 16 steps ahead of the dependent add chain; only the
// adds are serially dependent.  N must be a multiple of 4.
template <int N>
__device__ __forceinline__ float pn_chain_u(const float *a, const float *b, float acc) {
  constexpr int U = 16, NF = N / U, R = N % U;
#pragma unroll 1
  for (int blk = 0; blk < NF; blk++) {
    float4 av[4]; float bv[U];
#pragma unroll
    for (int v = 0; v < 4; v++) av[v] = *reinterpret_cast<const float4 *>(a + U * blk + 4 * v);
#pragma unroll
    for (int u = 0; u < U; u++) bv[u] = b[U * blk + u];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      acc = acc + av[v].x * bv[4 * v]; acc = acc + av[v].y * bv[4 * v + 1];
      acc = acc + av[v].z * bv[4 * v + 2]; acc = acc + av[v].w * bv[4 * v + 3];
    }
  }
  if (R) {
    float4 av[R / 4 ? R / 4 : 1]; float bv[R ? R : 1];
#pragma unroll
    for (int v = 0; v < R / 4; v++) av[v] = *reinterpret_cast<const float4 *>(a + U * NF + 4 * v);
#pragma unroll
    for (int u = 0; u < R; u++) bv[u] = b[U * NF + u];
#pragma unroll
    for (int v = 0; v < R / 4; v++) {
      acc = acc + av[v].x * bv[4 * v]; acc = acc + av[v].y * bv[4 * v + 1];
      acc = acc + av[v].z * bv[4 * v + 2]; acc = acc + av[v].w * bv[4 * v + 3];
    }
  }
  return acc;
}

// find_best_pitch (pitch.cpp:46-104, float instantiation).  Wave-uniform recurrence: every lane
// runs the same sequential loop on LDS-broadcast operands.  The squares y[j]^2 and the window
// updates d[i] = y[i+LEN]^2 - y[i]^2 are formed lane-parallel first (same roundings) into `scr`
// (>= max(LEN rounded up to 64... see callers) so the serial loops are one add (resp. add + max +
// compare) per step.  xcorr, y, scr 16-byte aligned; MAXP, LEN multiples of... any.
template <int LEN, int MAXP>
__device__ __forceinline__ void pn_find_best_pitch(const float *xcorr, const float *y, float *sq /*[64]*/,
                                                   float *d /*[MAXP rounded up to 4]*/, int lane, int &bp0, int &bp1) {
  // d[i] = y[i+LEN]*y[i+LEN] - y[i]*y[i]
  for (int i = lane; i < ((MAXP + 3) & ~3); i += LANES) {
    const int ic = i < MAXP ? i : MAXP - 1;
    const float a = y[ic + LEN], c = y[ic];
    d[i] = a * a - c * c;
  }
  float Syy = 1.0f;
  constexpr int NBy = (LEN + 63) / 64;
#pragma unroll 1
  for (int blk = 0; blk < NBy; blk++) {
    const int j = 64 * blk + lane;
    const float yv = y[j < LEN ? j : 0];
    PN_WAVE_SYNC();
    sq[lane] = yv * yv;
    PN_WAVE_SYNC();
#pragma unroll
    for (int v = 0; v < 16; v++) {
      if (64 * blk + 4 * v < LEN) {           // LEN is a multiple of 4; blk uniform
        const float4 q = *reinterpret_cast<const float4 *>(sq + 4 * v);
        Syy = Syy + q.x; Syy = Syy + q.y; Syy = Syy + q.z; Syy = Syy + q.w;
      }
    }
  }
  PN_WAVE_SYNC();
  float bn0 = -1, bn1 = -1, bd0 = 0, bd1 = 0;
  bp0 = 0; bp1 = 1;
#define PN_FBP_STEP(xc_, dd_, idx_)                                                                  \
  if ((idx_) < MAXP) {                                                                              \
    if ((xc_) > 0) {                                                                                \
      float x16 = (xc_);                                                                            \
      x16 *= 1e-12f;                                                                                \
      const float num = x16 * x16;                                                                  \
      if (num * bd1 > bn1 * Syy) {                                                                  \
        if (num * bd0 > bn0 * Syy) { bn1 = bn0; bd1 = bd0; bp1 = bp0; bn0 = num; bd0 = Syy; bp0 = (idx_); } \
        else { bn1 = num; bd1 = Syy; bp1 = (idx_); }                                                \
      }                                                                                             \
    }                                                                                               \
    Syy += (dd_);                                                                                   \
    Syy = (1 > Syy) ? 1 : Syy;                                                                      \
  }
#pragma unroll 1
  for (int i0 = 0; i0 < MAXP; i0 += 16) {
    float4 xv[4], dv[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = (i0 + 4 * v < ((MAXP + 3) & ~3)) ? i0 + 4 * v : 0;
      xv[v] = *reinterpret_cast<const float4 *>(xcorr + i);
      dv[v] = *reinterpret_cast<const float4 *>(d + i);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
      PN_FBP_STEP(xv[v].x, dv[v].x, i0 + 4 * v) PN_FBP_STEP(xv[v].y, dv[v].y, i0 + 4 * v + 1)
      PN_FBP_STEP(xv[v].z, dv[v].z, i0 + 4 * v + 2) PN_FBP_STEP(xv[v].w, dv[v].w, i0 + 4 * v + 3)
    }
  }
#undef PN_FBP_STEP
}

__device__ __forceinline__ float pn_pitch_gain(float xy, float xx, float yy) { return xy / sqrtf(1 + xx * yy); }

template <typename TIn>
__global__ __launch_bounds__(DSP_THREADS, PN_DSP_WAVES_PER_SIMD) void pn_frontend_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t,
    const TIn *__restrict__ in,           // [n_streams][480]
    float *__restrict__ hist,             // [n_streams][12][480] ring
    float2 *__restrict__ Xspec,           // [n_streams][400]
    float2 *__restrict__ Pspec,           // [n_streams][400]
    float *__restrict__ feat,             // [n_streams][PN_FEAT_STRIDE]
    int *__restrict__ silence,            // [n_streams]
    int *__restrict__ last_period, float *__restrict__ last_gain) {
  __shared__ PnDspShared SH;
  const int lane = threadIdx.x & (LANES - 1), wave = threadIdx.x >> 6;
  pn_stage_tables(SH.t, T);
  const PnDspTablesLds &S = SH.t;
  PnDspWaveLds &W = SH.w[wave];
  const int new_slot = frame_t % PN_HIST_FRAMES;
  const int base_slot = (frame_t + 1) % PN_HIST_FRAMES;   // slot of logical frame 0 (oldest)
  float *pbuf = reinterpret_cast<float *>(W.fft);         // [864]  pitch scratch aliases the FFT buffer
  // float offsets inside the wave's 1920-float buffer (all 16-byte aligned where b128-read):
  float *xcorr = pbuf + 864;                              // [ 864,1184) xcorr[294] (+pad); later p/q scratch of yy_lookup
  float *y4 = pbuf + 1184;                                // [1184,1571) y_lp4[387] = pbuf[2j]; later d[] of the fine pass
  float *yyl = pbuf + 1187;                               // [1187,1572) yy_lookup[385], &yyl[1] is 16-byte aligned
  float *x4 = pbuf + 1600;                                // [1600,1856) x_lp4[240] (+pad) = pbuf[384+2j]; later d[] of the coarse pass
  float *sq64 = pbuf + 1856;                              // [1856,1920) 64-float broadcast scratch
  float *prod = reinterpret_cast<float *>(W.fft + 480);   // [400]  per-bin X.P products (entries >= 480 are free after the FFT)

  for (int s = blockIdx.x * WPB + wave; s < n_streams; s += gridDim.x * WPB) {
    float *h = hist + (size_t)s * PN_HIST;
    // -- history: the shift+append of denoise.cpp:388-389 becomes one ring-slot write ---------
    for (int i = lane; i < PN_FRAME; i += LANES) {
      float v;
      if (sizeof(TIn) == 2) v = ((float)in[(size_t)s * PN_FRAME + i]) / 32768.f;   // main.cpp:34
      else v = (float)in[(size_t)s * PN_FRAME + i];
      h[new_slot * PN_FRAME + i] = v;
    }
    PN_WAVE_SYNC_GLOBAL();
    // -- X = FFT(window(comb_buf[2400,3360))), Ex (frame_analysis 333-346) ---------------------
    pn_window_scatter(S, W.fft, h, base_slot, 2400, lane);
    pn_fft960_lds(W.fft, S.tw, lane);
    for (int k = lane; k < PN_SPEC_BINS; k += LANES) Xspec[(size_t)s * PN_SPEC_BINS + k] = W.fft[k];
    const float Ex = pn_band_reduce<false>(S, W.fft, nullptr, lane);
    PN_WAVE_SYNC();
    // -- look-ahead band energy of the newest 960 samples (498-506) ------------------------------
    pn_window_scatter(S, W.fft, h, base_slot, PN_HIST - PN_WINDOW, lane);
    pn_fft960_lds(W.fft, S.tw, lane);
    const float Ey = pn_band_reduce<false>(S, W.fft, nullptr, lane);
    PN_WAVE_SYNC();

    // -- pitch_downsample (pitch.cpp:148-216) of pitch_buf == comb_buf[1632,3360) ----------------
#pragma unroll 2
    for (int it = 0; it < 14; it++) {
      const int i = lane + LANES * it;
      if (i < 864) {
        float v;
        if (i == 0) v = .5f * (.5f * (h[pn_ring(1632 + 1, base_slot)]) + h[pn_ring(1632, base_slot)]);
        else v = .5f * (.5f * (h[pn_ring(1632 + 2 * i - 1, base_slot)] + h[pn_ring(1632 + 2 * i + 1, base_slot)]) +
                        h[pn_ring(1632 + 2 * i, base_slot)]);
        pbuf[i] = v;
      }
    }
    PN_WAVE_SYNC();
    // _celt_autocorr (celt_lpc.cpp:198-279): lane k holds lag k (lanes > 4 shadow lag 4)
    float ac[5];
    {
      const int lag = lane < 4 ? lane : 4;
      float ack = pn_chain_u<860>(pbuf, pbuf + lag, 0.f);
      float d = 0;
      for (int i = lag + 860; i < 864; i++) d = d + pbuf[i] * pbuf[i - lag];
      ack += d;
#pragma unroll
      for (int k = 0; k < 5; k++) ac[k] = __shfl(ack, k);
    }
    ac[0] *= 1.0001f;
#pragma unroll
    for (int i = 1; i <= 4; i++) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
    // _celt_lpc (celt_lpc.cpp:37-88), p = 4; wave-uniform
    float lpc[4] = {0, 0, 0, 0};
    {
      float error = ac[0];
      if (ac[0] != 0) {
        bool done = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (!done) {
            float rr = 0;
#pragma unroll
            for (int j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
            rr += ac[i + 1];
            const float r = (float)((double)(-rr) / ((double)error + 0.00001));
            lpc[i] = r;
#pragma unroll
            for (int j = 0; j < ((i + 1) >> 1); j++) {
              const float t1 = lpc[j], t2 = lpc[i - 1 - j];
              lpc[j] = t1 + r * t2;
              lpc[i - 1 - j] = t2 + r * t1;
            }
            error = error - (r * r) * error;
            if (error < .001f * ac[0]) done = true;
          }
        }
      }
    }
    float lpc2[5];
    {
      float tmp = 1.0f;
#pragma unroll
      for (int i = 0; i < 4; i++) { tmp = .9f * tmp; lpc[i] = lpc[i] * tmp; }
      lpc2[0] = lpc[0] + .8f;
      lpc2[1] = lpc[1] + .8f * lpc[0];
      lpc2[2] = lpc[2] + .8f * lpc[1];
      lpc2[3] = lpc[3] + .8f * lpc[2];
      lpc2[4] = .8f * lpc[3];
    }
    // celt_fir5 (pitch.cpp:106-145), in place: read all taps first, then write
    {
      float y[14];
#pragma unroll
      for (int it = 0; it < 14; it++) {
        const int i = lane + LANES * it;
        float sum = 0;
        if (i < 864) {
          sum = pbuf[i];
          sum = sum + lpc2[0] * (i >= 1 ? pbuf[i - 1] : 0.f);
          sum = sum + lpc2[1] * (i >= 2 ? pbuf[i - 2] : 0.f);
          sum = sum + lpc2[2] * (i >= 3 ? pbuf[i - 3] : 0.f);
          sum = sum + lpc2[3] * (i >= 4 ? pbuf[i - 4] : 0.f);
          sum = sum + lpc2[4] * (i >= 5 ? pbuf[i - 5] : 0.f);
        }
        y[it] = sum;
      }
      PN_WAVE_SYNC();
#pragma unroll
      for (int it = 0; it < 14; it++) { const int i = lane + LANES * it; if (i < 864) pbuf[i] = y[it]; }
      PN_WAVE_SYNC();
    }

    // -- pitch_search (pitch.cpp:283-386): x_lp = pbuf+384, y = pbuf, len 960, max_pitch 588 ----
    // coarse: the 4x-decimated signals x_lp4[j] = pbuf[384+2j] (240) and y_lp4[j] = pbuf[2j] (387) are
    // first copied out contiguously (conflict-free reads, and x_lp4 can be lane-broadcast);
    // lane owns lags {lane, lane+64, lane+128}: three independent j-ascending chains
    for (int j = lane; j < 387; j += LANES) y4[j] = pbuf[2 * j];
    for (int j = lane; j < 256; j += LANES) x4[j] = pbuf[j < 240 ? 384 + 2 * j : 0];
    PN_WAVE_SYNC();
    {
      const int i0 = lane, i1 = lane + 64, i2 = (lane + 128 < 147) ? lane + 128 : 146;
      float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll 1
      for (int j0 = 0; j0 < 240; j0 += 8) {
        const float4 a0 = *reinterpret_cast<const float4 *>(x4 + j0), a1 = *reinterpret_cast<const float4 *>(x4 + j0 + 4);
        const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float b0[8], b1[8], b2[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { b0[u] = y4[i0 + j0 + u]; b1[u] = y4[i1 + j0 + u]; b2[u] = y4[i2 + j0 + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) { s0 = s0 + a[u] * b0[u]; s1 = s1 + a[u] * b1[u]; s2 = s2 + a[u] * b2[u]; }
      }
      xcorr[i0] = s0; xcorr[i1] = s1;
      if (lane + 128 < 147) xcorr[i2] = s2;
    }
    PN_WAVE_SYNC();
    int bp0, bp1;
    pn_find_best_pitch<240, 147>(xcorr, y4, sq64, x4, lane, bp0, bp1);   // x_lp4 is dead: its space holds d[]
    PN_WAVE_SYNC();
    // fine: only lags within +-2 of 2*best (pitch.cpp:344-361); other entries are 0
    for (int i = lane; i < 294; i += LANES) xcorr[i] = 0;
    PN_WAVE_SYNC();
    {
      const int c = (lane < 5) ? (2 * bp0 - 2 + lane) : (2 * bp1 - 2 + (lane - 5));
      const bool act = lane < 10 && c >= 0 && c < 294;
      const float sum = pn_chain_u<480>(pbuf + 384, pbuf + (act ? c : 0), 0.f);
      if (act) xcorr[c] = (-1 > sum) ? -1 : sum;   // duplicates (overlapping windows) write the same value
    }
    PN_WAVE_SYNC();
    pn_find_best_pitch<480, 294>(xcorr, pbuf, sq64, y4, lane, bp0, bp1);   // y_lp4 is dead: its space holds d[]
    int offset = 0;
    if (bp0 > 0 && bp0 < 294 - 1) {
      const float a = xcorr[bp0 - 1], b = xcorr[bp0], c = xcorr[bp0 + 1];
      if ((c - a) > .7f * (b - a)) offset = 1;
      else if ((a - c) > .7f * (b - c)) offset = -1;
    }
    const float pitch_corr = xcorr[bp0];
    int pitch_index = PN_PITCH_MAX - (2 * bp0 - offset);       // denoise.cpp:408
    PN_WAVE_SYNC();

    // -- remove_doubling (pitch.cpp:424-527): maxperiod 384, minperiod 30, N 480, x = pbuf+384 -----
    float pg;
    {
      const float *x = pbuf + 384;
      const int prev_period = last_period[s] / 2;
      const float prev_gain = last_gain[s];
      int T0 = pitch_index / 2;
      if (T0 >= 384) T0 = 383;
      // lane 0: xx, lane 1: xy(T0), lanes 2..15: xy(T1_k), lanes 16..29: xy2(T1b_k)  (k = 2..15)
      int lag = 0, T1 = 0, T1b = 0;
      const int k = (lane >= 16) ? lane - 14 : lane;            // lanes 2..15 and 16..29 -> k = 2..15
      if (lane == 1) lag = T0;
      else if (lane >= 2 && lane < 30) {
        static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
        T1 = (2 * T0 + k) / (2 * k);
        if (k == 2) { if (T1 + T0 > 384) T1b = T0; else T1b = T0 + T1; }
        else T1b = (2 * second_check[k] * T0 + k) / (2 * k);
        lag = (lane < 16) ? T1 : T1b;
      }
      const float dot = pn_chain_u<480>(x, x - lag, 0.f);   // lanes >= 30 shadow lag 0
      const float xx = __shfl(dot, 0);
      float xy = __shfl(dot, 1);
      // yy_lookup (pitch.cpp:449-455): strictly sequential running energy, wave-uniform.  The squares
      // x[-i]^2 and x[N-i]^2 are formed lane-parallel, 64 at a time, into a broadcast scratch (the
      // dead xcorr area); the recurrence then reads them 4 per ds_read_b128 and lane 0 stores the
      // clamped results 4 at a time (yyl[1] is 16-byte aligned).
      {
        float *pq = xcorr;                     // p[64] | q[64]
        float yy = xx;
        if (lane == 0) yyl[0] = xx;
#pragma unroll 1
        for (int blk = 0; blk < 6; blk++) {        // i = 1 + 64*blk + u, u < 64  (384 = 6*64)
          const int i = 1 + 64 * blk + lane;
          const float a = x[-i], c = x[480 - i];
          PN_WAVE_SYNC();
          pq[lane] = a * a; pq[64 + lane] = c * c;
          PN_WAVE_SYNC();
#pragma unroll
          for (int v = 0; v < 16; v++) {
            const float4 p4 = *reinterpret_cast<const float4 *>(pq + 4 * v);
            const float4 q4 = *reinterpret_cast<const float4 *>(pq + 64 + 4 * v);
            float4 o;
            yy = yy + p4.x - q4.x; o.x = (0 > yy) ? 0 : yy;
            yy = yy + p4.y - q4.y; o.y = (0 > yy) ? 0 : yy;
            yy = yy + p4.z - q4.z; o.z = (0 > yy) ? 0 : yy;
            yy = yy + p4.w - q4.w; o.w = (0 > yy) ? 0 : yy;
            if (lane == 0) *reinterpret_cast<float4 *>(yyl + 1 + 64 * blk + 4 * v) = o;
          }
        }
      }
      PN_WAVE_SYNC();
      float yy = yyl[T0];
      float best_xy = xy, best_yy = yy;
      const float g0 = pn_pitch_gain(xy, xx, yy);
      float g = g0;
      int Tsel = T0;
      // k = 2..15 evaluated in parallel on lanes 2..15; the sequential loop's "last hit wins"
      // becomes "highest k among hits"; its `break` at T1 < minperiod is a prefix condition.
      const float xy2 = __shfl(dot, (lane + 14) & 63);          // partner lane holds xy2 for the same k
      bool hit = false;
      float xyk = 0, yyk = 0, g1 = 0;
      if (lane >= 2 && lane < 16 && T1 >= 30) {
        xyk = .5f * (dot + xy2);
        yyk = .5f * (yyl[T1] + yyl[T1b]);
        g1 = pn_pitch_gain(xyk, xx, yyk);
        float cont;
        const int dT = (T1 - prev_period) < 0 ? -(T1 - prev_period) : (T1 - prev_period);
        if (dT <= 1) cont = prev_gain;
        else if (dT <= 2 && 5 * k * k < T0) cont = .5f * prev_gain;
        else cont = 0;
        float thresh = (.3f > .7f * g0 - cont) ? .3f : .7f * g0 - cont;
        if (T1 < 3 * 30) thresh = (.4f > .85f * g0 - cont) ? .4f : .85f * g0 - cont;
        hit = g1 > thresh;
      }
      const unsigned long long m = __ballot(hit);
      if (m) {
        const int win = 63 - __clzll(m);
        best_xy = __shfl(xyk, win); best_yy = __shfl(yyk, win);
        Tsel = __shfl(T1, win); g = __shfl(g1, win);
      }
      best_xy = (0 > best_xy) ? 0 : best_xy;
      if (best_yy <= best_xy) pg = 1.0f; else pg = best_xy / (best_yy + 1);
      const float xc = pn_chain_u<480>(x, x - (Tsel + (lane < 3 ? lane : 2) - 1), 0.f);
      const float xc0 = __shfl(xc, 0), xc1 = __shfl(xc, 1), xc2 = __shfl(xc, 2);
      int off2;
      if ((xc2 - xc0) > .7f * (xc1 - xc0)) off2 = 1;
      else if ((xc0 - xc2) > .7f * (xc1 - xc2)) off2 = -1;
      else off2 = 0;
      if (pg > g) pg = g;
      pitch_index = 2 * Tsel + off2;
      if (pitch_index < PN_PITCH_MIN) pitch_index = PN_PITCH_MIN;
    }
    if (lane == 0) { last_period[s] = pitch_index; last_gain[s] = pg; }
    PN_WAVE_SYNC();

    // -- comb filter (denoise.cpp:416-422) + window + FFT -> P, Ep, Exp -------------------------
    {
      const float scale = 1.f / PN_NFFT;
#pragma unroll 3
      for (int it = 0; it < 15; it++) {
        const int i = lane + LANES * it;
        float p = 0;
#pragma unroll
        for (int k = -PN_COMB_M; k <= PN_COMB_M; k++)
          p += h[pn_ring(2400 - pitch_index * k + i, base_slot)] * S.comb_w[k + PN_COMB_M];
        const float v = p * S.win[i < PN_FRAME ? i : PN_WINDOW - 1 - i];
        W.fft[S.bitrev[i]] = make_float2(scale * v, scale * 0.f);
      }
    }
    pn_fft960_lds(W.fft, S.tw, lane);
    PN_WAVE_SYNC_GLOBAL();                   // X was stored by this wave earlier in the iteration
    for (int k = lane; k < PN_SPEC_BINS; k += LANES) {
      const float2 P = W.fft[k];
      const float2 X = Xspec[(size_t)s * PN_SPEC_BINS + k];
      Pspec[(size_t)s * PN_SPEC_BINS + k] = P;
      float tmp = X.x * P.x;                 // compute_band_corr's per-bin term (denoise.cpp:136-137)
      tmp += X.y * P.y;
      prod[k] = tmp;
    }
    const float Ep = pn_band_reduce<false>(S, W.fft, nullptr, lane);
    PN_WAVE_SYNC();
    float Exp = pn_band_reduce<true>(S, nullptr, prod, lane);
    if (lane < PN_NB) {
      // double island, denoise.cpp:427
      Exp = (float)fmin(1.0, fmax(0.0, (double)Exp / sqrt(1e-15 + (double)(Ex * Ep))));
      W.e[0][lane] = Ex;
    }
    PN_WAVE_SYNC();
    // silence = sum(Ex) < 0.1 (429-433): sequential sum
    if (lane == 0) {
      float E = 0;
      for (int i = 0; i < PN_NB; i++) E += W.e[0][i];
      silence[s] = ((double)E < 0.1) ? 1 : 0;
    }
    // -- create_features (487-496) -----------------------------------------------------------------
    float *f = feat + (size_t)s * PN_FEAT_STRIDE;
    if (lane < PN_NB) { f[lane] = Ey * 30; f[PN_NB + lane] = Exp * 30; }
    if (lane == 0) { f[68] = (float)pitch_index / (PN_PITCH_MAX - 3 * PN_PITCH_MIN); f[69] = pitch_corr; }
    PN_WAVE_SYNC();
  }
}

// float -> int16 as the reference CLI's x86-64 build does it (main.cpp:36): truncate toward zero
// to int32 (cvttss2si; NaN / out of range -> 0x80000000), keep the low 16 bits.
__device__ __forceinline__ int16_t pn_f2s(float v) {
  const int32_t t = (fabsf(v) < 2147483648.f) ? (int32_t)v : (int32_t)0x80000000;
  return (int16_t)(uint16_t)((uint32_t)t & 0xffffu);
}

template <typename TOut>
__global__ __launch_bounds__(DSP_THREADS, PN_DSP_WAVES_PER_SIMD) void pn_backend_kernel(
    const PnTables *__restrict__ T, int n_streams,
    const float2 *__restrict__ Xspec, const float2 *__restrict__ Pspec,
    const float *__restrict__ gr,          // [n_streams][68]  g | r
    const int *__restrict__ silence,
    float *__restrict__ synth_mem,         // [n_streams][480]
    TOut *__restrict__ out) {              // [n_streams][480]
  __shared__ PnDspShared SH;
  const int lane = threadIdx.x & (LANES - 1), wave = threadIdx.x >> 6;
  pn_stage_tables(SH.t, T);
  const PnDspTablesLds &S = SH.t;
  PnDspWaveLds &W = SH.w[wave];
  const float scale = 1.f / PN_NFFT;
  for (int s = blockIdx.x * WPB + wave; s < n_streams; s += gridDim.x * WPB) {
    if (lane < PN_NB) {
      const float g = gr[(size_t)s * 68 + lane], r = gr[(size_t)s * 68 + PN_NB + lane];
      W.e[0][lane] = g; W.e[1][lane] = r; W.e[2][lane] = 1 - r;
    }
    PN_WAVE_SYNC();
    const bool sil = silence[s] != 0;
    // pitch_filter (436-485, skipped when silent, 536-538), gain (539-544), then the Hermitian
    // extension + scale + digit-reverse scatter of inverse_transform (306-317).  Bins >= 400
    // are exactly 0 (interp_band_gain never writes them).
#pragma unroll 3
    for (int it = 0; it < 15; it++) {
      const int i = lane + LANES * it;
      const int k = (i <= PN_FRAME) ? i : PN_WINDOW - i;
      float2 x = make_float2(0.f, 0.f);
      if (k < PN_SPEC_BINS) {
        x = Xspec[(size_t)s * PN_SPEC_BINS + k];
        const int b = S.band[k];
        const float fr = S.frac[k];
        if (!sil) {
          const float2 p = Pspec[(size_t)s * PN_SPEC_BINS + k];
          const float rf1 = (1 - fr) * W.e[2][b] + fr * W.e[2][b + 1];
          x.x = rf1 * x.x; x.y = rf1 * x.y;
          const float rf2 = (1 - fr) * W.e[1][b] + fr * W.e[1][b + 1];
          x.x += rf2 * p.x; x.y += rf2 * p.y;
        }
        const float gf = (1 - fr) * W.e[0][b] + fr * W.e[0][b + 1];
        x.x *= gf; x.y *= gf;
      }
      if (i > PN_FRAME) x.y = -x.y;
      W.fft[S.bitrev[i]] = make_float2(scale * x.x, scale * x.y);
    }
    pn_fft960_lds(W.fft, S.tw, lane);
    // reversed read-out x960 (318-323), window, overlap-add (352-359)
    float *sm = synth_mem + (size_t)s * PN_FRAME;
    for (int i = lane; i < PN_FRAME; i += LANES) {
      const float t_lo = (PN_WINDOW * W.fft[i == 0 ? 0 : PN_WINDOW - i].x) * S.win[i];
      const int i2 = PN_FRAME + i;                       // second half, window index 959 - i2
      const float t_hi = (PN_WINDOW * W.fft[PN_WINDOW - i2].x) * S.win[PN_WINDOW - 1 - i2];
      const float o = t_lo + sm[i];
      sm[i] = t_hi;
      if (sizeof(TOut) == 2) out[(size_t)s * PN_FRAME + i] = (TOut)pn_f2s(o * 32768);
      else out[(size_t)s * PN_FRAME + i] = (TOut)o;
    }
    PN_WAVE_SYNC();
  }
}

// ---- launchers -------------------------------------------------------------------------------
// blocks_per_cu: 2 fills the CUs (80 KB LDS each); 1 leaves half of every CU's LDS/registers free so
// that an MFMA-bound network kernel of another frame can be co-resident (pipelined mode)
static inline int pn_dsp_grid(int n_streams, int blocks_per_cu) {
  const int need = (n_streams + WPB - 1) / WPB;
  const int full = PN_DSP_WAVES_PER_SIMD * 4 / WPB;
  const int cap = 256 * ((blocks_per_cu > 0 && blocks_per_cu < full) ? blocks_per_cu : full);
  return need < cap ? need : cap;
}

void pn_launch_frontend(hipStream_t st, const PnTables *T, int n_streams, int frame_t, const void *in, int in_is_i16,
                        float *hist, float2 *Xs, float2 *Ps, float *feat, int *silence, int *last_period,
                        float *last_gain, int blocks_per_cu) {
  const int grid = pn_dsp_grid(n_streams, blocks_per_cu);
  if (in_is_i16)
    hipLaunchKernelGGL(pn_frontend_kernel<int16_t>, dim3(grid), dim3(DSP_THREADS), 0, st, T, n_streams, frame_t,
                       (const int16_t *)in, hist, Xs, Ps, feat, silence, last_period, last_gain);
  else
    hipLaunchKernelGGL(pn_frontend_kernel<float>, dim3(grid), dim3(DSP_THREADS), 0, st, T, n_streams, frame_t,
                       (const float *)in, hist, Xs, Ps, feat, silence, last_period, last_gain);
}

void pn_launch_backend(hipStream_t st, const PnTables *T, int n_streams, const float2 *Xs, const float2 *Ps,
                       const float *gr, const int *silence, float *synth_mem, void *out, int out_is_i16) {
  const int grid = pn_dsp_grid(n_streams, 0);
  if (out_is_i16)
    hipLaunchKernelGGL(pn_backend_kernel<int16_t>, dim3(grid), dim3(DSP_THREADS), 0, st, T, n_streams, Xs, Ps, gr, silence,
                       synth_mem, (int16_t *)out);
  else
    hipLaunchKernelGGL(pn_backend_kernel<float>, dim3(grid), dim3(DSP_THREADS), 0, st, T, n_streams, Xs, Ps, gr, silence,
                       synth_mem, (float *)out);
}
