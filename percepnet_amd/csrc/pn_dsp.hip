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
// sequential chain on one lane — one lane per lag/band, never a shuffle tree.
//
// Work distribution: blocks are a single wavefront (so __syncthreads() is a wave-local LDS
// fence); a block stages the shared read-only tables (twiddles, window, digit-reversal, band
// map: 13.6 KB) into LDS once and then loops over streams  s = blockIdx.x, += gridDim.x.
#include "pn_common.h"

#define LANES 64

struct PnDspShared {
  float2 tw[PN_NFFT];            // 7680 B
  float2 fft[PN_NFFT];           // 7680 B  FFT work buffer; pitch scratch aliases it
  float2 X[PN_SPEC_BINS];        // 3200 B  spectrum of the frame being enhanced
  float win[PN_FRAME];           // 1920 B
  float frac[PN_SPEC_BINS];      // 1600 B
  int16_t bitrev[PN_NFFT];       // 1920 B
  int16_t border[PN_NB + 2];
  uint8_t band[PN_SPEC_BINS];
  float e[4][PN_NB + 2];         // Ex, Ep, Exp, gains scratch
  float comb_w[8];
};

__device__ __forceinline__ void pn_stage_tables(PnDspShared &S, const PnTables *__restrict__ T) {
  const int lane = threadIdx.x;
  for (int i = lane; i < PN_NFFT; i += LANES) {
    S.tw[i] = make_float2(T->tw[2 * i], T->tw[2 * i + 1]);
    S.bitrev[i] = T->bitrev[i];
  }
  for (int i = lane; i < PN_FRAME; i += LANES) S.win[i] = T->half_window[i];
  for (int i = lane; i < PN_SPEC_BINS; i += LANES) { S.frac[i] = T->bin_frac[i]; S.band[i] = T->bin_band[i]; }
  if (lane < PN_NB + 2) S.border[lane] = T->border[lane];
  if (lane < 8) S.comb_w[lane] = T->comb_hann[lane];
  __syncthreads();
}

// ---- 960-point FFT in LDS (opus_fft_impl, kiss_fft.cpp:518-564, factors 5,3,4,4,4) -----------
// Input must already be scaled by 1/960 and digit-reverse scattered (opus_fft_c 578-585).
#define CMUL(m, a, b) do { (m).x = (a).x*(b).x - (a).y*(b).y; (m).y = (a).x*(b).y + (a).y*(b).x; } while (0)

__device__ __forceinline__ void pn_fft960_lds(float2 *F, const float2 *tw) {
  const int lane = threadIdx.x;
  __syncthreads();
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
  __syncthreads();
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
    __syncthreads();
  }
  // radix-3, m=64, fstride 5, mm 192 (kiss_fft.cpp:196-227); epi3 = tw[fstride*m]
  {
    const float epi3 = tw[320].y;
    for (int b = lane; b < 320; b += LANES) {
      const int i = b / 64, j = b % 64;
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
    __syncthreads();
  }
  // radix-5, m=192, fstride 1 (kiss_fft.cpp:259-304); ya = tw[m], yb = tw[2m]
  {
    const float2 ya = tw[192], yb = tw[384];
    for (int u = lane; u < 192; u += LANES) {
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
    __syncthreads();
  }
}

// ---- band reductions (denoise.cpp:89-160): lane b owns band b, sums in the reference's order ---
template <bool CORR>
__device__ __forceinline__ float pn_band_reduce(const PnDspShared &S, const float2 *A, const float2 *Bc) {
  const int b = threadIdx.x;
  float sum = 0;
  if (b < PN_NB) {
    if (b >= 1) {   // contributions `sum[i+1] += frac*tmp` of interval i = b-1
      const int lo = S.border[b - 1], hi = S.border[b];
      for (int k = lo; k < hi; k++) {
        float tmp;
        if (CORR) { tmp = A[k].x * Bc[k].x; tmp += A[k].y * Bc[k].y; }
        else      { tmp = A[k].x * A[k].x;  tmp += A[k].y * A[k].y; }
        sum += S.frac[k] * tmp;
      }
    }
    if (b <= PN_NB - 2) {  // contributions `sum[i] += (1-frac)*tmp` of interval i = b
      const int lo = S.border[b], hi = S.border[b + 1];
      for (int k = lo; k < hi; k++) {
        float tmp;
        if (CORR) { tmp = A[k].x * Bc[k].x; tmp += A[k].y * Bc[k].y; }
        else      { tmp = A[k].x * A[k].x;  tmp += A[k].y * A[k].y; }
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
__device__ __forceinline__ void pn_window_scatter(PnDspShared &S, const float *__restrict__ h, int base_slot, int j0) {
  const float scale = 1.f / PN_NFFT;
  for (int i = threadIdx.x; i < PN_WINDOW; i += LANES) {
    const float w = S.win[i < PN_FRAME ? i : PN_WINDOW - 1 - i];
    const float v = h[pn_ring(j0 + i, base_slot)] * w;
    S.fft[S.bitrev[i]] = make_float2(scale * v, scale * 0.f);
  }
}

// find_best_pitch (pitch.cpp:46-104, float instantiation); executed redundantly by every lane
// (wave-uniform control flow, LDS broadcast reads).  y[j] = yb[ystride*j].
__device__ __forceinline__ void pn_find_best_pitch(const float *xcorr, const float *yb, int ystride, int len,
                                                   int max_pitch, int &bp0, int &bp1) {
  float Syy = 1, bn0 = -1, bn1 = -1, bd0 = 0, bd1 = 0;
  bp0 = 0; bp1 = 1;
  for (int j = 0; j < len; j++) { const float v = yb[ystride * j]; Syy = Syy + v * v; }
  for (int i = 0; i < max_pitch; i++) {
    const float xc = xcorr[i];
    if (xc > 0) {
      float x16 = xc;
      x16 *= 1e-12f;
      const float num = x16 * x16;
      if (num * bd1 > bn1 * Syy) {
        if (num * bd0 > bn0 * Syy) { bn1 = bn0; bd1 = bd0; bp1 = bp0; bn0 = num; bd0 = Syy; bp0 = i; }
        else { bn1 = num; bd1 = Syy; bp1 = i; }
      }
    }
    const float a = yb[ystride * (i + len)], c = yb[ystride * i];
    Syy += a * a - c * c;
    Syy = (1 > Syy) ? 1 : Syy;
  }
}

__device__ __forceinline__ float pn_pitch_gain(float xy, float xx, float yy) { return xy / sqrtf(1 + xx * yy); }

template <typename TIn>
__global__ __launch_bounds__(LANES) void pn_frontend_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t,
    const TIn *__restrict__ in,           // [n_streams][480]
    float *__restrict__ hist,             // [n_streams][12][480] ring
    float2 *__restrict__ Xspec,           // [n_streams][400]
    float2 *__restrict__ Pspec,           // [n_streams][400]
    float *__restrict__ feat,             // [n_streams][PN_FEAT_STRIDE]
    int *__restrict__ silence,            // [n_streams]
    int *__restrict__ last_period, float *__restrict__ last_gain) {
  __shared__ PnDspShared S;
  const int lane = threadIdx.x;
  pn_stage_tables(S, T);
  const int new_slot = frame_t % PN_HIST_FRAMES;
  const int base_slot = (frame_t + 1) % PN_HIST_FRAMES;   // slot of logical frame 0 (oldest)
  float *pbuf = reinterpret_cast<float *>(S.fft);         // [864]  pitch scratch aliases the FFT buffer
  float *xcorr = pbuf + 864;                              // [294]
  float *yyl = xcorr + 296;                               // [385]

  for (int s = blockIdx.x; s < n_streams; s += gridDim.x) {
    float *h = hist + (size_t)s * PN_HIST;
    // -- history: the shift+append of denoise.cpp:388-389 becomes one ring-slot write ---------
    for (int i = lane; i < PN_FRAME; i += LANES) {
      float v;
      if (sizeof(TIn) == 2) v = ((float)in[(size_t)s * PN_FRAME + i]) / 32768.f;   // main.cpp:34
      else v = (float)in[(size_t)s * PN_FRAME + i];
      h[new_slot * PN_FRAME + i] = v;
    }
    __syncthreads();
    // -- X = FFT(window(comb_buf[2400,3360))), Ex (frame_analysis 333-346) ---------------------
    pn_window_scatter(S, h, base_slot, 2400);
    pn_fft960_lds(S.fft, S.tw);
    for (int k = lane; k < PN_SPEC_BINS; k += LANES) { S.X[k] = S.fft[k]; Xspec[(size_t)s * PN_SPEC_BINS + k] = S.fft[k]; }
    const float Ex = pn_band_reduce<false>(S, S.fft, nullptr);
    __syncthreads();
    // -- look-ahead band energy of the newest 960 samples (498-506) ------------------------------
    pn_window_scatter(S, h, base_slot, PN_HIST - PN_WINDOW);
    pn_fft960_lds(S.fft, S.tw);
    const float Ey = pn_band_reduce<false>(S, S.fft, nullptr);
    __syncthreads();

    // -- pitch_downsample (pitch.cpp:148-216) of pitch_buf == comb_buf[1632,3360) ----------------
    for (int i = lane; i < 864; i += LANES) {
      float v;
      if (i == 0) v = .5f * (.5f * (h[pn_ring(1632 + 1, base_slot)]) + h[pn_ring(1632, base_slot)]);
      else v = .5f * (.5f * (h[pn_ring(1632 + 2 * i - 1, base_slot)] + h[pn_ring(1632 + 2 * i + 1, base_slot)]) +
                      h[pn_ring(1632 + 2 * i, base_slot)]);
      pbuf[i] = v;
    }
    __syncthreads();
    // _celt_autocorr (celt_lpc.cpp:198-279): lane k holds lag k, sequential chains
    float ack = 0;
    if (lane <= 4) {
      for (int j = 0; j < 860; j++) ack = ack + pbuf[j] * pbuf[j + lane];
      float d = 0;
      for (int i = lane + 860; i < 864; i++) d = d + pbuf[i] * pbuf[i - lane];
      ack += d;
    }
    float ac[5];
#pragma unroll
    for (int k = 0; k < 5; k++) ac[k] = __shfl(ack, k);
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
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 14; it++) { const int i = lane + LANES * it; if (i < 864) pbuf[i] = y[it]; }
      __syncthreads();
    }

    // -- pitch_search (pitch.cpp:283-386): x_lp = pbuf+384, y = pbuf, len 960, max_pitch 588 ----
    // coarse: x_lp4[j] = pbuf[384+2j], y_lp4[j] = pbuf[2j]; one lane per lag, j-ascending chain
    for (int i = lane; i < 147; i += LANES) {
      float sum = 0;
      for (int j = 0; j < 240; j++) sum = sum + pbuf[384 + 2 * j] * pbuf[2 * (i + j)];
      xcorr[i] = sum;
    }
    __syncthreads();
    int bp0, bp1;
    pn_find_best_pitch(xcorr, pbuf, 2, 240, 147, bp0, bp1);
    __syncthreads();
    // fine: only lags within +-2 of 2*best (pitch.cpp:344-361); other entries are 0
    for (int i = lane; i < 294; i += LANES) xcorr[i] = 0;
    __syncthreads();
    {
      const int c = (lane < 5) ? (2 * bp0 - 2 + lane) : (2 * bp1 - 2 + (lane - 5));
      if (lane < 10 && c >= 0 && c < 294) {
        float sum = 0;
        for (int j = 0; j < 480; j++) sum = sum + pbuf[384 + j] * pbuf[c + j];
        xcorr[c] = (-1 > sum) ? -1 : sum;   // duplicates (overlapping windows) write the same value
      }
    }
    __syncthreads();
    pn_find_best_pitch(xcorr, pbuf, 1, 480, 294, bp0, bp1);
    int offset = 0;
    if (bp0 > 0 && bp0 < 294 - 1) {
      const float a = xcorr[bp0 - 1], b = xcorr[bp0], c = xcorr[bp0 + 1];
      if ((c - a) > .7f * (b - a)) offset = 1;
      else if ((a - c) > .7f * (b - c)) offset = -1;
    }
    const float pitch_corr = xcorr[bp0];
    int pitch_index = PN_PITCH_MAX - (2 * bp0 - offset);       // denoise.cpp:408
    __syncthreads();

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
      bool active = false;
      const int k = (lane >= 16) ? lane - 14 : lane;            // lanes 2..15 and 16..29 -> k = 2..15
      if (lane == 0) { lag = 0; active = true; }
      else if (lane == 1) { lag = T0; active = true; }
      else if (lane < 30) {
        static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
        T1 = (2 * T0 + k) / (2 * k);
        if (k == 2) { if (T1 + T0 > 384) T1b = T0; else T1b = T0 + T1; }
        else T1b = (2 * second_check[k] * T0 + k) / (2 * k);
        lag = (lane < 16) ? T1 : T1b;
        active = true;
      }
      float dot = 0;
      if (active) for (int i = 0; i < 480; i++) dot = dot + x[i] * x[i - lag];
      const float xx = __shfl(dot, 0);
      float xy = __shfl(dot, 1);
      // yy_lookup (pitch.cpp:449-455): strictly sequential running energy, wave-uniform
      {
        float yy = xx;
        if (lane == 0) yyl[0] = xx;
        for (int i = 1; i <= 384; i++) {
          yy = yy + x[-i] * x[-i] - x[480 - i] * x[480 - i];
          if (lane == 0) yyl[i] = (0 > yy) ? 0 : yy;
        }
      }
      __syncthreads();
      float yy = yyl[T0];
      float best_xy = xy, best_yy = yy;
      const float g0 = pn_pitch_gain(xy, xx, yy);
      float g = g0;
      int Tsel = T0;
      // k = 2..15 evaluated in parallel on lanes 2..15; the sequential loop's "last hit wins"
      // becomes "highest k among hits"; its `break` at T1 < minperiod is a prefix condition.
      const float xy2 = __shfl(dot, lane + 14);     // partner lane holds xy2 for the same k
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
      float xc = 0;
      if (lane < 3) for (int i = 0; i < 480; i++) xc = xc + x[i] * x[i - (Tsel + lane - 1)];
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
    __syncthreads();

    // -- comb filter (denoise.cpp:416-422) + window + FFT -> P, Ep, Exp -------------------------
    {
      const float scale = 1.f / PN_NFFT;
      for (int i = lane; i < PN_WINDOW; i += LANES) {
        float p = 0;
#pragma unroll
        for (int k = -PN_COMB_M; k <= PN_COMB_M; k++)
          p += h[pn_ring(2400 - pitch_index * k + i, base_slot)] * S.comb_w[k + PN_COMB_M];
        const float v = p * S.win[i < PN_FRAME ? i : PN_WINDOW - 1 - i];
        S.fft[S.bitrev[i]] = make_float2(scale * v, scale * 0.f);
      }
    }
    pn_fft960_lds(S.fft, S.tw);
    for (int k = lane; k < PN_SPEC_BINS; k += LANES) Pspec[(size_t)s * PN_SPEC_BINS + k] = S.fft[k];
    const float Ep = pn_band_reduce<false>(S, S.fft, nullptr);
    float Exp = pn_band_reduce<true>(S, S.X, S.fft);
    if (lane < PN_NB) {
      // double island, denoise.cpp:427
      Exp = (float)fmin(1.0, fmax(0.0, (double)Exp / sqrt(1e-15 + (double)(Ex * Ep))));
      S.e[0][lane] = Ex;
    }
    __syncthreads();
    // silence = sum(Ex) < 0.1 (429-433): sequential sum, wave-uniform
    if (lane == 0) {
      float E = 0;
      for (int i = 0; i < PN_NB; i++) E += S.e[0][i];
      silence[s] = ((double)E < 0.1) ? 1 : 0;
    }
    // -- create_features (487-496) -----------------------------------------------------------------
    float *f = feat + (size_t)s * PN_FEAT_STRIDE;
    if (lane < PN_NB) { f[lane] = Ey * 30; f[PN_NB + lane] = Exp * 30; }
    if (lane == 0) { f[68] = (float)pitch_index / (PN_PITCH_MAX - 3 * PN_PITCH_MIN); f[69] = pitch_corr; }
    __syncthreads();
  }
}

// float -> int16 as the reference CLI's x86-64 build does it (main.cpp:36): truncate toward zero
// to int32 (cvttss2si; NaN / out of range -> 0x80000000), keep the low 16 bits.
__device__ __forceinline__ int16_t pn_f2s(float v) {
  const int32_t t = (fabsf(v) < 2147483648.f) ? (int32_t)v : (int32_t)0x80000000;
  return (int16_t)(uint16_t)((uint32_t)t & 0xffffu);
}

template <typename TOut>
__global__ __launch_bounds__(LANES) void pn_backend_kernel(
    const PnTables *__restrict__ T, int n_streams,
    const float2 *__restrict__ Xspec, const float2 *__restrict__ Pspec,
    const float *__restrict__ gr,          // [n_streams][68]  g | r
    const int *__restrict__ silence,
    float *__restrict__ synth_mem,         // [n_streams][480]
    TOut *__restrict__ out) {              // [n_streams][480]
  __shared__ PnDspShared S;
  const int lane = threadIdx.x;
  pn_stage_tables(S, T);
  const float scale = 1.f / PN_NFFT;
  for (int s = blockIdx.x; s < n_streams; s += gridDim.x) {
    if (lane < PN_NB) {
      const float g = gr[(size_t)s * 68 + lane], r = gr[(size_t)s * 68 + PN_NB + lane];
      S.e[0][lane] = g; S.e[1][lane] = r; S.e[2][lane] = 1 - r;
    }
    __syncthreads();
    const bool sil = silence[s] != 0;
    // pitch_filter (436-485, skipped when silent, 536-538), gain (539-544), then the Hermitian
    // extension + scale + digit-reverse scatter of inverse_transform (306-317).  Bins >= 400
    // are exactly 0 (interp_band_gain never writes them).
    for (int i = lane; i < PN_WINDOW; i += LANES) {
      const int k = (i <= PN_FRAME) ? i : PN_WINDOW - i;
      float2 x = make_float2(0.f, 0.f);
      if (k < PN_SPEC_BINS) {
        x = Xspec[(size_t)s * PN_SPEC_BINS + k];
        const int b = S.band[k];
        const float fr = S.frac[k];
        if (!sil) {
          const float2 p = Pspec[(size_t)s * PN_SPEC_BINS + k];
          const float rf1 = (1 - fr) * S.e[2][b] + fr * S.e[2][b + 1];
          x.x = rf1 * x.x; x.y = rf1 * x.y;
          const float rf2 = (1 - fr) * S.e[1][b] + fr * S.e[1][b + 1];
          x.x += rf2 * p.x; x.y += rf2 * p.y;
        }
        const float gf = (1 - fr) * S.e[0][b] + fr * S.e[0][b + 1];
        x.x *= gf; x.y *= gf;
      }
      if (i > PN_FRAME) x.y = -x.y;
      S.fft[S.bitrev[i]] = make_float2(scale * x.x, scale * x.y);
    }
    pn_fft960_lds(S.fft, S.tw);
    // reversed read-out x960 (318-323), window, overlap-add (352-359)
    float *sm = synth_mem + (size_t)s * PN_FRAME;
    for (int i = lane; i < PN_FRAME; i += LANES) {
      const float t_lo = (PN_WINDOW * S.fft[i == 0 ? 0 : PN_WINDOW - i].x) * S.win[i];
      const int i2 = PN_FRAME + i;                       // second half, window index 959 - i2
      const float t_hi = (PN_WINDOW * S.fft[PN_WINDOW - i2].x) * S.win[PN_WINDOW - 1 - i2];
      const float o = t_lo + sm[i];
      sm[i] = t_hi;
      if (sizeof(TOut) == 2) out[(size_t)s * PN_FRAME + i] = (TOut)pn_f2s(o * 32768);
      else out[(size_t)s * PN_FRAME + i] = (TOut)o;
    }
    __syncthreads();
  }
}

// ---- launchers -------------------------------------------------------------------------------
static inline int pn_dsp_grid(int n_streams) {
  const int cap = 256 * 6;   // 256 CUs x 6 resident single-wave blocks (LDS-limited)
  return n_streams < cap ? n_streams : cap;
}

void pn_launch_frontend(hipStream_t st, const PnTables *T, int n_streams, int frame_t, const void *in, int in_is_i16,
                        float *hist, float2 *Xs, float2 *Ps, float *feat, int *silence, int *last_period,
                        float *last_gain) {
  const int grid = pn_dsp_grid(n_streams);
  if (in_is_i16)
    hipLaunchKernelGGL(pn_frontend_kernel<int16_t>, dim3(grid), dim3(LANES), 0, st, T, n_streams, frame_t,
                       (const int16_t *)in, hist, Xs, Ps, feat, silence, last_period, last_gain);
  else
    hipLaunchKernelGGL(pn_frontend_kernel<float>, dim3(grid), dim3(LANES), 0, st, T, n_streams, frame_t,
                       (const float *)in, hist, Xs, Ps, feat, silence, last_period, last_gain);
}

void pn_launch_backend(hipStream_t st, const PnTables *T, int n_streams, const float2 *Xs, const float2 *Ps,
                       const float *gr, const int *silence, float *synth_mem, void *out, int out_is_i16) {
  const int grid = pn_dsp_grid(n_streams);
  if (out_is_i16)
    hipLaunchKernelGGL(pn_backend_kernel<int16_t>, dim3(grid), dim3(LANES), 0, st, T, n_streams, Xs, Ps, gr, silence,
                       synth_mem, (int16_t *)out);
  else
    hipLaunchKernelGGL(pn_backend_kernel<float>, dim3(grid), dim3(LANES), 0, st, T, n_streams, Xs, Ps, gr, silence,
                       synth_mem, (float *)out);
}
