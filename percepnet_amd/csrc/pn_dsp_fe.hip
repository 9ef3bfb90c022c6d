// Front end of the PercepNet frame engine for gfx950: history ring write, window + 960-pt FFT
// (look-ahead and comb-filtered frame), ERB band energies / correlation, CELT pitch analysis
// (downsample + LPC whitening, coarse/fine xcorr search, octave-error removal), 7-tap comb filter,
// 70 features == compute_frame_features + compute_lookahead_band_energy + create_features
// (reference denoise.cpp:372-434, 498-506, 487-496).
//
// Mapping: FOUR streams per wavefront, 16 lanes each ("group").  The reference's per-frame work is
// dominated by short, strictly sequential float chains (inner products per pitch lag, running
// energies, Levinson, the best-pitch scan) that need 1..30 lanes; with one stream per wave most
// of the 64 lanes execute redundant copies.  With 16 lanes per stream every wave instruction of
// those phases advances four streams, and the data-parallel phases (FFT butterflies, windows,
// comb filter) simply take 4x the iterations at 16 lanes each — same work per stream.
//
// Numerics contract (unchanged): every arithmetic step is the reference's operation in the
// reference's order with separate IEEE binary32 rounding (-ffp-contract=off; divide/sqrt
// correctly rounded; the reference's double islands in double).  Order-sensitive reductions run
// as the reference's sequential chain on one lane (lane = lag / band); only the adds of a chain
// are serially dependent, operands are fetched ahead.  Chain operands that are uniform within a
// group (the x every lag is correlated against, the squares of the running-energy recurrences) are
// read with one ds_read_b128 per 4 steps at a group-uniform address.  Features, silence flags,
// pitch decisions are bit-identical to the CPU reference (tests/test_gpu_parity.py).
//
// One forward FFT per frame is saved exactly: the analysis window of frame t (comb_buf[2400,3360))
// holds the same samples, window and transform as the look-ahead window of frame t-5
// (comb_buf[4800,5760) then), so X(t) == Y(t-5) and Ex(t) == Ey(t-5) bit for bit; the look-ahead
// spectra are kept in a 6-slot ring in HBM (slot t%6 written, slot (t+1)%6 = Y(t-5) read; the
// back end reads that slot too).
//
// Work distribution: 256-thread blocks = 4 waves = 16 concurrent streams; the block stages the
// shared tables (twiddles, window, digit reversal, band map: 13.6 KB) into LDS once; every
// stream owns a private 8.3 KB LDS slice (FFT buffer, aliased by the pitch scratch); one block
// per CU (146 KB LDS), grid-stride over stream quartets.  No block barrier after the staging:
// groups never exchange data, LDS operations of one wave execute in order, PN_WAVE_SYNC is a
// compiler fence.
#include "pn_dsp_fe_helpers.inc"
#include "pn_launch.h"

struct alignas(16) FeStreamLds {
  float2 fft[PN_NFFT];           // 7680 B  FFT work buffer; pitch scratch / per-bin products alias it
  float e[4][PN_NB + 2];         // 576 B
};
struct FeShared {
  FeTablesLds t;
  FeStreamLds s[FE_SPB];
};

// float offsets inside a stream's 1920-float buffer during the pitch section (see kernel)
#define OFF_XCORR 0      // [0,296)    xcorr[294]; later p|q scratch of yy_lookup (128)
#define OFF_Y4 304       // [304,691)  y_lp4[387]; later d[] of the fine pass (296)
#define OFF_YYL 307      // [307,692)  yy_lookup[385]; &yyl[1] is 16-byte aligned
#define OFF_SQ 704       // [704,768)  64-float broadcast scratch
#define OFF_RAW 0        // [0,864)    decimated signal before the whitening FIR
#define OFF_PBUF 864     // [864,1728) whitened decimated signal  (pitch_buf>>1)
#define OFF_D1 1728      // [1728,1876) d[] of the coarse pass (148)
#define OFF_PROD 960     // [960,1360) per-bin X.P products (after the P FFT; bins live in [0,800))

template <typename TIn>
#ifndef PN_FE_WAVES_PER_SIMD
#define PN_FE_WAVES_PER_SIMD 1
#endif
__global__ __launch_bounds__(FE_THREADS, PN_FE_WAVES_PER_SIMD) void pn_frontend_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t, int slot_w, int slot_r,
    const TIn *__restrict__ in,           // stream s's frame at in + s*in_stride (480 contiguous samples)
    long long in_stride, float i16_scale, // int16 input: sample = (float)v * i16_scale (2^-15: main.cpp:34; 1: denoise.cpp:41,697)
    float *__restrict__ hist,             // [n_streams][12][480] ring
    float2 *__restrict__ yring,           // [6][n_streams][400] look-ahead spectra ring
    float *__restrict__ eyring,           // [6][n_streams][36]  look-ahead band energies ring
    float2 *__restrict__ Pspec,           // [n_streams][400]
    float *__restrict__ feat,             // [n_streams][PN_FEAT_STRIDE]
    int *__restrict__ silence,            // [n_streams]
    int *__restrict__ last_period, float *__restrict__ last_gain,
    float *__restrict__ aux) {            // optional [n_streams][PN_AUX_STRIDE]: Ep[34] | Exp[34] (un-scaled) | pitch_corr
  __shared__ FeShared SH;
  const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid >> 6;
  const int sub = lane / L, l = lane % L, gb = sub * L;
  {
    FeTablesLds &S = SH.t;
    for (int i = tid; i < PN_NFFT; i += FE_THREADS) {
      S.tw[i] = make_float2(T->tw[2 * i], T->tw[2 * i + 1]);
      S.bitrev[i] = T->bitrev[i];
    }
    for (int i = tid; i < PN_FRAME; i += FE_THREADS) S.win[i] = T->half_window[i];
    for (int i = tid; i < PN_SPEC_BINS; i += FE_THREADS) { S.frac[i] = T->bin_frac[i]; S.band[i] = T->bin_band[i]; }
    if (tid < PN_NB + 2) S.border[tid] = T->border[tid];
    if (tid < 8) S.comb_w[tid] = T->comb_hann[tid];
#if PN_FE_G == 4
    if (tid < 64) S.bmap[tid] = kFeBandMap[tid >> 2][tid & 3];
#endif
    __syncthreads();
  }
  const FeTablesLds &S = SH.t;
  FeStreamLds &W = SH.s[wave * G + sub];
  float *buf = reinterpret_cast<float *>(W.fft);
  float *pbuf = buf + OFF_PBUF, *xcorr = buf + OFF_XCORR, *y4 = buf + OFF_Y4, *yyl = buf + OFF_YYL,
        *sq64 = buf + OFF_SQ, *d1 = buf + OFF_D1, *prod = buf + OFF_PROD, *raw = buf + OFF_RAW;
  const int new_slot = frame_t % PN_HIST_FRAMES;
  const int base_slot = (frame_t + 1) % PN_HIST_FRAMES;   // slot of logical frame 0 (oldest)
  const float scale = 1.f / PN_NFFT;

  for (int s0 = (blockIdx.x * FE_WPB + wave) * G; s0 < n_streams; s0 += gridDim.x * FE_SPB) {
    const int s = s0 + sub;
#ifdef PN_FE_CLOCKS
    long long tmark_ = __builtin_readcyclecounter();
#endif
    if (s < n_streams) {
      float *h = hist + (size_t)s * PN_HIST_STRIDE;
      // -- history: the shift+append of denoise.cpp:388-389 becomes one ring-slot write ---------
      {
        constexpr int NI = (PN_FRAME / 4 + L - 1) / L;     // float4 per lane (8 at L=16, the last one partial)
        float4 nv[NI];
#pragma unroll
        for (int it = 0; it < NI; it++) {                  // all loads first, then the stores
          const int i4 = l + L * it, i4c = i4 < PN_FRAME / 4 ? i4 : 0;
          if (sizeof(TIn) == 2) {
            const short4 q = *reinterpret_cast<const short4 *>(in + (size_t)s * in_stride + 4 * i4c);
            // a power-of-two scale: the product is exact, == the reference's division (main.cpp:34)
            nv[it] = make_float4(((float)q.x) * i16_scale, ((float)q.y) * i16_scale, ((float)q.z) * i16_scale, ((float)q.w) * i16_scale);
          } else {
            nv[it] = *reinterpret_cast<const float4 *>(in + (size_t)s * in_stride + 4 * i4c);
          }
        }
#pragma unroll
        for (int it = 0; it < NI; it++) {
          const int i4 = l + L * it;
          if (i4 < PN_FRAME / 4) {
            *reinterpret_cast<float4 *>(h + new_slot * PN_FRAME + 4 * i4) = nv[it];
            if (new_slot == 0 && i4 < 2) *reinterpret_cast<float4 *>(h + PN_HIST + 4 * i4) = nv[it];   // mirror of the ring's first 8 samples
          }
        }
      }
      PN_WAVE_SYNC_GLOBAL();
      FE_MARK(0);
      // -- Y = FFT(window(newest 960 samples)), Ey (compute_lookahead_band_energy 498-506); kept in
      //    the ring: it is X / Ex of frame t+5 (frame_analysis 333-346) --------------------------------
      {
        constexpr int NI = (PN_WINDOW / 4 + L - 1) / L;   // float4 per lane (15 at L=16)
        float4 hv[NI];
#pragma unroll
        for (int it = 0; it < NI; it++) {              // all global loads in flight before first use
          const int i4 = l + L * it;
          hv[it] = *reinterpret_cast<const float4 *>(h + fe_ring(PN_HIST - PN_WINDOW + 4 * (i4 < PN_WINDOW / 4 ? i4 : 0), base_slot));
        }
#pragma unroll
        for (int it = 0; it < NI; it++) {
          const int i = 4 * (l + L * it);
          if (i >= PN_WINDOW) continue;
          const float vv[4] = {hv[it].x, hv[it].y, hv[it].z, hv[it].w};
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int ii = i + c;
            const float w = S.win[ii < PN_FRAME ? ii : PN_WINDOW - 1 - ii];   // apply_window 282-289
            W.fft[S.bitrev[ii]] = make_float2(scale * (vv[c] * w), scale * 0.f);
          }
        }
      }
      FE_MARK(1);   // window + scatter
      fe_fft960(W.fft, S.tw, l);
      FE_MARK(2);   // look-ahead FFT
      {
        float2 *yw = yring + ((size_t)slot_w * n_streams + s) * PN_SPEC_BINS;
        for (int k = l; k < PN_SPEC_BINS; k += L) yw[k] = W.fft[k];
        float *ew = eyring + ((size_t)slot_w * n_streams + s) * 36;
#pragma unroll
        for (int c = 0; c < NBND; c++) {
          const int b = FE_BAND_OF(S, l, c);
          const float e = fe_band<false>(S, W.fft, nullptr, b);
          if (b < PN_NB) { ew[b] = e; W.e[1][b] = e; }          // Ey of this frame (features)
        }
      }
      const float2 *Xr = yring + ((size_t)slot_r * n_streams + s) * PN_SPEC_BINS;   // X(t)  = Y(t-5)
      const float *Exr = eyring + ((size_t)slot_r * n_streams + s) * 36;            // Ex(t) = Ey(t-5)
      for (int b = l; b < PN_NB; b += L) W.e[0][b] = Exr[b];
      PN_WAVE_SYNC();
      FE_MARK(3);

#if defined(PN_FE_ABL) && PN_FE_ABL == 1
      continue;   // timing ablation (tools/kernel_times.py): history write + look-ahead FFT + band energies only
#endif
      // -- pitch_downsample (pitch.cpp:148-216) of pitch_buf == comb_buf[1632,3360) ----------------
      // outputs 2m, 2m+1 need x[4m-1 .. 4m+3]
      {
        constexpr int NM = (432 + L - 1) / L;
        float4 dv[NM]; float dm1[NM];
#pragma unroll
        for (int it = 0; it < NM; it++) {
          const int m = (l + L * it < 432) ? l + L * it : 0;
          dv[it] = *reinterpret_cast<const float4 *>(h + fe_ring(1632 + 4 * m, base_slot));
          dm1[it] = h[fe_ring(1632 + (m > 0 ? 4 * m - 1 : 0), base_slot)];
        }
#pragma unroll
        for (int it = 0; it < NM; it++) {
          const int m = l + L * it;
          if (m >= 432) continue;
          const float4 v = dv[it];
          const float o0 = (m == 0) ? .5f * (.5f * (v.y) + v.x) : .5f * (.5f * (dm1[it] + v.y) + v.x);
          const float o1 = .5f * (.5f * (v.y + v.w) + v.z);
          *reinterpret_cast<float2 *>(raw + 2 * m) = make_float2(o0, o1);
        }
      }
      PN_WAVE_SYNC();
      FE_MARK(4);   // downsample
      // _celt_autocorr (celt_lpc.cpp:198-279): lane k holds lag k (lanes > 4 shadow lag 4)
      float ac[5];
      {
        const int lag = l < 4 ? l : 4;
        float ack = fe_chain<860>(raw, raw + lag, 0.f);
        float d = 0;
        for (int i = lag + 860; i < 864; i++) d = d + raw[i] * raw[i - lag];
        ack += d;
#pragma unroll
        for (int k = 0; k < 5; k++) ac[k] = __shfl(ack, gb + k);
      }
      ac[0] *= 1.0001f;
#pragma unroll
      for (int i = 1; i <= 4; i++) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
      // _celt_lpc (celt_lpc.cpp:37-88), p = 4; group-uniform
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
      FE_MARK(5);   // autocorr + LPC
      // celt_fir5 (pitch.cpp:106-145): out of place, raw -> pbuf
#pragma unroll 6
      for (int i = l; i < 864; i += L) {
        float sum = raw[i];
        sum = sum + lpc2[0] * (i >= 1 ? raw[i - 1] : 0.f);
        sum = sum + lpc2[1] * (i >= 2 ? raw[i - 2] : 0.f);
        sum = sum + lpc2[2] * (i >= 3 ? raw[i - 3] : 0.f);
        sum = sum + lpc2[3] * (i >= 4 ? raw[i - 4] : 0.f);
        sum = sum + lpc2[4] * (i >= 5 ? raw[i - 5] : 0.f);
        pbuf[i] = sum;
      }
      PN_WAVE_SYNC();

      FE_MARK(6);   // FIR
#if defined(PN_FE_ABL) && PN_FE_ABL == 2
      continue;   // timing ablation: + downsample, autocorr, LPC, FIR
#endif
      // -- pitch_search (pitch.cpp:283-386): x_lp = pbuf+384, y = pbuf, len 960, max_pitch 588 ----
      // coarse: x_lp4[j] = pbuf[384+2j] (240, group-uniform operand, read straight from pbuf),
      // y_lp4[j] = pbuf[2j] (387, copied out contiguously); lane owns lags l + L*c, c < NCH
      for (int j = l; j < 387; j += L) y4[j] = pbuf[2 * j];
      PN_WAVE_SYNC();
      {
        // two register sets of 8 steps each: the LDS reads of the next 8 steps are in flight while the
        // (per-lag serially dependent) adds of the current 8 execute
        float sacc[NCH];
        int lagc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) { sacc[c] = 0; lagc[c] = (l + L * c <= 146) ? l + L * c : 146; }
#ifndef PN_FE_CXS
#define PN_FE_CXS 4
#endif
        constexpr int CXS = PN_FE_CXS;                 // steps per register set (even, 240 % (2*CXS) == 0)
        static_assert(CXS % 2 == 0 && 240 % (2 * CXS) == 0, "CXS");
        float xa0[CXS], xa1[CXS], yb0[NCH][CXS], yb1[NCH][CXS];
#define FE_CX_LOAD(xa, yb, j0) do {                                                                    \
          _Pragma("unroll") for (int v_ = 0; v_ < CXS / 2; v_++) {                                         \
            const float4 q_ = *reinterpret_cast<const float4 *>(pbuf + 384 + 2 * (j0) + 4 * v_);           \
            (xa)[2 * v_] = q_.x; (xa)[2 * v_ + 1] = q_.z; }                                                \
          _Pragma("unroll") for (int c_ = 0; c_ < NCH; c_++)                                               \
            _Pragma("unroll") for (int u_ = 0; u_ < CXS; u_++) (yb)[c_][u_] = y4[lagc[c_] + (j0) + u_];    \
        } while (0)
#define FE_CX_MAC(xa, yb) do {                                                                         \
          _Pragma("unroll") for (int u_ = 0; u_ < CXS; u_++)                                               \
            _Pragma("unroll") for (int c_ = 0; c_ < NCH; c_++) sacc[c_] = sacc[c_] + (xa)[u_] * (yb)[c_][u_]; \
        } while (0)
        FE_CX_LOAD(xa0, yb0, 0);
#pragma unroll 1
        for (int j0 = 0; j0 < 240; j0 += 2 * CXS) {
          FE_CX_LOAD(xa1, yb1, j0 + CXS);
          FE_CX_MAC(xa0, yb0);
          if (j0 + 2 * CXS < 240) FE_CX_LOAD(xa0, yb0, j0 + 2 * CXS);
          FE_CX_MAC(xa1, yb1);
        }
#undef FE_CX_LOAD
#undef FE_CX_MAC
#pragma unroll
        for (int c = 0; c < NCH; c++) if (l + L * c < 147) xcorr[l + L * c] = sacc[c];
      }
      PN_WAVE_SYNC();
      FE_MARK(7);   // y4 copy + coarse xcorr
      int bp0, bp1;
      fe_find_best_pitch<240, 147>(xcorr, y4, sq64, d1, l, bp0, bp1);
      PN_WAVE_SYNC();
      FE_MARK(8);   // find_best_pitch coarse
      // fine: only lags within +-2 of 2*best (pitch.cpp:344-361); other entries are 0
      for (int i = l; i < 296; i += L) xcorr[i] = 0;
      PN_WAVE_SYNC();
      {
        const int c = (l < 5) ? (2 * bp0 - 2 + l) : (2 * bp1 - 2 + (l - 5));
        const bool act = l < 10 && c >= 0 && c < 294;
        const float sum = fe_chain<480>(pbuf + 384, pbuf + (act ? c : 0), 0.f);
        if (act) xcorr[c] = (-1 > sum) ? -1 : sum;   // duplicates (overlapping windows) write the same value
      }
      PN_WAVE_SYNC();
      FE_MARK(9);   // fine xcorr
      fe_find_best_pitch<480, 294>(xcorr, pbuf, sq64, y4, l, bp0, bp1);   // y_lp4 is dead: its space holds d[]
      int offset = 0;
      if (bp0 > 0 && bp0 < 294 - 1) {
        const float a = xcorr[bp0 - 1], b = xcorr[bp0], c = xcorr[bp0 + 1];
        if ((c - a) > .7f * (b - a)) offset = 1;
        else if ((a - c) > .7f * (b - c)) offset = -1;
      }
      const float pitch_corr = xcorr[bp0];
      int pitch_index = PN_PITCH_MAX - (2 * bp0 - offset);       // denoise.cpp:408
      PN_WAVE_SYNC();

      FE_MARK(10);  // find_best_pitch fine + interp
#if defined(PN_FE_ABL) && PN_FE_ABL == 3
      continue;   // timing ablation: + pitch_search
#endif
      // -- remove_doubling (pitch.cpp:424-527): maxperiod 384, minperiod 30, N 480, x = pbuf+384 -----
      float pg;
      {
        const float *x = pbuf + 384;
        const int prev_period = last_period[s] / 2;
        const float prev_gain = last_gain[s];
        int T0 = pitch_index / 2;
        if (T0 >= 384) T0 = 383;
        // lane 0: xx ; lane 1: xy(T0) ; lanes 2..15: k = l: xy(T1_k) and xy2(T1b_k)
        int lag1 = 0, lag2 = 0, T1 = 0, T1b = 0;
        const int k = l;
        if (l == 1) { lag1 = T0; lag2 = T0; }
        else if (l >= 2 && l < 16) {
          static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
          T1 = (2 * T0 + k) / (2 * k);
          if (k == 2) { if (T1 + T0 > 384) T1b = T0; else T1b = T0 + T1; }
          else T1b = (2 * second_check[k] * T0 + k) / (2 * k);
          lag1 = T1; lag2 = T1b;
        }
        float dot1 = 0, dot2 = 0;
        fe_chain2<480>(x, x - lag1, x - lag2, dot1, dot2);
        const float xx = __shfl(dot1, gb);
        float xy = __shfl(dot1, gb + 1);
        FE_MARK(11);  // remove_doubling: 2x14 dot chains
        // yy_lookup (pitch.cpp:449-455): strictly sequential running energy, group-uniform.  Squares
        // formed lane-parallel, 64 at a time, into a broadcast scratch (the dead xcorr area); the
        // recurrence reads them 4 per ds_read_b128; lane 0 stores the clamped results 4 at a time.
        {
          float *pq = xcorr;                     // p[64] | q[64]
          float yy = xx;
          if (l == 0) yyl[0] = xx;
#pragma unroll 1
          for (int blk = 0; blk < 6; blk++) {        // i = 1 + 64*blk + u, u < 64  (384 = 6*64)
            float pa[64 / L], qa[64 / L];
#pragma unroll
            for (int w = 0; w < 64 / L; w++) {
              const int i = 1 + 64 * blk + l + L * w;
              const float a = x[-i], c = x[480 - i];
              pa[w] = a * a; qa[w] = c * c;
            }
            PN_WAVE_SYNC();
#pragma unroll
            for (int w = 0; w < 64 / L; w++) { pq[l + L * w] = pa[w]; pq[64 + l + L * w] = qa[w]; }
            PN_WAVE_SYNC();
#pragma unroll
            for (int h0 = 0; h0 < 16; h0 += 8) {       // operands of 32 steps read before the chain, results stored after it
              float4 p4[8], q4[8], o[8];
#pragma unroll
              for (int v = 0; v < 8; v++) {
                p4[v] = *reinterpret_cast<const float4 *>(pq + 4 * (h0 + v));
                q4[v] = *reinterpret_cast<const float4 *>(pq + 64 + 4 * (h0 + v));
              }
#pragma unroll
              for (int v = 0; v < 8; v++) {
                yy = yy + p4[v].x - q4[v].x; o[v].x = (0 > yy) ? 0 : yy;
                yy = yy + p4[v].y - q4[v].y; o[v].y = (0 > yy) ? 0 : yy;
                yy = yy + p4[v].z - q4[v].z; o[v].z = (0 > yy) ? 0 : yy;
                yy = yy + p4[v].w - q4[v].w; o[v].w = (0 > yy) ? 0 : yy;
              }
              if (l == 0) {
#pragma unroll
                for (int v = 0; v < 8; v++) *reinterpret_cast<float4 *>(yyl + 1 + 64 * blk + 4 * (h0 + v)) = o[v];
              }
            }
          }
        }
        PN_WAVE_SYNC();
        FE_MARK(12);  // yy_lookup recurrence
        float yy = yyl[T0];
        float best_xy = xy, best_yy = yy;
        const float g0 = fe_pitch_gain(xy, xx, yy);
        float g = g0;
        int Tsel = T0;
        // k = 2..15 evaluated in parallel on lanes 2..15 of the group; the sequential loop's "last hit
        // wins" becomes "highest k among hits"; its `break` at T1 < minperiod is a prefix condition.
        bool hit = false;
        float xyk = 0, yyk = 0, g1 = 0;
        if (l >= 2 && l < 16 && T1 >= 30) {
          xyk = .5f * (dot1 + dot2);
          yyk = .5f * (yyl[T1] + yyl[T1b]);
          g1 = fe_pitch_gain(xyk, xx, yyk);
          float cont;
          const int dT = (T1 - prev_period) < 0 ? -(T1 - prev_period) : (T1 - prev_period);
          if (dT <= 1) cont = prev_gain;
          else if (dT <= 2 && 5 * k * k < T0) cont = .5f * prev_gain;
          else cont = 0;
          float thresh = (.3f > .7f * g0 - cont) ? .3f : .7f * g0 - cont;
          if (T1 < 3 * 30) thresh = (.4f > .85f * g0 - cont) ? .4f : .85f * g0 - cont;
          hit = g1 > thresh;
        }
        const unsigned m = (unsigned)((__ballot(hit) >> gb) & 0xffffull);   // hits live on lanes 2..15 of the group
        if (m) {
          const int win = gb + 31 - __clz(m);
          best_xy = __shfl(xyk, win); best_yy = __shfl(yyk, win);
          Tsel = __shfl(T1, win); g = __shfl(g1, win);
        }
        best_xy = (0 > best_xy) ? 0 : best_xy;
        if (best_yy <= best_xy) pg = 1.0f; else pg = best_xy / (best_yy + 1);
        const float xc = fe_chain<480>(x, x - (Tsel + (l < 3 ? l : 2) - 1), 0.f);
        const float xc0 = __shfl(xc, gb), xc1 = __shfl(xc, gb + 1), xc2 = __shfl(xc, gb + 2);
        int off2;
        if ((xc2 - xc0) > .7f * (xc1 - xc0)) off2 = 1;
        else if ((xc0 - xc2) > .7f * (xc1 - xc2)) off2 = -1;
        else off2 = 0;
        if (pg > g) pg = g;
        pitch_index = 2 * Tsel + off2;
        if (pitch_index < PN_PITCH_MIN) pitch_index = PN_PITCH_MIN;
      }
      if (l == 0) { last_period[s] = pitch_index; last_gain[s] = pg; }
      PN_WAVE_SYNC();

      FE_MARK(13);  // decisions + final 3 chains
#if defined(PN_FE_ABL) && PN_FE_ABL == 4
      continue;   // timing ablation: + remove_doubling
#endif
      // -- comb filter (denoise.cpp:416-422) + window + FFT -> P, Ep, Exp -------------------------
      {
        // each lane filters 4 consecutive samples per step: one (unaligned) dwordx4 load per tap instead of four
        // dword loads — the phase is bound by the number of VMEM instructions a wave can issue, not by bytes
#ifndef PN_FE_COMB_CH
#define PN_FE_COMB_CH (L == 16 ? 5 : 4)
#endif
        constexpr int CH = PN_FE_COMB_CH;               // 4-sample groups per lane per chunk: 7*CH dwordx4 loads in flight
        constexpr int NGT = PN_WINDOW / 4;              // 240 groups of 4 samples
        constexpr int NG = (NGT + L - 1) / L;           // groups per lane (15 at L=16; 8 at L=32, the last one half empty)
        constexpr bool RAGGED = NGT % L != 0;
        static_assert(NG % CH == 0, "chunk");
#pragma unroll 1
        for (int q0 = 0; q0 < NG; q0 += CH) {
          fe_f4u cv[CH][7];
#pragma unroll
          for (int q = 0; q < CH; q++) {
            const int gi = l + L * (q0 + q), gc = (RAGGED && gi >= NGT) ? 0 : gi;
#pragma unroll
            for (int k = -PN_COMB_M; k <= PN_COMB_M; k++)
              cv[q][k + PN_COMB_M] = *reinterpret_cast<const fe_f4u *>(h + fe_ring(2400 - pitch_index * k + 4 * gc, base_slot));
          }
#pragma unroll
          for (int q = 0; q < CH; q++) {
            const int gi = l + L * (q0 + q);
            if (RAGGED && gi >= NGT) continue;
#pragma unroll
            for (int c = 0; c < 4; c++) {
              const int i = 4 * gi + c;
              float p = 0;
#pragma unroll
              for (int k = 0; k < 7; k++) p += cv[q][k][c] * S.comb_w[k];
              const float v = p * S.win[i < PN_FRAME ? i : PN_WINDOW - 1 - i];
              W.fft[S.bitrev[i]] = make_float2(scale * v, scale * 0.f);
            }
          }
        }
      }
      FE_MARK(14);  // comb filter + window + scatter
      fe_fft960(W.fft, S.tw, l);
      FE_MARK(15);  // P FFT
      {
        constexpr int NK = PN_SPEC_BINS / L;            // 25 at L=16
        float2 xv[NK];
#pragma unroll
        for (int it = 0; it < NK; it++) xv[it] = Xr[l + L * it];
#pragma unroll
        for (int it = 0; it < NK; it++) {
          const int k = l + L * it;
          const float2 P = W.fft[k];
          Pspec[(size_t)s * PN_SPEC_BINS + k] = P;
          float tmp = xv[it].x * P.x;            // compute_band_corr's per-bin term (denoise.cpp:136-137)
          tmp += xv[it].y * P.y;
          prod[k] = tmp;
        }
        for (int k = l + L * NK; k < PN_SPEC_BINS; k += L) {   // remainder when L does not divide 400
          const float2 P = W.fft[k]; const float2 X = Xr[k];
          Pspec[(size_t)s * PN_SPEC_BINS + k] = P;
          float tmp = X.x * P.x; tmp += X.y * P.y; prod[k] = tmp;
        }
      }
      FE_MARK(16);  // Pspec store + X.P products
      float Ep[NBND];
#pragma unroll
      for (int c = 0; c < NBND; c++) Ep[c] = fe_band<false>(S, W.fft, nullptr, FE_BAND_OF(S, l, c));
      PN_WAVE_SYNC();
      FE_MARK(17);  // Ep bands
      float *f = feat + (size_t)s * PN_FEAT_STRIDE;
#pragma unroll
      for (int c = 0; c < NBND; c++) {
        const int b = FE_BAND_OF(S, l, c);
        float Exp = fe_band<true>(S, nullptr, prod, b);
        if (b < PN_NB) {
          const float Ex = W.e[0][b];
          // double island, denoise.cpp:427
          Exp = (float)fmin(1.0, fmax(0.0, (double)Exp / sqrt(1e-15 + (double)(Ex * Ep[c]))));
          f[b] = W.e[1][b] * 30;              // create_features (487-496)
          f[PN_NB + b] = Exp * 30;
          if (aux) { aux[(size_t)s * PN_AUX_STRIDE + b] = Ep[c]; aux[(size_t)s * PN_AUX_STRIDE + PN_NB + b] = Exp; }
        }
      }
      // silence = sum(Ex) < 0.1 (429-433): sequential sum
      if (l == 0) {
        float E = 0;
        for (int i = 0; i < PN_NB; i++) E += W.e[0][i];
        silence[s] = ((double)E < 0.1) ? 1 : 0;
        f[68] = (float)pitch_index / (PN_PITCH_MAX - 3 * PN_PITCH_MIN);
        f[69] = pitch_corr;
        if (aux) aux[(size_t)s * PN_AUX_STRIDE + 2 * PN_NB] = pitch_corr;
      }
      PN_WAVE_SYNC();
      FE_MARK(18);  // Exp bands + features
    }
  }
}

// ---- launcher ---------------------------------------------------------------------------------
void pn_launch_frontend(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in,
                        int in_is_i16, long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring,
                        float2 *Ps, float *feat, int *silence, int *last_period, float *last_gain, float *aux, int grid_cap) {
  const int need = (n_streams + FE_SPB - 1) / FE_SPB;
  const int cap = 256 * (16 / FE_SPB);                 // LDS-resident blocks on 256 CUs
  int grid = need < cap ? need : cap;                  // grid-stride
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  const int frame_t = (int)(frame % PN_HIST_FRAMES);
  const int slot_w = (int)(frame % 6), slot_r = (int)((frame + 1) % 6);
  if (in_is_i16)
    hipLaunchKernelGGL(pn_frontend_kernel<int16_t>, dim3(grid), dim3(FE_THREADS), 0, st, T, n_streams, frame_t,
                       slot_w, slot_r, (const int16_t *)in, in_stride, i16_scale, hist, yring, eyring, Ps, feat, silence,
                       last_period, last_gain, aux);
  else
    hipLaunchKernelGGL(pn_frontend_kernel<float>, dim3(grid), dim3(FE_THREADS), 0, st, T, n_streams, frame_t, slot_w,
                       slot_r, (const float *)in, in_stride, i16_scale, hist, yring, eyring, Ps, feat, silence, last_period,
                       last_gain, aux);
}
