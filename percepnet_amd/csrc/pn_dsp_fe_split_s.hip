// Phase-split front end, spectral kernels (gfx950): ONE stream per wavefront, 64 lanes, four waves per SIMD.
//
//   pn_fe_spec_in_kernel   history ring write, window + 960-pt FFT of the newest 960 samples -> look-ahead spectrum Y
//                          and band energies Ey into their 6-slot rings (== the analysis side of frame_analysis for
//                          frame t+5 and compute_lookahead_band_energy, reference denoise.cpp:333-346, 498-506)
//   pn_fe_spec_out_kernel  7-tap comb filter at the pitch period the pitch kernel found, window + FFT -> P, Ep, the
//                          X.P band correlation, the 68 band features, silence flag
//                          (denoise.cpp:416-434, create_features 487-496)
// The pitch analysis between the two lives in pn_dsp_fe_split_p.hip (four streams per wave).
//
// Why this shape.  Measured on MI355X (profiles/r03a_valu_issue_probe.log, r03c PMC): a lone wave issues one instruction
// per 4.5 cycles whatever its kind, two waves per SIMD double that and four reach the VALU peak — so the data-parallel
// phases want one stream per wave and four waves per SIMD (<= 128 registers, 8 KB of LDS per stream) — and from there the
// CU-wide LDS pipe is what limits (first split version: 54 % of its LDS cycles were bank conflicts, 2.6 k LDS cycles
// per stream-FFT-and-bands).  Hence:
//   * the 960-point FFT runs as THREE register-fused passes instead of five in-place stages:
//       P1  lane l < 60 holds window samples 4l+c+240k (four float4 straight from HBM): window, scale and the first
//           radix-4 stage (kiss_fft.cpp:112-131) happen in registers; there is no digit-reversal scatter
//       P2  lane l < 60: radix-4 stages m=4 and m=16 (139-166) on the 16 elements 64*blk + 16a + 4a' + j', in place
//       P3  lane u < 64: radix-3 (196-227) and radix-5 (259-304) on the 15 elements u + 64b + 192c; the results stay
//           in registers (bins >= 400 are never used, denoise.cpp:89-182: their butterflies' outputs are not formed)
//     with the LDS layout phi(i) = i + 4*(i >> 6) (float2 units) that makes the P2 and P3 accesses bank-conflict-free;
//     tools/fft960_model.py proves bit for bit that the three passes equal the five stages and prints the conflict
//     factors;
//   * the band reductions read their operands band-major (PnTables.band_*): the per-bin terms frac*tmp / (1-frac)*tmp
//     are formed lane-parallel and stored where the band that sums them reads them as aligned float4 — one
//     ds_read_b128 per four chain steps instead of two scalar reads per step.
//
// Numerics contract: as pn_dsp_fe.hip — every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (-ffp-contract=off); results are bit-identical to the single-launch kernel and
// to the CPU reference (the band sums add explicit +0 padding terms: exact, a running sum that starts at +0 is never -0).
#define PN_FE_G 1
#include "pn_dsp_fe_helpers.inc"

#define FS_WPB 4                        // waves (= concurrent streams) per block
#define FS_THREADS (LANES * FS_WPB)
#include "pn_fft960.h"
#include "pn_launch.h"
#include <stdlib.h>

#ifndef PN_FS_WAVES_IN
#define PN_FS_WAVES_IN 4                // waves per SIMD the register budget of spec_in is cut for (4: 128 registers)
#endif
#ifndef PN_FS_PREFETCH
#define PN_FS_PREFETCH 1                 // spec_out: the next stream's period is requested one stream ahead
#endif
#ifndef PN_FS_WAVES_OUT
#define PN_FS_WAVES_OUT 3               // spec_out (28 comb-tap loads + the X spectrum on top): 168 registers, three 4-wave blocks per CU
#endif
struct FsShared {
  float2 f[FS_WPB][FS_NF];              // 8160 B per wave; the band-major operand arrays alias it after the FFT
  float win[PN_FRAME];                  // 1920 B: half window (denoise.cpp:191-192), read four weights at a time
  float frac[PN_SPEC_BINS];             // 1600 B: PnTables.bin_frac
  unsigned pos[PN_SPEC_BINS];           // 1600 B: band_pos_a | band_pos_b << 16
};

// ---- band reductions on band-major operands (denoise.cpp:89-160) ------------------------------------------------------------
// bins of lane u: k(j) = u + 64 j, j = 0..5, and 384 + u for u < 16 (j = 6)
#define FS_NBIN 7
struct FsBands {
  const float *frac;                   // LDS copies of the per-bin tables (block-shared)
  const unsigned *pos;
  int start, nq;                       // band lane (lane < 34): first float and number of quads of its operand run
};
__device__ __forceinline__ void fs_bands_init(FsBands &B, const FsShared &SH, const PnTables *__restrict__ T, int l) {
  B.frac = SH.frac; B.pos = SH.pos;
  B.start = l < PN_NB ? T->band_start[l] : 0;
  B.nq = l < PN_NB ? T->band_nq[l] : 0;
}
// zero the layout (its pad slots must read +0), then scatter the per-bin terms of tmp[j] (bin k(j)) into it
__device__ __forceinline__ void fs_bands_fill(float *C, const FsBands &B, const float *tmp, int l) {
#pragma unroll
  for (int it = 0; it < (PN_BAND_LAYOUT_FLOATS / 4 + 63) / 64; it++) {
    const int q = l + 64 * it;
    if (q < PN_BAND_LAYOUT_FLOATS / 4) *reinterpret_cast<float4 *>(C + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float fr[FS_NBIN]; unsigned ps[FS_NBIN];
#pragma unroll
  for (int j = 0; j < FS_NBIN; j++) {
    const int k = (j < 6) ? l + 64 * j : (l < 16 ? 384 + l : 0);
    fr[j] = B.frac[k]; ps[j] = B.pos[k];
  }
#pragma unroll
  for (int j = 0; j < FS_NBIN; j++) {
    if (j == 6 && l >= 16) continue;
    const float fa = fr[j] * tmp[j], fb = (1 - fr[j]) * tmp[j];
    C[ps[j] & 0xffff] = fa;
    C[ps[j] >> 16] = fb;
  }
}
// the chains: lane b < 34 sums its run in order; NARR arrays at C, C + PN_BAND_LAYOUT_FLOATS, ...
template <int NARR>
__device__ __forceinline__ void fs_bands_sum(const float *C, const FsBands &B, int l, float *sum) {
#pragma unroll
  for (int h = 0; h < NARR; h++) sum[h] = 0;
  const float *p = C + B.start;
  for (int q = 0; q < 25; q++) {                    // 25 = the longest run (band 32: 45 + 51 bins)
    if (q < B.nq) {
#pragma unroll
      for (int h = 0; h < NARR; h++) {
        const float4 v = *reinterpret_cast<const float4 *>(p + h * PN_BAND_LAYOUT_FLOATS + 4 * q);
        sum[h] += v.x; sum[h] += v.y; sum[h] += v.z; sum[h] += v.w;
      }
    }
  }
  if (l == 0 || l == PN_NB - 1)
#pragma unroll
    for (int h = 0; h < NARR; h++) sum[h] *= 2;
}

// ---- spectral-in: history write + look-ahead FFT + band energies ------------------------------------------------------
template <typename TIn>
__global__ __launch_bounds__(FS_THREADS, PN_FS_WAVES_IN) void pn_fe_spec_in_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t, int slot_w,
    const TIn *__restrict__ in, long long in_stride, float i16_scale,
    float *__restrict__ hist, float2 *__restrict__ yring, float *__restrict__ eyring) {
  __shared__ FsShared SH;
  const int tid = threadIdx.x, l = tid & (LANES - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the stream pointers below are scalar
  float2 *F = SH.f[wave];
  float *C = reinterpret_cast<float *>(F);
  for (int i = tid; i < PN_FRAME; i += FS_THREADS) SH.win[i] = T->half_window[i];
  for (int i = tid; i < PN_SPEC_BINS; i += FS_THREADS) {
    SH.frac[i] = T->bin_frac[i];
    SH.pos[i] = (unsigned)T->band_pos_a[i] | ((unsigned)T->band_pos_b[i] << 16);
  }
  __syncthreads();
  FsLane Z; FsBands B;
  fs_lane_init(Z, T, l);
  fs_bands_init(B, SH, T, l);
  const int new_slot = frame_t % PN_HIST_FRAMES;
  const int prev_slot = (frame_t + PN_HIST_FRAMES - 1) % PN_HIST_FRAMES;
  const int lc = l < 60 ? l : 59;
  for (int s = blockIdx.x * FS_WPB + wave; s < n_streams; s += gridDim.x * FS_WPB) {
    float *h = hist + (size_t)s * PN_HIST_STRIDE;
    // frame = the previous input frame (ring slot t-1) | the new one; lane l < 60 takes samples 4l + 240k
    float4 x[4];
#pragma unroll
    for (int k = 0; k < 2; k++) x[k] = *reinterpret_cast<const float4 *>(h + prev_slot * PN_FRAME + 4 * lc + 240 * k);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (sizeof(TIn) == 2) {
        const short4 q = *reinterpret_cast<const short4 *>(in + (size_t)s * in_stride + 4 * lc + 240 * k);
        // a power-of-two scale: the product is exact, == the reference's division (main.cpp:34)
        x[2 + k] = make_float4(((float)q.x) * i16_scale, ((float)q.y) * i16_scale, ((float)q.z) * i16_scale, ((float)q.w) * i16_scale);
      } else {
        x[2 + k] = *reinterpret_cast<const float4 *>(in + (size_t)s * in_stride + 4 * lc + 240 * k);
      }
    }
    if (l < 60) {                                  // the shift+append of denoise.cpp:388-389 is one ring-slot write
#pragma unroll
      for (int k = 0; k < 2; k++) {
        *reinterpret_cast<float4 *>(h + new_slot * PN_FRAME + 4 * l + 240 * k) = x[2 + k];
        if (new_slot == 0 && k == 0 && l < 2) *reinterpret_cast<float4 *>(h + PN_HIST + 4 * l) = x[2];   // mirror of the ring's first 8 samples
      }
    }
    fs_fft_p1(F, SH.win, Z, x, l);
    float2 w[3][5];
    fs_fft_p23<false>(F, T, l, w);
    const float2 o2 = w[0][2];
    float2 *yw = yring + ((size_t)slot_w * n_streams + s) * PN_SPEC_BINS;
    float tmp[FS_NBIN];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const float2 a = w[j % 3][j / 3];            // bin u + 64 j = u + 64 (j % 3) + 192 (j / 3)
      yw[l + 64 * j] = a;
      float t = a.x * a.x; t += a.y * a.y; tmp[j] = t;
    }
    if (l < 16) yw[384 + l] = o2;
    { float t = o2.x * o2.x; t += o2.y * o2.y; tmp[6] = t; }
    PN_WAVE_SYNC();
    fs_bands_fill(C, B, tmp, l);
    PN_WAVE_SYNC();
    float e;
    fs_bands_sum<1>(C, B, l, &e);
    if (l < PN_NB) eyring[((size_t)slot_w * n_streams + s) * 36 + l] = e;
    PN_WAVE_SYNC();
  }
}

// Everything of a stream after its comb-filtered frame x[]: window + transform -> P, band energy Ep and correlation X.P,
// features, silence flag (denoise.cpp:423-434, create_features 487-496).  X(t) is requested AFTER the transform: held across it
// its 14 registers cost 11 spills at three waves per SIMD (measured: no gain).
__device__ __forceinline__ void fs_spec_out_tail(const float4 *x, const float2 *__restrict__ Xr, float Ex, float Ey, int s, int l, float2 *F, float *C,
                                                 const FsShared &SH, const FsLane &Z, const FsBands &B, const PnTables *__restrict__ T,
                                                 float2 *__restrict__ Pspec, float *__restrict__ feat, int *__restrict__ silence, float *__restrict__ aux) {
  fs_fft_p1(F, SH.win, Z, x, l);
  float2 w[3][5];
  fs_fft_p23<false>(F, T, l, w);
  const float2 o2 = w[0][2];
  float2 xv[FS_NBIN];
#pragma unroll
  for (int j = 0; j < FS_NBIN; j++) xv[j] = Xr[(j < 6) ? l + 64 * j : (l < 16 ? 384 + l : 0)];
  float tp[FS_NBIN], tx[FS_NBIN];
#pragma unroll
  for (int j = 0; j < FS_NBIN; j++) {
    const float2 P = j < 6 ? w[j % 3][j / 3] : o2;
    if (j < 6 || l < 16) Pspec[(size_t)s * PN_SPEC_BINS + (j < 6 ? l + 64 * j : 384 + l)] = P;
    float t = P.x * P.x; t += P.y * P.y; tp[j] = t;        // compute_band_energy's per-bin term (denoise.cpp:100-101)
    float u = xv[j].x * P.x; u += xv[j].y * P.y; tx[j] = u;  // compute_band_corr's (136-137)
  }
  PN_WAVE_SYNC();
  fs_bands_fill(C, B, tp, l);
  fs_bands_fill(C + PN_BAND_LAYOUT_FLOATS, B, tx, l);
  PN_WAVE_SYNC();
  float sums[2];
  fs_bands_sum<2>(C, B, l, sums);
  const float Ep = sums[0];
  float Exp = sums[1];
  float *f = feat + (size_t)s * PN_FEAT_STRIDE;
  if (l < PN_NB) {
    // double island, denoise.cpp:427
    Exp = (float)fmin(1.0, fmax(0.0, (double)Exp / sqrt(1e-15 + (double)(Ex * Ep))));
    f[l] = Ey * 30;              // create_features (487-496)
    f[PN_NB + l] = Exp * 30;
    if (aux) { aux[(size_t)s * PN_AUX_STRIDE + l] = Ep; aux[(size_t)s * PN_AUX_STRIDE + PN_NB + l] = Exp; }
  }
  // silence = sum(Ex) < 0.1 (429-433): sequential sum over the 34 bands, in band order
  {
    float E = 0;
#pragma unroll
    for (int i = 0; i < PN_NB; i++) E += __shfl(Ex, i);
    if (l == 0) silence[s] = ((double)E < 0.1) ? 1 : 0;
  }
  PN_WAVE_SYNC();
}

// ---- spectral-out: comb filter at the pitch period + window + FFT -> P, Ep, Exp, features ---------------------------------
__global__ __launch_bounds__(FS_THREADS, PN_FS_WAVES_OUT) void pn_fe_spec_out_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t, int slot_w, int slot_r,
    const float *__restrict__ hist, const float2 *__restrict__ yring, const float *__restrict__ eyring,
    const int *__restrict__ last_period,      // written by the pitch kernel of this frame
    float2 *__restrict__ Pspec, float *__restrict__ feat, int *__restrict__ silence, float *__restrict__ aux) {
  __shared__ FsShared SH;
  const int tid = threadIdx.x, l = tid & (LANES - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the stream pointers below are scalar
  float2 *F = SH.f[wave];
  float *C = reinterpret_cast<float *>(F);
  for (int i = tid; i < PN_FRAME; i += FS_THREADS) SH.win[i] = T->half_window[i];
  for (int i = tid; i < PN_SPEC_BINS; i += FS_THREADS) {
    SH.frac[i] = T->bin_frac[i];
    SH.pos[i] = (unsigned)T->band_pos_a[i] | ((unsigned)T->band_pos_b[i] << 16);
  }
  __syncthreads();
  FsLane Z; FsBands B;
  fs_lane_init(Z, T, l);
  fs_bands_init(B, SH, T, l);
  float cw[7];
#pragma unroll
  for (int k = 0; k < 7; k++) cw[k] = T->comb_hann[k];
  const int base_slot0 = (frame_t + 1) % PN_HIST_FRAMES;   // slot of logical frame 0 (oldest)
  const int lc = l < 60 ? l : 59;
#if PN_FS_PREFETCH & 1
  // the pitch period of a wave's NEXT stream is requested one stream ahead: otherwise every stream starts with two dependent
  // memory round trips (the period, then the 28 tap loads whose addresses it decides) with nothing else in flight
  int period_next = 0;
  { const int s0 = blockIdx.x * FS_WPB + wave; if (s0 < n_streams) period_next = last_period[s0]; }
#endif
  for (int s = blockIdx.x * FS_WPB + wave; s < n_streams; s += gridDim.x * FS_WPB) {
    const float *h = hist + (size_t)s * PN_HIST_STRIDE;
#if PN_FS_PREFETCH & 1
    const int pitch_index = period_next;
    { const int sn = s + gridDim.x * FS_WPB; period_next = last_period[sn < n_streams ? sn : s]; }
#else
    const int pitch_index = last_period[s];
#endif
    int base_slot = base_slot0;
    asm volatile("" : "+v"(base_slot));              // keeps the 28 ring offsets below from being hoisted out of the stream loop
    const float2 *Xr = yring + ((size_t)slot_r * n_streams + s) * PN_SPEC_BINS;   // X(t)  = Y(t-5)
    const float Ex = l < PN_NB ? eyring[((size_t)slot_r * n_streams + s) * 36 + l] : 0.f;   // Ex(t) = Ey(t-5)
    const float Ey = l < PN_NB ? eyring[((size_t)slot_w * n_streams + s) * 36 + l] : 0.f;   // Ey of this frame
    // comb filter (denoise.cpp:416-422): lane l < 60 filters samples 4l + 240k .. +3, k = 0..3; one unaligned dwordx4
    // load per tap (the ring carries an 8-sample mirror).  Two k at a time: 14 loads in flight.
    // Measured in round 5 (profiles/r05_fe_steady_state.log, r05_front_end_variants.log, r05_fetch_size_calibration.log): in steady state the windows
    // two taps share ARE served once — 22.5 KB of traffic per stream against 21.8 KB of algorithmic bytes (mean period 446);
    // a tap-major order with 8 .. 28 loads in flight moved 23.4 KB in the same time, a cross-stream pipeline at two waves
    // per SIMD (all 28 windows of the next stream requested under the current stream's transform) was 35 % slower.
    float4 x[4];
    int bs = base_slot;
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += 2) {
      // the second half's ring offsets are made to depend on the first half's result: its 14 loads (56 registers) are
      // issued after the first half's arithmetic instead of next to the first 14
      if (k0) asm volatile("" : "+v"(bs) : "v"(x[1].w));
      fe_f4u cv[2][7];
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int k = -PN_COMB_M; k <= PN_COMB_M; k++)
          cv[q][k + PN_COMB_M] = *reinterpret_cast<const fe_f4u *>(h + fe_ring(2400 - pitch_index * k + 4 * lc + 240 * (k0 + q), bs));
#pragma unroll
      for (int q = 0; q < 2; q++) {
        float p[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          float acc = 0;
#pragma unroll
          for (int k = 0; k < 7; k++) acc += cv[q][k][c] * cw[k];
          p[c] = acc;
        }
        x[k0 + q] = make_float4(p[0], p[1], p[2], p[3]);
      }
    }
    fs_spec_out_tail(x, Xr, Ex, Ey, s, l, F, C, SH, Z, B, T, Pspec, feat, silence, aux);
  }
}


// ---- launchers --------------------------------------------------------------------------------------------------------
static int fs_grid(int n_streams, int blocks_per_cu, int grid_cap) {
  const int need = (n_streams + FS_WPB - 1) / FS_WPB;
  const int cap = grid_cap > 0 ? grid_cap : 256 * blocks_per_cu;   // resident 4-wave blocks on 256 CUs
  return need < cap ? need : cap;
}
void pn_launch_fe_spec_in(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in, int in_is_i16,
                          long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring, int grid_cap) {
  const int frame_t = (int)(frame % PN_HIST_FRAMES), slot_w = (int)(frame % 6);
  if (in_is_i16)
    hipLaunchKernelGGL(pn_fe_spec_in_kernel<int16_t>, dim3(fs_grid(n_streams, PN_FS_WAVES_IN, grid_cap)), dim3(FS_THREADS), 0, st, T, n_streams, frame_t,
                       slot_w, (const int16_t *)in, in_stride, i16_scale, hist, yring, eyring);
  else
    hipLaunchKernelGGL(pn_fe_spec_in_kernel<float>, dim3(fs_grid(n_streams, PN_FS_WAVES_IN, grid_cap)), dim3(FS_THREADS), 0, st, T, n_streams, frame_t,
                       slot_w, (const float *)in, in_stride, i16_scale, hist, yring, eyring);
}
void pn_launch_fe_spec_out(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const float *hist,
                           const float2 *yring, const float *eyring, const int *last_period, float2 *Ps, float *feat,
                           int *silence, float *aux, int grid_cap) {
  const int frame_t = (int)(frame % PN_HIST_FRAMES), slot_w = (int)(frame % 6), slot_r = (int)((frame + 1) % 6);
  hipLaunchKernelGGL(pn_fe_spec_out_kernel, dim3(fs_grid(n_streams, PN_FS_WAVES_OUT, grid_cap)), dim3(FS_THREADS), 0, st, T, n_streams, frame_t, slot_w,
                       slot_r, hist, yring, eyring, last_period, Ps, feat, silence, aux);
}

// The three phase kernels in sequence = pn_launch_frontend (same arguments, same results bit for bit).
void pn_launch_frontend_split(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in, int in_is_i16,
                              long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring, float2 *Ps,
                              float *feat, int *silence, int *last_period, float *last_gain, float *aux, int grid_cap) {
  pn_launch_fe_spec_in(st, T, n_streams, frame, in, in_is_i16, in_stride, i16_scale, hist, yring, eyring, grid_cap);
  pn_launch_fe_pitch(st, n_streams, frame, hist, feat, last_period, last_gain, aux, grid_cap);
  pn_launch_fe_spec_out(st, T, n_streams, frame, hist, yring, eyring, last_period, Ps, feat, silence, aux, grid_cap);
}
