// Phase-split front end, spectral kernels (gfx950): ONE stream per wavefront, 64 lanes.
//
//   pn_fe_spec_in_kernel   history ring write, window + 960-pt FFT of the newest 960 samples -> look-ahead spectrum Y
//                          and band energies Ey into their 6-slot rings (== the analysis side of frame_analysis for
//                          frame t+5 and compute_lookahead_band_energy, reference denoise.cpp:333-346, 498-506)
//   pn_fe_spec_out_kernel  7-tap comb filter at the pitch period the pitch kernel found, window + FFT -> P, Ep, the
//                          X.P band correlation, the 68 band features, silence flag
//                          (denoise.cpp:416-434, create_features 487-496)
// The pitch analysis between the two lives in pn_dsp_fe_split_p.hip (four streams per wave).  Why three kernels: the
// single-launch front end (pn_dsp_fe.hip) needs 8.3 KB of LDS per stream for its FFT buffer and 475 registers per lane
// for the batched loads that hide latency with ONE wave per SIMD; its data-parallel phases (FFT, windows, comb filter)
// have no use for four streams per wave.  Here they run with one stream per wave and FOUR waves per SIMD (<= 128
// registers, 36.6 KB of LDS per 4-wave block, four blocks per CU): measured on MI355X, every instruction a lone wave
// issues costs 4.5 cycles whatever its kind, two waves per SIMD double the rate and four reach the VALU peak
// (profiles/r03a_valu_issue_probe.log).
//
// Numerics contract: as pn_dsp_fe.hip — every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (-ffp-contract=off); results are bit-identical to the single-launch kernel and
// to the CPU reference.  The FFT butterflies are the reference's (kiss_fft.cpp:112-304); only which lane evaluates
// which butterfly changes, and the twiddles a lane needs are per-lane constants held in registers (with 64 lanes the
// twiddle index of a butterfly depends on the lane only; the twelve of the last stage are fetched from the L1-resident
// global table while the radix-3 stage runs), so no twiddle table is staged in LDS.
#define PN_FE_G 1
#include "pn_dsp_fe_helpers.inc"

#define FS_WPB 4                        // waves (= concurrent streams) per block
#define FS_THREADS (LANES * FS_WPB)

struct alignas(16) FsTablesLds {
  float win[PN_FRAME];           // 1920 B
  float frac[PN_SPEC_BINS];      // 1600 B
  int16_t bitrev[PN_NFFT];       // 1920 B
  int16_t border[PN_NB + 2];
  float comb_w[8];
};
struct alignas(16) FsWaveLds {
  float2 fft[PN_NFFT];           // 7680 B
};
struct FsShared {
  FsTablesLds t;
  FsWaveLds w[FS_WPB];
};

__device__ __forceinline__ void fs_stage_tables(FsTablesLds &S, const PnTables *__restrict__ T) {
  const int tid = threadIdx.x;
  for (int i = tid; i < PN_NFFT; i += FS_THREADS) S.bitrev[i] = T->bitrev[i];
  for (int i = tid; i < PN_FRAME; i += FS_THREADS) S.win[i] = T->half_window[i];
  for (int i = tid; i < PN_SPEC_BINS; i += FS_THREADS) S.frac[i] = T->bin_frac[i];
  if (tid < PN_NB + 2) S.border[tid] = T->border[tid];
  if (tid < 8) S.comb_w[tid] = T->comb_hann[tid];
  __syncthreads();
}

// The twiddles one lane ever needs (kiss_fft.cpp:139-304 index them by butterfly position; with 64 lanes and
// butterfly b = lane + 64*iteration the position inside a stage depends on the lane only).
struct FsTw {
  float2 a4[3];        // radix-4, m=4:  tw[j*60*k], j = lane % 4, k = 1..3
  float2 a16[3];       // radix-4, m=16: tw[j*15*k], j = lane % 16
  float2 r3[2];        // radix-3, m=64: tw[5*lane], tw[10*lane]
  float2 ya, yb;       // tw[192], tw[384]
  float epi3;          // tw[320].y
};
__device__ __forceinline__ float2 fs_tw(const PnTables *__restrict__ T, int i) { return make_float2(T->tw[2 * i], T->tw[2 * i + 1]); }
__device__ __forceinline__ void fs_load_tw(FsTw &W, const PnTables *__restrict__ T, int lane) {
#pragma unroll
  for (int k = 1; k <= 3; k++) { W.a4[k - 1] = fs_tw(T, (lane & 3) * 60 * k); W.a16[k - 1] = fs_tw(T, (lane & 15) * 15 * k); }
  W.r3[0] = fs_tw(T, 5 * lane); W.r3[1] = fs_tw(T, 10 * lane);
  W.ya = fs_tw(T, 192); W.yb = fs_tw(T, 384);
  W.epi3 = T->tw[2 * 320 + 1];
}

// 960-point FFT in LDS by the 64 lanes of one wave (opus_fft_impl, kiss_fft.cpp:518-564, factors 5,3,4,4,4); input
// already scaled by 1/960 and digit-reverse scattered (opus_fft_c 578-585).  Same butterfly arithmetic as fe_fft960.
__device__ __forceinline__ void fs_fft960(float2 *F, const FsTw &W, const PnTables *__restrict__ T, int l) {
  PN_WAVE_SYNC();
#pragma unroll 1                             // rolled: four waves per SIMD hide the latency, and 128 registers hold without spills
  for (int it = 0; it < 4; it++) {           // radix-4, m=1 (kiss_fft.cpp:112-131): 240 butterflies
    const int b = l + 64 * it;
    if (b < 240) {
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
  }
  PN_WAVE_SYNC();
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {     // radix-4, m=4 (fstride 60) then m=16 (fstride 15) (139-166)
    const int m = pass ? 16 : 4, mm = pass ? 64 : 16;
    const float2 t1 = pass ? W.a16[0] : W.a4[0], t2 = pass ? W.a16[1] : W.a4[1], t3 = pass ? W.a16[2] : W.a4[2];
#pragma unroll 1
    for (int it = 0; it < 4; it++) {
      const int b = l + 64 * it;
      if (b < 240) {
        const int i = b / m, j = b % m;
        float2 *f = F + i * mm + j;
        float2 f0 = f[0], fm = f[m], f2m = f[2 * m], f3m = f[3 * m];
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
    }
    PN_WAVE_SYNC();
  }
  float2 r5[3][4];                           // radix-5 twiddles tw[k*u], u = lane + 64*it: in flight during the radix-3 stage
#pragma unroll
  for (int it = 0; it < 3; it++)
#pragma unroll
    for (int k = 1; k <= 4; k++) r5[it][k - 1] = fs_tw(T, k * (l + 64 * it));
#pragma unroll
  for (int i = 0; i < 5; i++) {              // radix-3, m=64, fstride 5 (196-227): 320 butterflies, j = lane
    float2 *f = F + i * 192 + l;
    float2 f0 = f[0], fm = f[64], f2m = f[128], s0, s1, s2, s3;
    CMUL(s1, fm, W.r3[0]); CMUL(s2, f2m, W.r3[1]);
    s3.x = s1.x + s2.x; s3.y = s1.y + s2.y;
    s0.x = s1.x - s2.x; s0.y = s1.y - s2.y;
    fm.x = f0.x - s3.x * .5f; fm.y = f0.y - s3.y * .5f;
    s0.x *= W.epi3; s0.y *= W.epi3;
    f0.x += s3.x; f0.y += s3.y;
    f2m.x = fm.x + s0.y; f2m.y = fm.y - s0.x;
    fm.x = fm.x - s0.y; fm.y = fm.y + s0.x;
    f[0] = f0; f[64] = fm; f[128] = f2m;
  }
  PN_WAVE_SYNC();
  {
    const float2 ya = W.ya, yb = W.yb;       // radix-5, m=192, fstride 1 (259-304): 192 butterflies
#pragma unroll
    for (int it = 0; it < 3; it++) {
      float2 *f = F + l + 64 * it;
      float2 f0 = f[0], f1 = f[192], f2 = f[384], f3 = f[576], f4 = f[768];
      float2 s0 = f0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12;
      CMUL(s1, f1, r5[it][0]); CMUL(s2, f2, r5[it][1]); CMUL(s3, f3, r5[it][2]); CMUL(s4, f4, r5[it][3]);
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
  }
  PN_WAVE_SYNC();
}

// ---- spectral-in: history write + look-ahead FFT + band energies ------------------------------------------------------
template <typename TIn>
__global__ __launch_bounds__(FS_THREADS, 4) void pn_fe_spec_in_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t, int slot_w,
    const TIn *__restrict__ in, long long in_stride, float i16_scale,
    float *__restrict__ hist, float2 *__restrict__ yring, float *__restrict__ eyring) {
  __shared__ FsShared SH;
  const int tid = threadIdx.x, l = tid & (LANES - 1), wave = tid >> 6;
  fs_stage_tables(SH.t, T);
  const FsTablesLds &S = SH.t;
  float2 *F = SH.w[wave].fft;
  FsTw W;
  fs_load_tw(W, T, l);
  const int new_slot = frame_t % PN_HIST_FRAMES;
  const int prev_slot = (frame_t + PN_HIST_FRAMES - 1) % PN_HIST_FRAMES;
  const float scale = 1.f / PN_NFFT;
  for (int s = blockIdx.x * FS_WPB + wave; s < n_streams; s += gridDim.x * FS_WPB) {
    float *h = hist + (size_t)s * PN_HIST_STRIDE;
    // window = the previous frame (ring slot t-1) | the new frame: 2 x 120 float4, lane l takes float4 l and l + 64
    float4 ov[2], nv[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int i4 = l + 64 * it, i4c = i4 < PN_FRAME / 4 ? i4 : 0;
      ov[it] = *reinterpret_cast<const float4 *>(h + prev_slot * PN_FRAME + 4 * i4c);
      if (sizeof(TIn) == 2) {
        const short4 q = *reinterpret_cast<const short4 *>(in + (size_t)s * in_stride + 4 * i4c);
        // a power-of-two scale: the product is exact, == the reference's division (main.cpp:34)
        nv[it] = make_float4(((float)q.x) * i16_scale, ((float)q.y) * i16_scale, ((float)q.z) * i16_scale, ((float)q.w) * i16_scale);
      } else {
        nv[it] = *reinterpret_cast<const float4 *>(in + (size_t)s * in_stride + 4 * i4c);
      }
    }
#pragma unroll
    for (int it = 0; it < 2; it++) {           // the shift+append of denoise.cpp:388-389 is one ring-slot write
      const int i4 = l + 64 * it;
      if (i4 < PN_FRAME / 4) {
        *reinterpret_cast<float4 *>(h + new_slot * PN_FRAME + 4 * i4) = nv[it];
        if (new_slot == 0 && i4 < 2) *reinterpret_cast<float4 *>(h + PN_HIST + 4 * i4) = nv[it];   // mirror of the ring's first 8 samples
      }
    }
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const int i4 = l + 64 * it;
        if (i4 >= PN_FRAME / 4) continue;
        const float4 v4 = half ? nv[it] : ov[it];
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int ii = half * PN_FRAME + 4 * i4 + c;
          const float w = S.win[ii < PN_FRAME ? ii : PN_WINDOW - 1 - ii];   // apply_window 282-289
          F[S.bitrev[ii]] = make_float2(scale * (vv[c] * w), scale * 0.f);
        }
      }
    fs_fft960(F, W, T, l);
    float2 *yw = yring + ((size_t)slot_w * n_streams + s) * PN_SPEC_BINS;
#pragma unroll
    for (int it = 0; it < 7; it++) { const int k = l + 64 * it; if (k < PN_SPEC_BINS) yw[k] = F[k]; }
    const float e = fe_band<false>(S, F, nullptr, l);
    if (l < PN_NB) eyring[((size_t)slot_w * n_streams + s) * 36 + l] = e;
    PN_WAVE_SYNC();
  }
}

// ---- spectral-out: comb filter at the pitch period + window + FFT -> P, Ep, Exp, features ---------------------------------
__global__ __launch_bounds__(FS_THREADS, 4) void pn_fe_spec_out_kernel(
    const PnTables *__restrict__ T, int n_streams, int frame_t, int slot_w, int slot_r,
    const float *__restrict__ hist, const float2 *__restrict__ yring, const float *__restrict__ eyring,
    const int *__restrict__ last_period,      // written by the pitch kernel of this frame
    float2 *__restrict__ Pspec, float *__restrict__ feat, int *__restrict__ silence, float *__restrict__ aux) {
  __shared__ FsShared SH;
  const int tid = threadIdx.x, l = tid & (LANES - 1), wave = tid >> 6;
  fs_stage_tables(SH.t, T);
  const FsTablesLds &S = SH.t;
  float2 *F = SH.w[wave].fft;
  float *prod = reinterpret_cast<float *>(F) + 960;     // per-bin X.P products: bins live in F[0,400) = floats [0,800)
  FsTw W;
  fs_load_tw(W, T, l);
  const int base_slot = (frame_t + 1) % PN_HIST_FRAMES;   // slot of logical frame 0 (oldest)
  const float scale = 1.f / PN_NFFT;
  for (int s = blockIdx.x * FS_WPB + wave; s < n_streams; s += gridDim.x * FS_WPB) {
    const float *h = hist + (size_t)s * PN_HIST_STRIDE;
    const int pitch_index = last_period[s];
    const float2 *Xr = yring + ((size_t)slot_r * n_streams + s) * PN_SPEC_BINS;   // X(t)  = Y(t-5)
    const float Ex = l < PN_NB ? eyring[((size_t)slot_r * n_streams + s) * 36 + l] : 0.f;   // Ex(t) = Ey(t-5)
    const float Ey = l < PN_NB ? eyring[((size_t)slot_w * n_streams + s) * 36 + l] : 0.f;   // Ey of this frame
    // comb filter (denoise.cpp:416-422): lane l filters 4 consecutive samples per group, groups l + 64*it (240 groups);
    // one unaligned dwordx4 load per tap (the ring carries an 8-sample mirror)
#pragma unroll 1
    for (int q0 = 0; q0 < 4; q0 += 2) {
      fe_f4u cv[2][7];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int gi = l + 64 * (q0 + q), gc = gi < PN_WINDOW / 4 ? gi : 0;
#pragma unroll
        for (int k = -PN_COMB_M; k <= PN_COMB_M; k++)
          cv[q][k + PN_COMB_M] = *reinterpret_cast<const fe_f4u *>(h + fe_ring(2400 - pitch_index * k + 4 * gc, base_slot));
      }
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int gi = l + 64 * (q0 + q);
        if (gi >= PN_WINDOW / 4) continue;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int i = 4 * gi + c;
          float p = 0;
#pragma unroll
          for (int k = 0; k < 7; k++) p += cv[q][k][c] * S.comb_w[k];
          const float v = p * S.win[i < PN_FRAME ? i : PN_WINDOW - 1 - i];
          F[S.bitrev[i]] = make_float2(scale * v, scale * 0.f);
        }
      }
    }
    fs_fft960(F, W, T, l);
    {
      float2 xv[7];
#pragma unroll
      for (int it = 0; it < 7; it++) { const int k = l + 64 * it; xv[it] = Xr[k < PN_SPEC_BINS ? k : 0]; }
#pragma unroll
      for (int it = 0; it < 7; it++) {
        const int k = l + 64 * it;
        if (k >= PN_SPEC_BINS) continue;
        const float2 P = F[k];
        Pspec[(size_t)s * PN_SPEC_BINS + k] = P;
        float tmp = xv[it].x * P.x;            // compute_band_corr's per-bin term (denoise.cpp:136-137)
        tmp += xv[it].y * P.y;
        prod[k] = tmp;
      }
    }
    PN_WAVE_SYNC();
    const float Ep = fe_band<false>(S, F, nullptr, l);
    float Exp = fe_band<true>(S, nullptr, prod, l);
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
}

// ---- launchers --------------------------------------------------------------------------------------------------------
static int fs_grid(int n_streams) {
  const int need = (n_streams + FS_WPB - 1) / FS_WPB;
  const int cap = 256 * 4;                               // four LDS-resident blocks on each of 256 CUs
  return need < cap ? need : cap;
}
void pn_launch_fe_spec_in(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in, int in_is_i16,
                          long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring) {
  const int frame_t = (int)(frame % PN_HIST_FRAMES), slot_w = (int)(frame % 6);
  if (in_is_i16)
    hipLaunchKernelGGL(pn_fe_spec_in_kernel<int16_t>, dim3(fs_grid(n_streams)), dim3(FS_THREADS), 0, st, T, n_streams, frame_t,
                       slot_w, (const int16_t *)in, in_stride, i16_scale, hist, yring, eyring);
  else
    hipLaunchKernelGGL(pn_fe_spec_in_kernel<float>, dim3(fs_grid(n_streams)), dim3(FS_THREADS), 0, st, T, n_streams, frame_t,
                       slot_w, (const float *)in, in_stride, i16_scale, hist, yring, eyring);
}
void pn_launch_fe_spec_out(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const float *hist,
                           const float2 *yring, const float *eyring, const int *last_period, float2 *Ps, float *feat,
                           int *silence, float *aux) {
  const int frame_t = (int)(frame % PN_HIST_FRAMES), slot_w = (int)(frame % 6), slot_r = (int)((frame + 1) % 6);
  hipLaunchKernelGGL(pn_fe_spec_out_kernel, dim3(fs_grid(n_streams)), dim3(FS_THREADS), 0, st, T, n_streams, frame_t, slot_w,
                     slot_r, hist, yring, eyring, last_period, Ps, feat, silence, aux);
}
