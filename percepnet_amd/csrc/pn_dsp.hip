// Back end of the PercepNet frame engine for gfx950 — one wavefront (64 lanes) per stream:
// pitch-filter mix, band-gain interpolation, inverse transform, window, overlap-add, PCM conversion
//   == pitch_filter + gain apply + frame_synthesis (reference denoise.cpp:436-485, 539-545) + the
//      CLI's float->short (main.cpp:36).
// (The front end — features, pitch, FFTs of the analysis side — is pn_dsp_fe.hip.)
//
// Numerics contract: every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (-ffp-contract=off), so given identical g/r the synthesis
// is bit-identical to the CPU reference.  512-thread blocks = 8 wavefronts = 8 concurrent streams;
// the block stages the shared tables into LDS once; each wave owns a private FFT buffer and loops
// over streams; no block-level barrier after the staging (PN_WAVE_SYNC is a compiler fence).
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

// float -> int16 as the reference CLI's x86-64 build does it (main.cpp:36): truncate toward zero
// to int32 (cvttss2si; NaN / out of range -> 0x80000000), keep the low 16 bits.
__device__ __forceinline__ int16_t pn_f2s(float v) {
  const int32_t t = (fabsf(v) < 2147483648.f) ? (int32_t)v : (int32_t)0x80000000;
  return (int16_t)(uint16_t)((uint32_t)t & 0xffffu);
}

// PF: optional envelope post-filter (reference post_filtering, denoise.cpp:216-250) on the
// gains before pitch_filter and the gain stage, where the reference's TEST synthesis has it (743).
template <typename TOut, bool PF>
__global__ __launch_bounds__(DSP_THREADS, PN_DSP_WAVES_PER_SIMD) void pn_backend_kernel(
    const PnTables *__restrict__ T, int n_streams,
    const float2 *__restrict__ Xspec, const float2 *__restrict__ Pspec,
    const float *__restrict__ gr,          // [n_streams][68]  g | r
    const float *__restrict__ ex,          // PF only: [n_streams][36] band energies of Xspec
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
    // global operands are read in batches ahead of their use (a load waited for inside the loop costs its full
    // latency every iteration); three batches of five keep the kernel at 4 waves per SIMD without spills
    float smv[(PN_FRAME + LANES - 1) / LANES];
#pragma unroll
    for (int it = 0; it < (PN_FRAME + LANES - 1) / LANES; it++) {
      const int i = lane + LANES * it;
      smv[it] = synth_mem[(size_t)s * PN_FRAME + (i < PN_FRAME ? i : 0)];
    }
    const bool sil = silence[s] != 0;
    if (lane < PN_NB) {
      const float g = gr[(size_t)s * 68 + lane], r = gr[(size_t)s * 68 + PN_NB + lane];
      W.e[0][lane] = g; W.e[1][lane] = r; W.e[2][lane] = 1 - r;
      if (PF) W.e[3][lane] = ex[(size_t)s * 36 + lane];
    }
    PN_WAVE_SYNC();
    if (PF) {
      // warped gain g*sinf(pi/2*g) (the pi/2*g product is double, 227), two sequential float sums in band
      // order (231-238), one global factor (241-244).  Every lane runs the 34-term chains redundantly on
      // LDS broadcasts; sinf is OCML's here and libm's in the reference (both within 1 ULP of sin).
      float gw = 0.f;
      float *gws = reinterpret_cast<float *>(W.fft);          // the FFT buffer is free until the scatter below
      if (lane < PN_NB) { const float g = W.e[0][lane]; gw = g * sinf((float)(M_PI / 2 * (double)g)); gws[lane] = gw; }
      PN_WAVE_SYNC();
      float E0 = 0.f, E1 = 0.f;
#pragma unroll 2
      for (int i = 0; i < PN_NB; i++) {
        const float e = W.e[3][i];
        E0 += W.e[0][i] * e;
        E1 += gws[i] * e;
      }
      const float E_div = E0 / (E1 + 1e-6f);
      const float G = sqrtf(((1 + 0.02f) * E_div) / (1 + 0.02f * (E_div * E_div)));
      PN_WAVE_SYNC();
      if (lane < PN_NB) W.e[0][lane] = G * gw;
      PN_WAVE_SYNC();
    }
    // pitch_filter (436-485, skipped when silent, 536-538), gain (539-544), then the Hermitian
    // extension + scale + digit-reverse scatter of inverse_transform (306-317).  Bins >= 400
    // are exactly 0 (interp_band_gain never writes them).
#pragma unroll 1
    for (int h0 = 0; h0 < 15; h0 += 5) {
      float2 xv[5], pv[5];
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (h0 + u < 15) {
          const int i = lane + LANES * (h0 + u);
          const int k = (i <= PN_FRAME) ? i : PN_WINDOW - i;
          const int kc = k < PN_SPEC_BINS ? k : PN_SPEC_BINS - 1;
          xv[u] = Xspec[(size_t)s * PN_SPEC_BINS + kc];
          pv[u] = Pspec[(size_t)s * PN_SPEC_BINS + kc];
        }
      }
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (h0 + u < 15) {
          const int i = lane + LANES * (h0 + u);
          const int k = (i <= PN_FRAME) ? i : PN_WINDOW - i;
          float2 x = make_float2(0.f, 0.f);
          if (k < PN_SPEC_BINS) {
            x = xv[u];
            const int b = S.band[k];
            const float fr = S.frac[k];
            if (!sil) {
              const float2 p = pv[u];
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
      }
    }
    pn_fft960_lds(W.fft, S.tw, lane);
    // reversed read-out x960 (318-323), window, overlap-add (352-359)
    float *sm = synth_mem + (size_t)s * PN_FRAME;
#pragma unroll
    for (int it = 0; it < (PN_FRAME + LANES - 1) / LANES; it++) {
      const int i = lane + LANES * it;
      if (i < PN_FRAME) {
        const float t_lo = (PN_WINDOW * W.fft[i == 0 ? 0 : PN_WINDOW - i].x) * S.win[i];
        const int i2 = PN_FRAME + i;                       // second half, window index 959 - i2
        const float t_hi = (PN_WINDOW * W.fft[PN_WINDOW - i2].x) * S.win[PN_WINDOW - 1 - i2];
        const float o = t_lo + smv[it];
        sm[i] = t_hi;
        if (sizeof(TOut) == 2) out[(size_t)s * PN_FRAME + i] = (TOut)pn_f2s(o * 32768);
        else out[(size_t)s * PN_FRAME + i] = (TOut)o;
      }
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

void pn_launch_backend(hipStream_t st, const PnTables *T, int n_streams, const float2 *Xs, const float2 *Ps,
                       const float *gr, const float *ex_postfilter, const int *silence, float *synth_mem, void *out,
                       int out_is_i16) {
  const int grid = pn_dsp_grid(n_streams, 0);
  const dim3 g(grid), b(DSP_THREADS);
  if (ex_postfilter) {
    if (out_is_i16)
      hipLaunchKernelGGL((pn_backend_kernel<int16_t, true>), g, b, 0, st, T, n_streams, Xs, Ps, gr, ex_postfilter, silence,
                         synth_mem, (int16_t *)out);
    else
      hipLaunchKernelGGL((pn_backend_kernel<float, true>), g, b, 0, st, T, n_streams, Xs, Ps, gr, ex_postfilter, silence,
                         synth_mem, (float *)out);
  } else {
    if (out_is_i16)
      hipLaunchKernelGGL((pn_backend_kernel<int16_t, false>), g, b, 0, st, T, n_streams, Xs, Ps, gr, ex_postfilter, silence,
                         synth_mem, (int16_t *)out);
    else
      hipLaunchKernelGGL((pn_backend_kernel<float, false>), g, b, 0, st, T, n_streams, Xs, Ps, gr, ex_postfilter, silence,
                         synth_mem, (float *)out);
  }
}
