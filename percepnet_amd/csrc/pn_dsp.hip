// Back end of the PercepNet frame engine for gfx950 — one wavefront (64 lanes) per stream:
// pitch-filter mix, band-gain interpolation, inverse transform, window, overlap-add, PCM conversion
//   == pitch_filter + gain apply + frame_synthesis (reference denoise.cpp:436-485, 539-545) + the
//      CLI's float->short (main.cpp:36).
// (The front end — features, pitch, FFTs of the analysis side — is pn_dsp_fe_split_*.hip / pn_dsp_fe.hip.)
//
// Numerics contract: every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (-ffp-contract=off), so given identical g/r the synthesis
// is bit-identical to the CPU reference.  512-thread blocks = 8 wavefronts = 8 concurrent streams, two blocks per CU.
//
// Round 3: the inverse transform (a forward 960-point FFT of the Hermitian-extended, conjugated spectrum read out in
// reverse, denoise.cpp:306-323) runs as the three register-fused passes of pn_fft960.h instead of a digit-reversal
// scatter + five in-place LDS stages (60 % of this kernel's LDS cycles were bank conflicts): lane l < 60 forms the 16
// inputs 4l + c + 240k of its four first-stage butterflies directly from the 400-bin spectra (pitch filter, gain,
// conjugate, 1/960), and the last pass leaves the 960 real outputs in registers, from where window, overlap-add and the
// PCM cast write straight to HBM.
#include "pn_fft960.h"
#include "pn_launch.h"

#define LANES 64
#ifndef PN_DSP_WPB
#define PN_DSP_WPB 8          // wavefronts (= concurrent streams) per block
#endif
#ifndef PN_DSP_WAVES_PER_SIMD
#define PN_DSP_WAVES_PER_SIMD 4
#endif
#define WPB PN_DSP_WPB
#define DSP_THREADS (LANES * WPB)

// same-wave global-memory store -> load ordering (drains vmcnt)
#define PN_WAVE_SYNC_GLOBAL() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

struct alignas(16) PnDspWaveLds {
  float2 fft[FS_NF];             // 8160 B  FFT work buffer (layout phi)
  float e[4][PN_NB + 2];         // g | r | 1 - r | Ex (post-filter)
};
struct PnDspShared {
  float win[PN_FRAME];           // 1920 B
  float frac[PN_SPEC_BINS];      // 1600 B
  uint8_t band[PN_SPEC_BINS];
  PnDspWaveLds w[WPB];
};

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
  const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < PN_FRAME; i += DSP_THREADS) SH.win[i] = T->half_window[i];
  for (int i = tid; i < PN_SPEC_BINS; i += DSP_THREADS) { SH.frac[i] = T->bin_frac[i]; SH.band[i] = T->bin_band[i]; }
  __syncthreads();
  PnDspWaveLds &W = SH.w[wave];
  FsLane Z;
  fs_lane_init(Z, T, lane);
  const float scale = 1.f / PN_NFFT;
  const int lc = lane < 60 ? lane : 59;
  for (int s = blockIdx.x * WPB + wave; s < n_streams; s += gridDim.x * WPB) {
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
      float *gws = reinterpret_cast<float *>(W.fft);          // the FFT buffer is free until P1 below
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
    // pitch_filter (436-485, skipped when silent, 536-538), gain (539-544), then the Hermitian extension + 1/960 of
    // inverse_transform (306-317).  Input i of the transform is bin k = i (i <= 480) or 960 - i (conjugated); bins >= 400
    // are exactly 0 (interp_band_gain never writes them).  Lane l < 60, butterfly c, input j: i = 4l + c + 240j.
    // (round 5: a two-register-set software pipeline over the four butterflies — the spectra of c + 1 requested before the
    // arithmetic of c — unrolls to 27 spilled registers at four waves per SIMD: 0.296 vs 0.250 ms, profiles/r05_front_end_variants.log)
    if (lane < 60) {
      const float2 *Xs = Xspec + (size_t)s * PN_SPEC_BINS, *Ps = Pspec + (size_t)s * PN_SPEC_BINS;
#pragma unroll 1
      for (int c = 0; c < 4; c++) {
        float2 xv[4], pv[4];
        int kk[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int i = 4 * lc + c + 240 * j;
          kk[j] = (i <= PN_FRAME) ? i : PN_WINDOW - i;
          const int kc = kk[j] < PN_SPEC_BINS ? kk[j] : PN_SPEC_BINS - 1;
          xv[j] = Xs[kc];
          pv[j] = Ps[kc];
        }
        float2 f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int i = 4 * lc + c + 240 * j;
          float2 x = make_float2(0.f, 0.f);
          if (kk[j] < PN_SPEC_BINS) {
            x = xv[j];
            const int b = SH.band[kk[j]];
            const float fr = SH.frac[kk[j]];
            if (!sil) {
              const float2 p = pv[j];
              const float rf1 = (1 - fr) * W.e[2][b] + fr * W.e[2][b + 1];
              x.x = rf1 * x.x; x.y = rf1 * x.y;
              const float rf2 = (1 - fr) * W.e[1][b] + fr * W.e[1][b + 1];
              x.x += rf2 * p.x; x.y += rf2 * p.y;
            }
            const float gf = (1 - fr) * W.e[0][b] + fr * W.e[0][b + 1];
            x.x *= gf; x.y *= gf;
          }
          if (i > PN_FRAME) x.y = -x.y;
          f[j] = make_float2(scale * x.x, scale * x.y);
        }
        fs_bfly4_m1(f);
        const int off = c == 0 ? Z.p1off[0] : (c == 1 ? Z.p1off[1] : (c == 2 ? Z.p1off[2] : Z.p1off[3]));
        float4 *dst = reinterpret_cast<float4 *>(W.fft + off);
        dst[0] = make_float4(f[0].x, f[0].y, f[1].x, f[1].y);
        dst[1] = make_float4(f[2].x, f[2].y, f[3].x, f[3].y);
      }
    }
    float2 w[3][5];
    fs_fft_p23<true>(W.fft, T, lane, w);
    // reversed read-out x960 (318-323), window, overlap-add (352-359): output p = u + 64b + 192c of the transform is
    //   p == 0 or p >= 481: time sample i = (960 - p) % 960 of the first half  -> out[i] = 960*re * win[i] + synth_mem[i]
    //   1 <= p <= 480:      time sample 480 + i, i = 480 - p, of the second half -> synth_mem[i] = 960*re * win[479 - i]
    float *sm = synth_mem + (size_t)s * PN_FRAME;
    float smv[15];
#pragma unroll
    for (int q = 0; q < 15; q++) {
      const int p = lane + 64 * (q % 3) + 192 * (q / 3);
      const bool lo = p == 0 || p > PN_FRAME;
      smv[q] = lo ? sm[p == 0 ? 0 : PN_WINDOW - p] : 0.f;
    }
    PN_WAVE_SYNC_GLOBAL();                     // the loads above are this stream's old overlap memory: read before it is rewritten
#pragma unroll
    for (int q = 0; q < 15; q++) {
      const int p = lane + 64 * (q % 3) + 192 * (q / 3);
      const float re = w[q % 3][q / 3].x;
      if (p == 0 || p > PN_FRAME) {
        const int i = p == 0 ? 0 : PN_WINDOW - p;
        const float o = (PN_WINDOW * re) * SH.win[i] + smv[q];
        if (sizeof(TOut) == 2) out[(size_t)s * PN_FRAME + i] = (TOut)pn_f2s(o * 32768);
        else out[(size_t)s * PN_FRAME + i] = (TOut)o;
      } else {
        sm[PN_FRAME - p] = (PN_WINDOW * re) * SH.win[p - 1];
      }
    }
    PN_WAVE_SYNC();
  }
}

// ---- launchers -------------------------------------------------------------------------------
// blocks_per_cu: 2 fills the CUs (80 KB LDS each); 1 leaves half of every CU's LDS/registers free so
// that an MFMA-bound network kernel of another frame can be co-resident (pipelined mode)
static inline int pn_dsp_grid(int n_streams, int blocks_per_cu, int grid_cap) {
  const int need = (n_streams + WPB - 1) / WPB;
  const int full = PN_DSP_WAVES_PER_SIMD * 4 / WPB;
  const int cap = grid_cap > 0 ? grid_cap : 256 * ((blocks_per_cu > 0 && blocks_per_cu < full) ? blocks_per_cu : full);
  return need < cap ? need : cap;
}

void pn_launch_backend(hipStream_t st, const PnTables *T, int n_streams, const float2 *Xs, const float2 *Ps,
                       const float *gr, const float *ex_postfilter, const int *silence, float *synth_mem, void *out,
                       int out_is_i16, int grid_cap) {
  const int grid = pn_dsp_grid(n_streams, 0, grid_cap);
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
