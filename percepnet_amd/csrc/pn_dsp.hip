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
    // are exactly 0 (interp_band_gain never writes them).
    // Round 6: in two phases.  (A) every lane takes PAIRS of bins (one coalesced 16-byte load per spectrum: 4 + 4 load
    // instructions per stream, each bin's filter / gain arithmetic done once) and parks 1/960 of the result in the first 3200
    // bytes of the FFT buffer; (B) lane l < 60 collects the 16 inputs 4l + c + 240j of its four first-stage butterflies from
    // there (mirrored inputs: imaginary part negated, which commutes with the scaling bit for bit; absent bins: +0 / -0 as
    // before).  Until round 5 every INPUT was fetched and filtered separately: 32 strided 8-byte loads per lane whose lines
    // the CU's 16 waves evicted from L1 between the four butterflies, and each bin's arithmetic done twice.
    // (round 5: a two-register-set software pipeline over the four butterflies — the spectra of c + 1 requested before the
    // arithmetic of c — unrolls to 27 spilled registers at four waves per SIMD: 0.296 vs 0.250 ms, profiles/r05_front_end_variants.log)
    {
      const float4 *X4 = reinterpret_cast<const float4 *>(Xspec + (size_t)s * PN_SPEC_BINS);
      const float4 *P4 = reinterpret_cast<const float4 *>(Pspec + (size_t)s * PN_SPEC_BINS);
      float4 xq[4], pq[4];
      int la = lane;
      asm volatile("" : "+v"(la));                          // (opaque, as lq below)
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int m = la + 64 * it, mc = m < PN_SPEC_BINS / 2 ? m : PN_SPEC_BINS / 2 - 1;
        xq[it] = X4[mc];
        pq[it] = sil ? make_float4(0.f, 0.f, 0.f, 0.f) : P4[mc];
      }
      float4 *Y4 = reinterpret_cast<float4 *>(W.fft);
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int m = la + 64 * it;
        if (m < PN_SPEC_BINS / 2) {
          float2 y[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int k = 2 * m + h;
            float2 x = h ? make_float2(xq[it].z, xq[it].w) : make_float2(xq[it].x, xq[it].y);
            const int b = SH.band[k];
            const float fr = SH.frac[k];
            if (!sil) {
              const float2 p = h ? make_float2(pq[it].z, pq[it].w) : make_float2(pq[it].x, pq[it].y);
              const float rf1 = (1 - fr) * W.e[2][b] + fr * W.e[2][b + 1];
              x.x = rf1 * x.x; x.y = rf1 * x.y;
              const float rf2 = (1 - fr) * W.e[1][b] + fr * W.e[1][b + 1];
              x.x += rf2 * p.x; x.y += rf2 * p.y;
            }
            const float gf = (1 - fr) * W.e[0][b] + fr * W.e[0][b + 1];
            x.x *= gf; x.y *= gf;
            y[h] = make_float2(scale * x.x, scale * x.y);
          }
          Y4[m] = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
        }
      }
    }
    PN_WAVE_SYNC();
    __builtin_amdgcn_sched_barrier(0);
    {
      float2 fin[4][4];                                     // [butterfly c][input j]
      // input i = 4l + c + 240j: bin i for j < 2, bin 960 - i (conjugated) for j >= 2 — except i = 480 itself (l = c = 0, j = 2:
      // its own mirror, not conjugated).  Two base addresses + immediates; a bin index past 399 (up to 480) reads stale bytes
      // of the FFT buffer that the select below replaces by the reference's exact zero
      int lq = lc;
      asm volatile("" : "+v"(lq));                          // opaque: the lane's addresses and masks are rebuilt per stream, not hoisted and spilled
      const float2 *Yd = W.fft + 4 * lq, *Ym = W.fft + (PN_WINDOW - 4 * lq);
#pragma unroll
      for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int o = c + 240 * j, kk = j < 2 ? 4 * lq + o : PN_WINDOW - 4 * lq - o;
          float2 x = j < 2 ? Yd[o] : Ym[-o];
          if (kk >= PN_SPEC_BINS) x = make_float2(0.f, 0.f);
          if (j == 3 || (j == 2 && (c > 0 || lq > 0))) x.y = -x.y;
          fin[c][j] = x;
        }
      PN_WAVE_SYNC();                                       // every input is in registers: the first stage may overwrite the parked bins
      if (lane < 60) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          fs_bfly4_m1(fin[c]);
          float4 *dst = reinterpret_cast<float4 *>(W.fft + Z.p1off[c]);
          dst[0] = make_float4(fin[c][0].x, fin[c][0].y, fin[c][1].x, fin[c][1].y);
          dst[1] = make_float4(fin[c][2].x, fin[c][2].y, fin[c][3].x, fin[c][3].y);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);                      // the twiddle loads of the next passes stay behind the 32 input registers
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
