// Phase-split front end, pitch kernel (gfx950): FOUR streams per wavefront, 16 lanes each, two waves per SIMD.
//
//   pn_fe_pitch_kernel   pitch_downsample + LPC whitening, pitch_search (coarse / fine cross-correlation +
//                        find_best_pitch), remove_doubling  ==  the pitch analysis of compute_frame_features
//                        (reference denoise.cpp:399-414; pitch.cpp:148-216, 283-386, 424-527; celt_lpc.cpp:37-88,198-279)
//   in:  the history ring (logical samples [1632,3360) — no sample of the newest frame, so this kernel does not depend
//        on the spectral-in kernel of the same frame), last_period / last_gain of the previous frame
//   out: last_period (the pitch index the spectral-out kernel filters at), last_gain, features 68 (period) and 69 (corr)
//
// What the measurements of round 3 say about this work (profiles/r03a_valu_issue_probe.log): a lone wave issues one
// instruction per 4.5 cycles whatever its kind and a dependent add costs no more than an independent one — the serial
// chains are ISSUE-bound, not latency-bound; two waves per SIMD double the rate; from there on the CU-wide LDS pipe is
// the limiter (a ds_read_b32 costs two LDS cycles per wave-instruction, a bank conflict doubles it).  So this kernel is
// built for (a) two waves per SIMD: 5056 bytes of LDS per stream (two 16-stream blocks per CU) and <= 256 registers
// without spills, and (b) few LDS cycles: the 4x-decimated cross-correlation keeps a sliding window of its per-lane
// operand in registers (each lane owns 11 CONSECUTIVE lags: one new value per step serves 11 multiply-adds), the
// whitening FIR runs in place (descending, no second buffer), the sparse fine search never materialises its 294-entry
// correlation array, and yy_lookup values are captured in flight instead of being stored.
//
// Numerics contract: as pn_dsp_fe.hip — every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (-ffp-contract=off; divide/sqrt correctly rounded; double islands in double);
// every order-sensitive sum is the reference's sequential chain on one lane.  Bit-identical to the single-launch kernel.
#define PN_FE_G 4
#include "pn_dsp_fe_helpers.inc"
#include <stdlib.h>
#include "pn_launch.h"

#define FP_SPB 16                       // streams per block (4 waves)
#define FP_THREADS 256
#define FP_NCH 11                       // coarse lags per lane (lane l owns lags 11 l .. 11 l + 10; 16 * 11 >= 147)

// per-stream LDS slice, in floats
#define FP_PBUF 0                       // [0,864)     decimated signal, whitened in place
#define FP_SCR 864                      // [864,1264)  scratch:
#define FP_XC (FP_SCR + 0)              //   [0,176)   coarse xcorr (lag 11 l + c), later p|q of yy_lookup (128)
#define FP_D (FP_SCR + 176)             //   [176,324) d[] of the coarse best-pitch scan (148); later the 64-float d block of the fine scan
#define FP_SQ (FP_SCR + 324)            //   [324,388) 64-float broadcast scratch
#ifndef PN_FP_ROWS
#define PN_FP_ROWS 0                    // 1: consecutive-lag chains (autocorrelation, final three) read one row per 12 steps and shift it with DPP
                                        //    (fewer LDS cycles, one more VALU op per step: pays only while the LDS pipe is the limiter)
#endif
#ifndef PN_FP_PAIRS
#define PN_FP_PAIRS 1                   // 1: remove_doubling's arbitrary-lag chains read aligned pairs (fp_chain2_pairs)
#endif
#define FP_SLICE 1264                   // 5056 bytes; 16 streams = 80 896 bytes per block, two blocks per CU


// ---- serial chains with few LDS cycles --------------------------------------------------------------------------------
// (profiles/r03c_fe_split_v1_pmc.txt: this kernel is bound by the CU's LDS pipe; 42 % of its LDS cycles were bank conflicts
// of remove_doubling's per-lane arbitrary-lag reads, 9.5 conflict cycles per ds_read_b32.)

// v[lane] -> v[lane + n] within the 16-lane row (n = 0..15; lanes whose source falls off the row keep garbage that
// their caller never uses): v_mov_b32_dpp row_shl:n, which the compiler folds into the multiply that consumes it
template <int n>
__device__ __forceinline__ float fp_row_shl(float v) {
  if (n == 0) return v;
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x100 + (n & 15), 0xf, 0xf, true));
}
// acc + sum_{j<N} a[j] * b[lane k: j + k], adds strictly in j order, for chains whose per-lane operands are CONSECUTIVE:
// lane k of the group (k <= 15 - 11) correlates the group-uniform a[] against b[j + k].  One ds_read_b32 (the 16 lanes read
// b[j0 .. j0+15]) then serves 12 steps through row shifts; a[] is read four steps per ds_read_b128 at a uniform address.
// a + j0 must be 16-byte aligned for every block start j0 (multiples of 12: a itself 16-byte aligned).
template <int N>
__device__ __forceinline__ float fp_chain_rows(const float *a, const float *b, int l, float acc) {
  static_assert(N % 4 == 0, "N");
  constexpr int NB12 = N / 12, R = N % 12;
  float4 a0[3], a1[3]; float v0, v1;
#define FP_CR_LOAD(av, vv, blk) do {                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; q_++) (av)[q_] = *reinterpret_cast<const float4 *>(a + 12 * (blk) + 4 * q_); \
    (vv) = b[12 * (blk) + l];                                                                  \
  } while (0)
#define FP_CR_MAC(av, vv) do {                                                                  \
    acc = acc + (av)[0].x * fp_row_shl<0>(vv); acc = acc + (av)[0].y * fp_row_shl<1>(vv);       \
    acc = acc + (av)[0].z * fp_row_shl<2>(vv); acc = acc + (av)[0].w * fp_row_shl<3>(vv);       \
    acc = acc + (av)[1].x * fp_row_shl<4>(vv); acc = acc + (av)[1].y * fp_row_shl<5>(vv);       \
    acc = acc + (av)[1].z * fp_row_shl<6>(vv); acc = acc + (av)[1].w * fp_row_shl<7>(vv);       \
    acc = acc + (av)[2].x * fp_row_shl<8>(vv); acc = acc + (av)[2].y * fp_row_shl<9>(vv);       \
    acc = acc + (av)[2].z * fp_row_shl<10>(vv); acc = acc + (av)[2].w * fp_row_shl<11>(vv);     \
  } while (0)
  FP_CR_LOAD(a0, v0, 0);
#pragma unroll 1
  for (int blk = 0; blk < NB12; blk += 2) {
    if (blk + 1 < NB12) FP_CR_LOAD(a1, v1, blk + 1);
    FP_CR_MAC(a0, v0);
    if (blk + 2 < NB12) FP_CR_LOAD(a0, v0, blk + 2);
    if (blk + 1 < NB12) FP_CR_MAC(a1, v1);
  }
  if (R) {                                          // R in {4, 8}
    float4 ar[2]; float vr;
#pragma unroll
    for (int q = 0; q < R / 4; q++) ar[q] = *reinterpret_cast<const float4 *>(a + 12 * NB12 + 4 * q);
    vr = b[12 * NB12 + l];
    acc = acc + ar[0].x * fp_row_shl<0>(vr); acc = acc + ar[0].y * fp_row_shl<1>(vr);
    acc = acc + ar[0].z * fp_row_shl<2>(vr); acc = acc + ar[0].w * fp_row_shl<3>(vr);
    if (R > 4) {
      acc = acc + ar[1].x * fp_row_shl<4>(vr); acc = acc + ar[1].y * fp_row_shl<5>(vr);
      acc = acc + ar[1].z * fp_row_shl<6>(vr); acc = acc + ar[1].w * fp_row_shl<7>(vr);
    }
  }
#undef FP_CR_LOAD
#undef FP_CR_MAC
  return acc;
}

// Two chains sharing the uniform operand, per-lane operands at ARBITRARY lags (remove_doubling's 28 + 2 inner products):
// acc1 += a . b1, acc2 += a . b2, adds strictly in j order.  Every lane reads its operands as 8-byte ALIGNED pairs
// (ds_read_b64: 64 banks, half the instructions) and picks the run that starts at its own parity (o1 / o2 = the parity of
// the lane's first element: 1 = its run starts at the second float of the first pair): one v_cndmask per operand
// instead of ~10 LDS bank-conflict cycles per scalar read.  b1 - o1, b2 - o2 must be 8-byte aligned.
template <int N>
__device__ __forceinline__ void fp_chain2_pairs(const float *a, const float *b1, bool o1, const float *b2, bool o2,
                                                float &acc1, float &acc2) {
  constexpr int U = 16, NF = N / U;
  static_assert(N % U == 0 && NF % 2 == 0, "N");
  typedef float fp_f2 __attribute__((ext_vector_type(2)));
  // explicit LDS address space + volatile: each pair stays ONE ds_read_b64 (2 LDS cycles); as plain loads the compiler
  // either splits them (ds_read2_b32, alignment unproven) or merges two into ds_read2_b64 (8 cycles per two pairs)
  typedef __attribute__((address_space(3))) const volatile fp_f2 fp_lds_f2;
  fp_lds_f2 *p1 = (fp_lds_f2 *)(b1 - (o1 ? 1 : 0)), *p2 = (fp_lds_f2 *)(b2 - (o2 ? 1 : 0));
  // the pairs are unpacked into scalars at once: element u of the lane's run is f[u] or f[u + 1] (ONE v_cndmask per
  // operand; left as vector lanes the compiler turns the choice into a dynamic vector index = a chain of 16 selects)
  float4 a0[4], a1[4]; float r0[18], s0[18], r1[18], s1[18];
#define FP_C2_LOAD(av, rv, sv, blk) do {                                                        \
    _Pragma("unroll") for (int v_ = 0; v_ < 4; v_++) (av)[v_] = *reinterpret_cast<const float4 *>(a + 16 * (blk) + 4 * v_); \
    _Pragma("unroll") for (int u_ = 0; u_ < 9; u_++) {                                          \
      const fp_f2 t1_ = p1[8 * (blk) + u_], t2_ = p2[8 * (blk) + u_];                           \
      (rv)[2 * u_] = t1_.x; (rv)[2 * u_ + 1] = t1_.y; (sv)[2 * u_] = t2_.x; (sv)[2 * u_ + 1] = t2_.y; } \
  } while (0)
#define FP_C2_MAC(av, rv, sv) do {                                                              \
    _Pragma("unroll") for (int v_ = 0; v_ < 4; v_++) {                                          \
      const float ax_[4] = {(av)[v_].x, (av)[v_].y, (av)[v_].z, (av)[v_].w};                    \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; e_++) {                                        \
        const float y1_ = o1 ? (rv)[4 * v_ + e_ + 1] : (rv)[4 * v_ + e_];                       \
        const float y2_ = o2 ? (sv)[4 * v_ + e_ + 1] : (sv)[4 * v_ + e_];                       \
        acc1 = acc1 + ax_[e_] * y1_;                                                            \
        acc2 = acc2 + ax_[e_] * y2_;                                                            \
      }                                                                                         \
    }                                                                                           \
  } while (0)
  FP_C2_LOAD(a0, r0, s0, 0);
#pragma unroll 1
  for (int blk = 0; blk < NF; blk += 2) {
    FP_C2_LOAD(a1, r1, s1, blk + 1);
    FP_C2_MAC(a0, r0, s0);
    if (blk + 2 < NF) FP_C2_LOAD(a0, r0, s0, blk + 2);
    FP_C2_MAC(a1, r1, s1);
  }
#undef FP_C2_LOAD
#undef FP_C2_MAC
}

// find_best_pitch for the fine search (pitch.cpp:46-104 on the sparse xcorr of pitch.cpp:344-361): only the <= 10 lags
// within +-2 of twice the two coarse candidates carry a correlation, every other entry is 0 and skipped by the
// reference's `if (xcorr[i] > 0)`.  The running energy still visits all 294 candidates; each lane captures it at its own
// candidate, then the candidates are replayed in ascending lag order through the reference's (best, second best) update.
// cidx/cval/cact: this lane's candidate lag, max(-1, sum) and "inside [0,294)"; lanes >= 10 are inactive.
__device__ __forceinline__ void fp_fine_best_pitch(const float *y, float *sq, float *dblk, int l, int gb, int cidx, float cval,
                                                   bool cact, int dup_lo, int &bp0_out, int &bp1_out) {
  constexpr int LEN = 480, MAXP = 294, MP4 = 296;
  // initial energy Syy = 1 + sum_{j<LEN} y[j]^2, j ascending (pitch.cpp:62-63); squares lane-parallel 64 at a time
  float Syy = 1.0f;
#pragma unroll 1
  for (int blk = 0; blk < LEN / 64 + 1; blk++) {
    float yv[4];
#pragma unroll
    for (int w = 0; w < 4; w++) { const int j = 64 * blk + l + L * w; yv[w] = y[j < LEN ? j : 0]; }
    PN_WAVE_SYNC();
#pragma unroll
    for (int w = 0; w < 4; w++) sq[l + L * w] = yv[w] * yv[w];
    PN_WAVE_SYNC();
    if (blk < LEN / 64) Syy = fe_sum_sq<16>(sq, Syy);
    else Syy = fe_sum_sq<(LEN % 64) / 4>(sq, Syy);
  }
  // running energy over the candidates, captured where this lane's candidate sits (before that candidate's update)
  float cap = 0.f;
#pragma unroll 1
  for (int blk = 0; blk < (MP4 + 63) / 64; blk++) {
    float dv[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int i = 64 * blk + l + L * w, ic = i < MAXP ? i : MAXP - 1;
      const float a = y[ic + LEN], c = y[ic];
      dv[w] = a * a - c * c;
    }
    PN_WAVE_SYNC();
#pragma unroll
    for (int w = 0; w < 4; w++) dblk[l + L * w] = dv[w];
    PN_WAVE_SYNC();
    const int rel = cidx - 64 * blk;
#pragma unroll
    for (int h0 = 0; h0 < 16; h0 += 8) {
      float4 d4[8];
#pragma unroll
      for (int v = 0; v < 8; v++) d4[v] = *reinterpret_cast<const float4 *>(dblk + 4 * (h0 + v));
#pragma unroll
      for (int v = 0; v < 8; v++) {
        const float dd[4] = {d4[v].x, d4[v].y, d4[v].z, d4[v].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          cap = (rel == 4 * (h0 + v) + e) ? Syy : cap;
          const float t = Syy + dd[e];
          Syy = (1 > t) ? 1 : t;
        }
      }
    }
  }
  // candidates: positive correlation, not a repeat of a lane of the first window
  const bool cand = cact && cval > 0 && !(l >= 5 && cidx >= dup_lo && cidx <= dup_lo + 4);
  float x16 = cval;
  x16 *= 1e-12f;
  const float num = x16 * x16;
  int rank = 0;
#pragma unroll
  for (int m = 0; m < 10; m++) {
    const int cm = __shfl(cidx, gb + m);
    const int km = __shfl((int)cand, gb + m);
    rank += (km && cm < cidx) ? 1 : 0;
  }
  float bn0 = -1.f, bn1 = -1.f, bd0 = 0.f, bd1 = 0.f; int bp0 = 0, bp1 = 1;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned field = (unsigned)((__ballot(cand && rank == r) >> gb) & 0xffffull);
    const int src = gb + (field ? __builtin_ctz(field) : 0);
    const float n_r = __shfl(num, src), s_r = __shfl(cap, src);
    const int i_r = __shfl(cidx, src);
    const float nm = field ? n_r : __builtin_nanf("");      // NaN: no r-th candidate in this stream (every comparison false)
    const bool c1 = nm * bd1 > bn1 * s_r;
    const bool c0 = c1 && (nm * bd0 > bn0 * s_r);
    bn1 = c0 ? bn0 : (c1 ? nm : bn1); bd1 = c0 ? bd0 : (c1 ? s_r : bd1); bp1 = c0 ? bp0 : (c1 ? i_r : bp1);
    bn0 = c0 ? nm : bn0; bd0 = c0 ? s_r : bd0; bp0 = c0 ? i_r : bp0;
  }
  bp0_out = bp0; bp1_out = bp1;
}

// xcorr[t] of the sparse fine correlation: the value of an active lane whose lag is t, else 0
__device__ __forceinline__ float fp_sparse_at(int t, int gb, int cidx, float cval, bool cact) {
  const unsigned field = (unsigned)((__ballot(cact && cidx == t) >> gb) & 0xffffull);
  const float v = __shfl(cval, gb + (field ? __builtin_ctz(field) : 0));
  return field ? v : 0.f;
}

__global__ __launch_bounds__(FP_THREADS, 2) void pn_fe_pitch_kernel(
    int n_streams, int frame_t, const float *__restrict__ hist, float *__restrict__ feat,
    int *__restrict__ last_period, float *__restrict__ last_gain, float *__restrict__ aux, int stagger) {
  __shared__ __attribute__((aligned(16))) float SH[FP_SPB * FP_SLICE];
  const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid >> 6;
  const int sub = lane / L, l = lane % L, gb = sub * L;
  const int slice0 = (wave * G + sub) * FP_SLICE;
  const int base_slot0 = (frame_t + 1) % PN_HIST_FRAMES;   // slot of logical frame 0 (oldest)

  // Every wave runs the same sequence of VALU-heavy (decimated correlation) and LDS-heavy (remove_doubling) phases, and
  // all waves of a launch start together: left alone they load the VALUs and then the LDS pipe in unison.  Waves 2 and 3
  // of every block start `stagger` sleep quanta (of 127 x 64 cycles) late, so that half of a CU's waves are in the other
  // kind of phase.  No barrier follows: the waves of a block never exchange data.
  if (wave >= 2)
    for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(127);

  for (int s0 = (blockIdx.x * (FP_SPB / G) + wave) * G; s0 < n_streams; s0 += gridDim.x * FP_SPB) {
    const int s = s0 + sub;
    if (s < n_streams) {
      const float *h = hist + (size_t)s * PN_HIST_STRIDE;
      // opaque copy: keeps the ~60 loop-invariant ring offsets of the loads below from being hoisted out of the stream
      // loop and held (spilled) across the whole kernel
      int base_slot = base_slot0, slice = slice0;
      asm volatile("" : "+v"(base_slot), "+v"(slice));     // (same for the LDS addresses: one base register + immediates)
      float *buf = SH + slice;
      float *pbuf = buf + FP_PBUF, *raw = buf + FP_PBUF, *xcs = buf + FP_XC, *d1 = buf + FP_D, *sq64 = buf + FP_SQ;
      // -- pitch_downsample (pitch.cpp:148-216) of pitch_buf == comb_buf[1632,3360): outputs 2m, 2m+1 need x[4m-1 .. 4m+3]
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
        constexpr int NM = 14;                             // 2 x 14 x 16 = 448 >= 432 pairs
        float4 dv[NM]; float dm1[NM];
#pragma unroll
        for (int it = 0; it < NM; it++) {
          const int mm = l + L * (half * NM + it), m = mm < 432 ? mm : 0;
          dv[it] = *reinterpret_cast<const float4 *>(h + fe_ring(1632 + 4 * m, base_slot));
          dm1[it] = h[fe_ring(1632 + (m > 0 ? 4 * m - 1 : 0), base_slot)];
        }
#pragma unroll
        for (int it = 0; it < NM; it++) {
          const int m = l + L * (half * NM + it);
          if (m >= 432) continue;
          const float4 v = dv[it];
          const float o0 = (m == 0) ? .5f * (.5f * (v.y) + v.x) : .5f * (.5f * (dm1[it] + v.y) + v.x);
          const float o1 = .5f * (.5f * (v.y + v.w) + v.z);
          *reinterpret_cast<float2 *>(raw + 2 * m) = make_float2(o0, o1);
        }
      }
      PN_WAVE_SYNC();
      // _celt_autocorr (celt_lpc.cpp:198-279): lane k holds lag k (lanes > 4 shadow lag 4)
      float ac[5];
      {
        const int lag = l < 4 ? l : 4;
#if PN_FP_ROWS
        float ack = fp_chain_rows<860>(raw, raw, l, 0.f);     // lane k <= 4: sum_i raw[i] * raw[i + k]; lanes > 4 are never read
#else
        float ack = fe_chain<860>(raw, raw + lag, 0.f);
#endif
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
      // celt_fir5 (pitch.cpp:106-145) with zero memory == a pure 5-tap FIR: IN PLACE, highest index first — an output
      // only reads inputs at its own index and below, so descending blocks never read a value already overwritten;
      // inside a block all reads are issued before the first write
#pragma unroll 1
      for (int blk = 8; blk >= 0; blk--) {                 // 9 blocks of 6 x 16 outputs = 864
        float x0[6], x1[6], x2[6], x3[6], x4[6], x5[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const int i = l + L * (6 * blk + q);
          x0[q] = raw[i];
          x1[q] = i >= 1 ? raw[i - 1] : 0.f; x2[q] = i >= 2 ? raw[i - 2] : 0.f; x3[q] = i >= 3 ? raw[i - 3] : 0.f;
          x4[q] = i >= 4 ? raw[i - 4] : 0.f; x5[q] = i >= 5 ? raw[i - 5] : 0.f;
        }
        PN_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const int i = l + L * (6 * blk + q);
          float sum = x0[q];
          sum = sum + lpc2[0] * x1[q];
          sum = sum + lpc2[1] * x2[q];
          sum = sum + lpc2[2] * x3[q];
          sum = sum + lpc2[3] * x4[q];
          sum = sum + lpc2[4] * x5[q];
          pbuf[i] = sum;
        }
        PN_WAVE_SYNC();
      }

      // -- pitch_search (pitch.cpp:283-386): x_lp = pbuf+384, y = pbuf, len 960, max_pitch 588 ------------------------------
      // coarse (4x decimation, 147 lags x 240 steps): x_lp4[j] = pbuf[384+2j] (group-uniform), y_lp4[j] = pbuf[2j].
      // Lane l owns the 11 consecutive lags 11 l + c.  Twelve registers hold y_lp4[11 l + e] for e = j .. j+11 (register
      // e mod 12): step j uses e = j .. j+10 and then refills the register of e = j with e = j + 12, needed two steps later.
      {
        float acc[FP_NCH];
#pragma unroll
        for (int c = 0; c < FP_NCH; c++) acc[c] = 0;
        const float *yb = pbuf + 2 * (FP_NCH * l);
        const float *xb = pbuf + 384;
        float R[12];
#pragma unroll
        for (int e = 0; e < 12; e++) R[e] = yb[2 * e];
        float4 xq[6];
#pragma unroll
        for (int v = 0; v < 6; v++) xq[v] = *reinterpret_cast<const float4 *>(xb + 4 * v);
#pragma unroll 1
        for (int j0 = 0; j0 < 240; j0 += 12) {
          float4 xn[6];
          const int jn = j0 + 12 < 240 ? j0 + 12 : j0;       // the last block re-reads itself
#pragma unroll
          for (int v = 0; v < 6; v++) xn[v] = *reinterpret_cast<const float4 *>(xb + 2 * jn + 4 * v);
#pragma unroll
          for (int u = 0; u < 12; u++) {
            const float xj = (u & 1) ? xq[u >> 1].z : xq[u >> 1].x;      // pbuf[384 + 2 (j0 + u)]
#pragma unroll
            for (int c = 0; c < FP_NCH; c++) acc[c] = acc[c] + xj * R[(u + c) % 12];
            R[u] = yb[2 * (j0 + u + 12)];
          }
#pragma unroll
          for (int v = 0; v < 6; v++) xq[v] = xn[v];
        }
#pragma unroll
        for (int c = 0; c < FP_NCH; c++) xcs[FP_NCH * l + c] = acc[c];
      }
      PN_WAVE_SYNC();
      int bp0, bp1;
      fe_find_best_pitch<240, 147, 2>(xcs, pbuf, sq64, d1, l, bp0, bp1);
      PN_WAVE_SYNC();
      // fine (2x decimation): only lags within +-2 of 2*best (pitch.cpp:344-361); every other xcorr entry is 0
      int cidx; float cval; bool cact;
      const int dup_lo = 2 * bp0 - 2;
      {
        cidx = (l < 5) ? (2 * bp0 - 2 + l) : (2 * bp1 - 2 + (l - 5));
        cact = l < 10 && cidx >= 0 && cidx < 294;
        const float sum = fe_chain<480>(pbuf + 384, pbuf + (cact ? cidx : 0), 0.f);
        cval = (-1 > sum) ? -1 : sum;
      }
      fp_fine_best_pitch(pbuf, sq64, d1, l, gb, cidx, cval, cact, dup_lo, bp0, bp1);
      int offset = 0;
      if (bp0 > 0 && bp0 < 294 - 1) {
        const float a = fp_sparse_at(bp0 - 1, gb, cidx, cval, cact), b = fp_sparse_at(bp0, gb, cidx, cval, cact),
                    c = fp_sparse_at(bp0 + 1, gb, cidx, cval, cact);
        if ((c - a) > .7f * (b - a)) offset = 1;
        else if ((a - c) > .7f * (b - c)) offset = -1;
      }
      const float pitch_corr = fp_sparse_at(bp0, gb, cidx, cval, cact);
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
        // lane 0: xx ; lane 1: xy(T0) ; lanes 2..15: k = l: xy(T1_k) and xy2(T1b_k)
        int lag1 = 0, lag2 = 0, T1 = 0, T1b = 0;
        const int k = l;
        if (l == 1) { lag1 = T0; lag2 = T0; }
        else if (l >= 2 && l < 16) {
          // second_check[16] = {0,0,3,2,3,2,5,2,3,2,3,2,5,2,3,2} (pitch.cpp:423) as nibbles of one constant (no table load)
          const int second_check_k = (int)((0x2325232325232300ull >> (4 * k)) & 7);
          T1 = (2 * T0 + k) / (2 * k);
          if (k == 2) { if (T1 + T0 > 384) T1b = T0; else T1b = T0 + T1; }
          else T1b = (2 * second_check_k * T0 + k) / (2 * k);
          lag1 = T1; lag2 = T1b;
        }
        float dot1 = 0, dot2 = 0;
        // x = pbuf + 384 sits at an even float of the (even) slice: the parity of x - lag is the parity of the lag
#if PN_FP_PAIRS
        fp_chain2_pairs<480>(x, x - lag1, (lag1 & 1) != 0, x - lag2, (lag2 & 1) != 0, dot1, dot2);
#else
        fe_chain2<480>(x, x - lag1, x - lag2, dot1, dot2);
#endif
        const float xx = __shfl(dot1, gb);
        float xy = __shfl(dot1, gb + 1);
        // yy_lookup (pitch.cpp:449-455): strictly sequential running energy, group-uniform.  Squares formed lane-parallel,
        // 64 at a time, into a broadcast scratch; the recurrence reads them 4 per ds_read_b128.  Instead of storing the
        // 384 clamped values, every lane captures the two it will look up (lane 1 looks up T0 twice).
        float y1 = xx, y2 = xx;                  // yy_lookup[0] = xx
        {
          float *pq = xcs;                       // p[64] | q[64]
          float yy = xx;
#pragma unroll 1
          for (int blk = 0; blk < 6; blk++) {        // i = 1 + 64*blk + u, u < 64  (384 = 6*64)
            float pa[4], qa[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
              const int i = 1 + 64 * blk + l + L * w;
              const float a = x[-i], c = x[480 - i];
              pa[w] = a * a; qa[w] = c * c;
            }
            PN_WAVE_SYNC();
#pragma unroll
            for (int w = 0; w < 4; w++) { pq[l + L * w] = pa[w]; pq[64 + l + L * w] = qa[w]; }
            PN_WAVE_SYNC();
            const int r1 = lag1 - 1 - 64 * blk, r2 = lag2 - 1 - 64 * blk;
#pragma unroll
            for (int h0 = 0; h0 < 16; h0 += 8) {       // operands of 32 steps read before the chain
              float4 p4[8], q4[8];
#pragma unroll
              for (int v = 0; v < 8; v++) {
                p4[v] = *reinterpret_cast<const float4 *>(pq + 4 * (h0 + v));
                q4[v] = *reinterpret_cast<const float4 *>(pq + 64 + 4 * (h0 + v));
              }
#pragma unroll
              for (int v = 0; v < 8; v++) {
                const float pp[4] = {p4[v].x, p4[v].y, p4[v].z, p4[v].w}, qq[4] = {q4[v].x, q4[v].y, q4[v].z, q4[v].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  yy = yy + pp[e] - qq[e];
                  const float o = (0 > yy) ? 0 : yy;
                  y1 = (r1 == 4 * (h0 + v) + e) ? o : y1;
                  y2 = (r2 == 4 * (h0 + v) + e) ? o : y2;
                }
              }
            }
          }
        }
        PN_WAVE_SYNC();
        float yy = __shfl(y1, gb + 1);           // yy_lookup[T0]
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
          yyk = .5f * (y1 + y2);
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
        // xcorr[k] = x . (x - (T + k - 1)), k = 0..2 (pitch.cpp:511-512): lane lam holds k = 2 - lam, so that the operands of
        // consecutive lanes are consecutive floats (x[j - Tsel - 1 + lam]) and one row read serves 12 steps
#if PN_FP_ROWS
        const float xc = fp_chain_rows<480>(x, x - (Tsel + 1), l, 0.f);
#else
        const float xc = fe_chain<480>(x, x - (Tsel + 1) + (l < 3 ? l : 2), 0.f);
#endif
        const float xc0 = __shfl(xc, gb + 2), xc1 = __shfl(xc, gb + 1), xc2 = __shfl(xc, gb);
        int off2;
        if ((xc2 - xc0) > .7f * (xc1 - xc0)) off2 = 1;
        else if ((xc0 - xc2) > .7f * (xc1 - xc2)) off2 = -1;
        else off2 = 0;
        if (pg > g) pg = g;
        pitch_index = 2 * Tsel + off2;
        if (pitch_index < PN_PITCH_MIN) pitch_index = PN_PITCH_MIN;
      }
      if (l == 0) {
        last_period[s] = pitch_index; last_gain[s] = pg;
        float *f = feat + (size_t)s * PN_FEAT_STRIDE;
        f[68] = (float)pitch_index / (PN_PITCH_MAX - 3 * PN_PITCH_MIN);     // create_features (denoise.cpp:494-495)
        f[69] = pitch_corr;
        if (aux) aux[(size_t)s * PN_AUX_STRIDE + 2 * PN_NB] = pitch_corr;
      }
      PN_WAVE_SYNC();
    }
  }
}

void pn_launch_fe_pitch(hipStream_t st, int n_streams, int64_t frame, const float *hist, float *feat, int *last_period,
                        float *last_gain, float *aux, int grid_cap) {
  const int need = (n_streams + FP_SPB - 1) / FP_SPB;
  const int cap = 256 * 2;                               // two LDS-resident blocks on each of 256 CUs
  int grid = need < cap ? need : cap;
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  static const int stagger = getenv("PERCEPNET_FP_STAGGER") ? atoi(getenv("PERCEPNET_FP_STAGGER")) : 0;
  // only worth it when a block walks several stream groups (the delay is paid once per launch)
  hipLaunchKernelGGL(pn_fe_pitch_kernel, dim3(grid), dim3(FP_THREADS), 0, st, n_streams, (int)(frame % PN_HIST_FRAMES), hist,
                     feat, last_period, last_gain, aux, need >= 4 * cap ? stagger : 0);
}
