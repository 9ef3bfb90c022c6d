// Phase-split front end, pitch kernel (gfx950): FOUR streams per wavefront, 16 lanes each, two waves per SIMD.
//
//   pn_fe_pitch_kernel   pitch_downsample + LPC whitening, pitch_search (coarse / fine cross-correlation +
//                        find_best_pitch), remove_doubling  ==  the pitch analysis of compute_frame_features
//                        (reference denoise.cpp:399-414; pitch.cpp:148-216, 283-386, 424-527; celt_lpc.cpp:37-88,198-279)
//   in:  the history ring (logical samples [1632,3360) — no sample of the newest frame, so this kernel does not depend
//        on the spectral-in kernel of the same frame), last_period / last_gain of the previous frame
//   out: last_period (the pitch index the spectral-out kernel filters at), last_gain, features 68 (period) and 69 (corr)
//
// What the measurements say about this work (profiles/r03a_valu_issue_probe.log, profiles/r04p_fe_pitch_variants.log): a
// lone wave issues one instruction per 4.5 cycles whatever its kind and a dependent add costs no more than an independent
// one — the serial chains are ISSUE-bound per wave, not latency-bound; two waves per SIMD double the rate; what is left is
// shared between the CU-wide LDS pipe and every exposed LDS round trip (~150 cycles).  So this kernel is built for
// (a) two waves per SIMD: 5056 bytes of LDS per stream (two 16-stream blocks per CU) and <= 256 registers;
// (b) few instructions per chain step: the 4x-decimated cross-correlation keeps a sliding window of its per-lane operand
//     in registers (each lane owns 11 CONSECUTIVE lags: one new value per step serves 11 multiply-adds), the whitening FIR
//     runs in place, the sparse fine search never materialises its 294-entry correlation array;
// (c) few LDS cycles and no LDS round trip inside a chain: an operand that is UNIFORM within a stream's 16 lanes is read
//     once, lane-distributed (one conflict-free ds_read_b32 = 16 operands), and broadcast by the DPP modifier of the VALU
//     instruction that consumes it (row_newbcast; row_shl for chains over consecutive lags).  Round 3 wrote such operands
//     to a broadcast scratch and read them back four at a time on every lane, and the compiler sank those reads to just
//     before their use: every fourth step of a recurrence waited for the LDS.
//
// Numerics contract: as pn_dsp_fe.hip — every arithmetic step is the reference's operation in the reference's order
// with separate IEEE binary32 rounding (-ffp-contract=off; divide/sqrt correctly rounded; double islands in double);
// every order-sensitive sum is the reference's sequential chain on one lane.  Bit-identical to the single-launch kernel
// (an independent implementation of the same analysis: tests/test_gpu_parity.py compares the two).
#define PN_FE_G 4
#include "pn_dsp_fe_helpers.inc"
#include <stdlib.h>
#include "pn_launch.h"

#define FP_SPB 16                       // streams per block (4 waves)
#define FP_THREADS 256
#define FP_NCH 11                       // coarse lags per lane (lane l owns lags 11 l .. 11 l + 10; 16 * 11 >= 147)
#ifndef PN_FP_DS_DPP
#define PN_FP_DS_DPP 1                   // pitch_downsample: x[4m - 1] from the neighbouring lane (row rotate) instead of a scalar load
#endif
#ifndef PN_FP_ROWS_PRE
#define PN_FP_ROWS_PRE 1                 // autocorrelation and the final three chains: products formed lane-parallel (fp_chain_rows_pre)
#endif
#ifndef PN_FP_DPP_ASM
#define PN_FP_DPP_ASM 2                  // group-uniform recurrences: DPP adds from inline assembly (no hazard padding on the accumulator)
#endif
#ifndef PN_FP_COARSE_PK
#define PN_FP_COARSE_PK 1                // coarse cross-correlation on packed f32 instructions (0: the scalar round-4 loop)
#endif

// per-stream LDS slice, in floats
#define FP_PBUF 0                       // [0,864)     decimated signal, whitened in place
#define FP_SCR 864                      // [864,1264)  scratch: coarse xcorr (176, lag 11 l + c); the fine scan's 304 running energies;
                                        //             yy_lookup's 384 running energies
#define FP_SLICE 1264                   // 5056 bytes; 16 streams = 80 896 bytes per block, two blocks per CU


// ---- DPP operands ------------------------------------------------------------------------------------------------------
// v[lane] -> v[lane + n] within the 16-lane row (lanes whose source falls off the row get 0, which their caller never
// uses): v_mov_b32_dpp row_shl:n, folded by the compiler into the multiply that consumes it
template <int n>
__device__ __forceinline__ float fp_row_shl(float v) {
  if (n == 0) return v;
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x100 + (n & 15), 0xf, 0xf, true));
}
// the value held by lane n of this lane's 16-lane row (= of this stream's group): row_newbcast:n, folded into the add /
// subtract / multiply that consumes it (v_add_f32_dpp ...)
template <int n>
__device__ __forceinline__ float fp_bc(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x150 + (n & 15), 0xf, 0xf, true));
}
// a product whose multiply carries a DPP operand, kept out of the SLP vectoriser's reach: paired into a v_pk_mul_f32 the
// DPP operand would have to be materialised by a v_mov_b32_dpp first
__device__ __forceinline__ float fp_mul_dpp(float dpp, float b) { float p = dpp * b; asm("" : "+v"(p)); return p; }
// acc + (the value lane n of the row holds in v) / acc - (...), as ONE DPP instruction WITHOUT wait states in front of it.
// Written in C++ (acc + fp_bc<n>(v)) the compiler emits the same v_add_f32_dpp but pads every DEPENDENT chain step with
// s_nop: LLVM's hazard recogniser applies "VALU writes a VGPR -> a DPP instruction reads it: 2 wait states" to EVERY register
// a DPP instruction reads, also the plain second operand (the running sum), and a step of these recurrences then costs 14
// ticks instead of ~5 (tools/probes/dpp_chain_probe.hip).  The hazard concerns the operand that goes through the DPP
// cross-lane path — here v, formed long before the chain starts (FP_DPP_SETTLE separates it from its producer) — not the
// forwarded accumulator: the outputs are bit-identical with the padding removed (every parity test; hash of tools/fp_variants.py).
#if PN_FP_DPP_ASM
template <int n>
__device__ __forceinline__ float fp_add_bc(float acc, float v) {
  float r;
  asm("v_add_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(acc), "n"(n & 15));
  return r;
}
template <int n>
__device__ __forceinline__ float fp_sub_bc(float acc, float v) {       // acc - bc(v): v_subrev computes src1 - src0
  float r;
  asm("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(acc), "n"(n & 15));
  return r;
}
// two wait states between the VALU instructions that formed the DPP operands (+v: after them) and the first DPP read
#define FP_DPP_SETTLE(...) asm volatile("s_nop 1" : __VA_ARGS__)
#else
template <int n> __device__ __forceinline__ float fp_add_bc(float acc, float v) { return acc + fp_bc<n>(v); }
template <int n> __device__ __forceinline__ float fp_sub_bc(float acc, float v) { return acc - fp_bc<n>(v); }
#define FP_DPP_SETTLE(...) do {} while (0)
#endif
#define FP_REP12(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11)
#define FP_REP16(M) FP_REP12(M) M(12) M(13) M(14) M(15)

// ---- inner-product chains (celt_inner_prod / xcorr_kernel, pitch.h:53-144): acc + sum_{j<N} a[j] * b[j], adds strictly
// in j order, a[] uniform within the group, b[] per lane --------------------------------------------------------------------
// All three forms run two register sets: the LDS reads of block k+1 are in flight while the serially dependent adds of
// block k execute.  Two things keep it that way: the loops are branch-free (the last trip re-reads a block it does not
// use — after a conditional load the wait at the join is for "everything", prefetch included), and a scheduling fence
// follows every batch of loads (left alone the machine scheduler sinks a batch to just before its first use, merges it
// with the next one and waits for both at once).
#define FP_FENCE() __builtin_amdgcn_sched_barrier(0)

// Per-lane operands at CONSECUTIVE lags: lane k of the group (k <= 15 - 11) correlates a[] against b[j + k].  One
// ds_read_b32 (the 16 lanes read b[j0 .. j0+15]) serves 12 steps through row shifts inside the multiplies; a[] is read four
// steps per ds_read_b128 at a uniform address (a + 12 blk 16-byte aligned for every blk: a itself 16-byte aligned).
// Three register sets: the operands of a block are requested two blocks (24 steps) before their use.
// (Measured alternatives, profiles/r04p_fe_pitch_variants.log: a[] through a row broadcast instead of the ds_read_b128 —
// three instructions per step, no LDS traffic — is slower, this kernel is bound by instruction issue per wave.)
template <int N>
__device__ __forceinline__ float fp_chain_rows(const float *a, const float *b, int l, float acc) {
  static_assert(N % 4 == 0, "N");
  constexpr int NB = N / 12, R = N % 12, NT = NB / 3;
  float4 a0[3], a1[3], a2[3]; float v0, v1, v2;
#define FP_CR_LOAD(av, vv, blk) do {                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; q_++) (av)[q_] = *reinterpret_cast<const float4 *>(a + 12 * (blk) + 4 * q_); \
    (vv) = b[12 * (blk) + l];                                                                  \
  } while (0)
#define FP_CR_STEP(u) if ((u) < nst_) acc = acc + fp_mul_dpp(fp_row_shl<u>(vv_), ax_[u]);
#define FP_CR_MAC(av, vv, nst) do {                                                             \
    const float vv_ = (vv); constexpr int nst_ = (nst);                                         \
    const float ax_[12] = {(av)[0].x, (av)[0].y, (av)[0].z, (av)[0].w, (av)[1].x, (av)[1].y, (av)[1].z, (av)[1].w,      \
                           (av)[2].x, (av)[2].y, (av)[2].z, (av)[2].w};                         \
    FP_REP12(FP_CR_STEP) } while (0)
  FP_CR_LOAD(a0, v0, 0);
  FP_CR_LOAD(a1, v1, 1);
#pragma unroll 1
  for (int t = 0; t < NT; t++) {
    const int blk = 3 * t;
    FP_CR_LOAD(a2, v2, blk + 2);
    FP_FENCE();
    FP_CR_MAC(a0, v0, 12);
    FP_CR_LOAD(a0, v0, blk + 3 < NB ? blk + 3 : NB - 1);
    FP_FENCE();
    FP_CR_MAC(a1, v1, 12);
    FP_CR_LOAD(a1, v1, blk + 4 < NB ? blk + 4 : NB - 1);
    FP_FENCE();
    FP_CR_MAC(a2, v2, 12);
  }
  if (NB % 3 >= 1) FP_CR_MAC(a0, v0, 12);            // blocks 3 NT, 3 NT + 1: requested by the last trip
  if (NB % 3 == 2) FP_CR_MAC(a1, v1, 12);
  if (R) {                                          // R in {4, 8}; reads up to a[N + 11 - R], b[N - R + 15]: inside the slice
    FP_CR_LOAD(a2, v2, NB);
    FP_CR_MAC(a2, v2, R);
  }
#undef FP_CR_LOAD
#undef FP_CR_STEP
#undef FP_CR_MAC
  return acc;
}

#if PN_FP_ROWS_PRE
// acc + (lane (this + n) of the row's value of v), one DPP instruction, no hazard padding on acc (see fp_add_bc)
template <int n>
__device__ __forceinline__ float fp_add_shl(float acc, float v) {
  if (n == 0) return acc + v;
  float r;
  asm("v_add_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(acc), "n"(n & 15));
  return r;
}
// Consecutive-lag chains with the PRODUCTS formed lane-parallel (round 5): lane k < NL of the group accumulates
// sum_{j<N} a[j] * b[j + k] — the same chain, product by product in j order — but the multiply for (lag k, step j0 + i) is done by
// lane k + NL i (i < S = 16 / NL): one v_mul serves S steps of all NL chains, and step i adds lane k + NL i's product to lane k
// through row_shl inside the add.  Per S steps: two per-lane ds_read_b32, one multiply, S adds (fp_chain_rows: per 12 steps 3
// ds_read_b128 + 1 ds_read_b32 + 12 multiplies + 12 adds).  Products past j = N - 1 are +0 (a sum that starts at +0 never is -0).
template <int N, int NL>
__device__ __forceinline__ float fp_chain_rows_pre(const float *a, const float *b, int l, float acc) {
  constexpr int S = 16 / NL, NF = N / S, U = 4, NT = NF / U;       // NF full blocks of S steps, U blocks per trip
  const int i = l / NL, k = l - NL * i;
  const bool on = i < S;
  const float *pa = a + (on ? i : 0), *pb = b + (on ? i + k : 0);
  float xa[U], xb[U], ya[U], yb[U];
#define FP_RP_LOAD(av, bv, blk0) do { _Pragma("unroll") for (int t_ = 0; t_ < U; t_++) { (av)[t_] = pa[S * ((blk0) + t_)]; (bv)[t_] = pb[S * ((blk0) + t_)]; } } while (0)
#define FP_RP_ADDS(q) do { if (S >= 1) acc = fp_add_shl<0>(acc, q); if (S >= 2) acc = fp_add_shl<NL>(acc, q); if (S >= 3) acc = fp_add_shl<2 * NL>(acc, q); \
                           if (S >= 4) acc = fp_add_shl<3 * NL>(acc, q); if (S >= 5) acc = fp_add_shl<4 * NL>(acc, q); } while (0)
#define FP_RP_MAC(av, bv) do { float q_[U]; _Pragma("unroll") for (int t_ = 0; t_ < U; t_++) { q_[t_] = (av)[t_] * (bv)[t_]; asm("" : "+v"(q_[t_])); } \
                               _Pragma("unroll") for (int t_ = 0; t_ < U; t_++) FP_RP_ADDS(q_[t_]); } while (0)
  FP_RP_LOAD(xa, xb, 0);
#pragma unroll 1
  for (int t = 0; t + 1 < NT; t += 2) {
    FP_RP_LOAD(ya, yb, U * (t + 1));
    FP_FENCE();
    FP_RP_MAC(xa, xb);
    FP_RP_LOAD(xa, xb, U * (t + 2 < NT ? t + 2 : NT - 1));      // the last trip re-reads a block it does not use
    FP_FENCE();
    FP_RP_MAC(ya, yb);
  }
  if (NT & 1) FP_RP_MAC(xa, xb);                                   // (requested by the last trip, or the only trip)
  // remaining full blocks and the partial one
#pragma unroll
  for (int blk = NT * U; blk * S < N; blk++) {
    const bool in = blk * S + i < N;
    float q = pa[in ? S * blk : 0] * pb[in ? S * blk : 0];
    q = in ? q : 0.f;
    asm("" : "+v"(q));
    FP_DPP_SETTLE("+v"(q));
    FP_RP_ADDS(q);
  }
#undef FP_RP_LOAD
#undef FP_RP_ADDS
#undef FP_RP_MAC
  return acc;
}
#endif

// Two chains sharing a[], per-lane operands at ARBITRARY lags (remove_doubling's 28 + 2 inner products): acc1 += a . b1,
// acc2 += a . b2.  Every lane reads its operands as 8-byte ALIGNED pairs (ds_read_b64: 64 banks, half the instructions)
// and picks the run that starts at its own parity (o1 / o2 = the parity of the lane's first element: 1 = its run starts at
// the second float of the first pair): one v_cndmask per operand instead of ~10 LDS bank-conflict cycles per scalar read
// (profiles/r03c_fe_split_v1_pmc.txt).  b1 - o1, b2 - o2 must be 8-byte aligned.  a[] is read
// lane-distributed (one conflict-free ds_read_b32 per 16 steps) and broadcast inside the multiplies (v_mul_f32_dpp
// row_newbcast).
// (Measured alternative, profiles/r04p_fe_pitch_variants.log: the two chains as the halves of one v_pk_mul_f32 / v_pk_add_f32
// per step with a[] from broadcast ds_read_b128 — fewer instructions in this phase, but the kernel as a whole 3 % slower.)
template <int N>
__device__ __forceinline__ void fp_chain2_pairs(const float *a, const float *b1, bool o1, const float *b2, bool o2, int l,
                                                float &acc1, float &acc2) {
  constexpr int NF = N / 16, NP = NF / 2;
  static_assert(N % 16 == 0 && NF % 2 == 0, "N");
  typedef float fp_f2 __attribute__((ext_vector_type(2)));
  // explicit LDS address space + volatile: each pair stays ONE ds_read_b64 (2 LDS cycles); as plain loads the compiler
  // either splits them (ds_read2_b32, alignment unproven) or merges two into ds_read2_b64 (8 cycles per two pairs)
  typedef __attribute__((address_space(3))) const volatile fp_f2 fp_lds_f2;
  fp_lds_f2 *p1 = (fp_lds_f2 *)(b1 - (o1 ? 1 : 0)), *p2 = (fp_lds_f2 *)(b2 - (o2 ? 1 : 0));
  // the pairs are unpacked into scalars at once: element u of the lane's run is f[u] or f[u + 1] (ONE v_cndmask per
  // operand; left as vector lanes the compiler turns the choice into a dynamic vector index = a chain of 16 selects)
  float a0, a1, r0[18], s0[18], r1[18], s1[18];
#define FP_C2_LOAD(av, rv, sv, blk) do {                                                        \
    (av) = a[16 * (blk) + l];                                                                   \
    _Pragma("unroll") for (int u_ = 0; u_ < 9; u_++) {                                          \
      const fp_f2 t1_ = p1[8 * (blk) + u_], t2_ = p2[8 * (blk) + u_];                           \
      (rv)[2 * u_] = t1_.x; (rv)[2 * u_ + 1] = t1_.y; (sv)[2 * u_] = t2_.x; (sv)[2 * u_ + 1] = t2_.y; } \
  } while (0)
#define FP_C2_STEP(u) {                                                                         \
    const float y1_ = o1 ? rv_[(u) + 1] : rv_[u], y2_ = o2 ? sv_[(u) + 1] : sv_[u];              \
    acc1 = acc1 + fp_mul_dpp(fp_bc<u>(av_), y1_); acc2 = acc2 + fp_mul_dpp(fp_bc<u>(av_), y2_); }
#define FP_C2_MAC(av, rv, sv) do { const float av_ = (av); const float (&rv_)[18] = (rv); const float (&sv_)[18] = (sv); \
    FP_REP16(FP_C2_STEP) } while (0)
  FP_C2_LOAD(a0, r0, s0, 0);
#pragma unroll 1
  for (int p = 0; p < NP; p++) {
    const int blk = 2 * p, nx = blk + 2 < NF ? blk + 2 : NF - 1;
    FP_C2_LOAD(a1, r1, s1, blk + 1);
    FP_FENCE();
    FP_C2_MAC(a0, r0, s0);
    FP_C2_LOAD(a0, r0, s0, nx);
    FP_FENCE();
    FP_C2_MAC(a1, r1, s1);
  }
#undef FP_C2_LOAD
#undef FP_C2_STEP
#undef FP_C2_MAC
}

// ---- find_best_pitch (pitch.cpp:46-104, float instantiation) ---------------------------------------------------------------
// Group-uniform recurrence.  Everything that is not order-dependent is formed lane-parallel first with the reference's
// roundings and stays lane-distributed in registers (element 16 w + u = lane u of register w): the squares y[j]^2 of the
// initial energy, the window updates d[i] = y[i+LEN]^2 - y[i]^2 and the numerators num[i] = (xcorr[i]*1e-12)^2, with NaN
// standing for "xcorr[i] <= 0: candidate skipped" (every comparison against NaN is false).  The serial part is the
// running-energy add + clamp per candidate and, only when some candidate of a group of four beats the current second
// best (in any of the wave's streams), the cross-multiplied comparisons and selects on the (best, second best) state.
__device__ __forceinline__ float fp_syy_next(float sy, float dd) { const float t = sy + dd; return (1 > t) ? 1 : t; }
template <int n> __device__ __forceinline__ float fp_syy_next_bc(float sy, float dw) { const float t = fp_add_bc<n>(sy, dw); return (1 > t) ? 1 : t; }
#define FP_FBP_STEP(nm_, sy_, idx_) do {                                                            \
    const float num = (nm_);                                                                        \
    const bool c1 = num * bd1 > bn1 * (sy_);                                                        \
    const bool c0 = c1 && (num * bd0 > bn0 * (sy_));                                                \
    bn1 = c0 ? bn0 : (c1 ? num : bn1); bd1 = c0 ? bd0 : (c1 ? (sy_) : bd1); bp1 = c0 ? bp0 : (c1 ? (idx_) : bp1); \
    bn0 = c0 ? num : bn0; bd0 = c0 ? (sy_) : bd0; bp0 = c0 ? (idx_) : bp0;                          \
  } while (0)
// four candidates i0_ + 4 g .. + 3 (operands: lanes 4 g .. 4 g + 3 of dw_ / nw_).  The state update re-broadcasts its
// numerators from an opaque copy of nw_: were they the values of the test above, those would have to be materialised by
// four v_mov_dpp on the hot path instead of riding inside the four multiplies.
#define FP_SCAN_GROUP(g) do {                                                                       \
    const float s0_ = Syy, s1_ = fp_syy_next_bc<4 * (g)>(s0_, dw_), s2_ = fp_syy_next_bc<4 * (g) + 1>(s1_, dw_),     \
                s3_ = fp_syy_next_bc<4 * (g) + 2>(s2_, dw_);                                        \
    Syy = fp_syy_next_bc<4 * (g) + 3>(s3_, dw_);                                                    \
    const bool any_ = (fp_mul_dpp(fp_bc<4 * (g)>(nw_), bd1) > bn1 * s0_) | (fp_mul_dpp(fp_bc<4 * (g) + 1>(nw_), bd1) > bn1 * s1_) |   \
                      (fp_mul_dpp(fp_bc<4 * (g) + 2>(nw_), bd1) > bn1 * s2_) | (fp_mul_dpp(fp_bc<4 * (g) + 3>(nw_), bd1) > bn1 * s3_); \
    if (__ballot(any_)) {                                                                           \
      float nx_ = nw_;                                                                              \
      asm volatile("" : "+v"(nx_));                                                                 \
      FP_FBP_STEP(fp_bc<4 * (g)>(nx_), s0_, i0_ + 4 * (g)); FP_FBP_STEP(fp_bc<4 * (g) + 1>(nx_), s1_, i0_ + 4 * (g) + 1); \
      FP_FBP_STEP(fp_bc<4 * (g) + 2>(nx_), s2_, i0_ + 4 * (g) + 2); FP_FBP_STEP(fp_bc<4 * (g) + 3>(nx_), s3_, i0_ + 4 * (g) + 3); \
    }                                                                                               \
  } while (0);

// coarse search (pitch.cpp:315-342): 147 candidates, len 240, on the 4x-decimated signal y_lp4[j] = y[2 j] read in place
// Syy: the initial energy 1 + sum_{j<240} y_lp4[j]^2 (accumulated beside the cross-correlation that produced xcorr)
__device__ __forceinline__ void fp_coarse_best_pitch(const float *xcorr, const float *y, int l, float Syy, int &bp0_out, int &bp1_out) {
  constexpr int LEN = 240, MAXP = 147, NW = (MAXP + 15) / 16;
  float dreg[NW], nreg[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) {
    const int i = l + 16 * w, ic = i < MAXP ? i : MAXP - 1;
    const float a = y[2 * (ic + LEN)], c = y[2 * ic], xc = xcorr[ic];
    dreg[w] = a * a - c * c;
    float x16 = xc;
    x16 *= 1e-12f;
    nreg[w] = (i < MAXP && xc > 0) ? x16 * x16 : __builtin_nanf("");
  }
  FP_FENCE();                                            // every operand read and formed before the serial part starts
  FP_DPP_SETTLE("+v"(dreg[0]), "+v"(dreg[1]), "+v"(dreg[2]), "+v"(dreg[3]), "+v"(dreg[4]), "+v"(dreg[5]), "+v"(dreg[6]), "+v"(dreg[7]), "+v"(dreg[8]), "+v"(dreg[9]));
  float bn0 = -1.f, bn1 = -1.f, bd0 = 0.f, bd1 = 0.f; int bp0 = 0, bp1 = 1;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    const float dw_ = dreg[w], nw_ = nreg[w];
    const int i0_ = 16 * w;
    FP_SCAN_GROUP(0)
    if (16 * w + 4 < MAXP) { FP_SCAN_GROUP(1) }
    if (16 * w + 8 < MAXP) { FP_SCAN_GROUP(2) }
    if (16 * w + 12 < MAXP) { FP_SCAN_GROUP(3) }
  }
  bp0_out = bp0; bp1_out = bp1;
}

// fine search (pitch.cpp:344-372): only the <= 10 lags within +-2 of twice the two coarse candidates carry a correlation,
// every other entry of the reference's xcorr[] is 0 and skipped by its `if (xcorr[i] > 0)`.  The running energy of
// find_best_pitch still visits all 294 candidates, but it depends on the signal only — so it rides inside the chain that
// computes the fine correlations (fp_fine_chain_scan: its add / compare / select fill issue slots beside that chain's
// multiply-adds instead of waiting on each other in a loop of their own), lane u keeping the energy seen by candidates
// == u (mod 16) (one select per step under a loop-invariant lane mask); the 320 values go to capbuf.
//
// acc + sum_{j<480} a[j] * b[j] (b[] per lane, sixteen scalars per block; a[] lane-distributed and broadcast inside the
// multiplies, as in fp_chain2_pairs) and, beside it, Syy_{i+1} = MAX32(1, Syy_i + y[i+480]^2 - y[i]^2) for
// i < 320 (candidates >= 294 repeat the last one; their values are never read), capbuf[i] = Syy_i
__device__ __forceinline__ float fp_fine_chain_scan(const float *a, const float *b, const float *y, float *capbuf, int l, float Syy) {
  constexpr int LEN = 480, MAXP = 294, NF = LEN / 16, NS = 20;        // NS: blocks that carry scan steps (even, >= 294 / 16)
  float acc = 0.f;
  float a0, a1, b0[16], b1[16], ya0, yc0, ya1, yc1;
#define FP_FS_LOAD(av, bv, yav, ycv, blk, scan) do { (av) = a[16 * (blk) + l];                     \
    _Pragma("unroll") for (int u_ = 0; u_ < 16; u_++) (bv)[u_] = b[16 * (blk) + u_];              \
    if (scan) { const int i_ = 16 * (blk) + l, ic_ = i_ < MAXP ? i_ : MAXP - 1; (yav) = y[ic_ + LEN]; (ycv) = y[ic_]; } } while (0)
#define FP_FS_STEP(u) acc = acc + fp_mul_dpp(fp_bc<u>(av_), bv_[u]);                               \
    if (scan_) { capw_ = (l == (u)) ? Syy : capw_; Syy = fp_syy_next_bc<u>(Syy, dw_); }
#define FP_FS_MAC(av, bv, yav, ycv, blk, scan) do {                                               \
    const float av_ = (av); const float (&bv_)[16] = (bv); constexpr bool scan_ = (scan);         \
    float dw_ = scan_ ? (yav) * (yav) - (ycv) * (ycv) : 0.f; float capw_ = 0.f;                   \
    if (scan_) FP_DPP_SETTLE("+v"(dw_));                                                          \
    FP_REP16(FP_FS_STEP)                                                                          \
    if (scan_) capbuf[16 * (blk) + l] = capw_; } while (0)
  FP_FS_LOAD(a0, b0, ya0, yc0, 0, true);
#pragma unroll 1
  for (int p = 0; p < NS / 2; p++) {
    const int blk = 2 * p;
    FP_FS_LOAD(a1, b1, ya1, yc1, blk + 1, true);
    FP_FENCE();
    FP_FS_MAC(a0, b0, ya0, yc0, blk, true);
    FP_FS_LOAD(a0, b0, ya0, yc0, blk + 2, true);            // block NS's scan operands are read and not used
    FP_FENCE();
    FP_FS_MAC(a1, b1, ya1, yc1, blk + 1, true);
  }
#pragma unroll 1
  for (int p = NS / 2; p < NF / 2; p++) {
    const int blk = 2 * p, nx = blk + 2 < NF ? blk + 2 : NF - 1;
    FP_FS_LOAD(a1, b1, ya1, yc1, blk + 1, false);
    FP_FENCE();
    FP_FS_MAC(a0, b0, ya0, yc0, blk, false);
    FP_FS_LOAD(a0, b0, ya0, yc0, nx, false);
    FP_FENCE();
    FP_FS_MAC(a1, b1, ya1, yc1, blk + 1, false);
  }
#undef FP_FS_LOAD
#undef FP_FS_STEP
#undef FP_FS_MAC
  return acc;
}

// find_best_pitch of the fine search: the candidates are replayed in ascending lag order through the reference's (best,
// second best) update.  cidx/cval/cact: this lane's candidate lag, max(-1, sum) and "inside [0,294)"; lanes >= 10 are
// inactive; capbuf: the running energies left by fp_fine_chain_scan.
__device__ __forceinline__ void fp_fine_best_pitch(const float *capbuf, int l, int gb, int cidx, float cval,
                                                   bool cact, int dup_lo, int &bp0_out, int &bp1_out) {
  const float cap = capbuf[cact ? cidx : 0];            // running energy before this lane's candidate
  PN_WAVE_SYNC();
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
    FP_FBP_STEP(nm, s_r, i_r);
  }
  bp0_out = bp0; bp1_out = bp1;
}
#undef FP_SCAN_GROUP
#undef FP_FBP_STEP

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
  // (slices in the order 0, 2, 1, 3 — the two streams of a 32-lane LDS group 32 banks apart, good for 8-byte runs — was
  // measured: 0.611 vs 0.577 ms; the 4-byte reads bank mod 32 and want the 16-bank offset this order gives them)
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
#ifdef PN_FE_CLOCKS
      long long tmark_ = __builtin_readcyclecounter();
#endif
      float *pbuf = buf + FP_PBUF, *raw = buf + FP_PBUF, *scr = buf + FP_SCR, *xcs = buf + FP_SCR;
      // -- pitch_downsample (pitch.cpp:148-216) of pitch_buf == comb_buf[1632,3360): outputs 2m, 2m+1 need x[4m-1 .. 4m+3]
#if PN_FP_DS_DPP
      // x[4m - 1] is the last sample of quad m - 1, which the previous lane of the row holds (lane 0: lane 15 of the
      // previous batch of 16 quads): a row rotate inside the add instead of a second, scalar global load per quad
      float wprev = 0.f;                                   // .w of the quads of the previous batch (carried across halves)
#endif
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
        constexpr int NM = 14;                             // 2 x 14 x 16 = 448 >= 432 pairs
        float4 dv[NM]; float dm1[NM];
#pragma unroll
        for (int it = 0; it < NM; it++) {
          const int mm = l + L * (half * NM + it), m = mm < 432 ? mm : 0;
          dv[it] = *reinterpret_cast<const float4 *>(h + fe_ring(1632 + 4 * m, base_slot));
#if !PN_FP_DS_DPP
          dm1[it] = h[fe_ring(1632 + (m > 0 ? 4 * m - 1 : 0), base_slot)];
#endif
        }
#if PN_FP_DS_DPP
#pragma unroll
        for (int it = 0; it < NM; it++) {
          const float z = (l == L - 1) ? wprev : dv[it].w;   // lane 15 passes the previous batch's last sample on to lane 0
          dm1[it] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, z), 0x121, 0xf, 0xf, true));   // row_ror:1: lane i <- lane i - 1 (mod 16)
          wprev = dv[it].w;
        }
#endif
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
      FE_MARK(0);   // downsample
      // _celt_autocorr (celt_lpc.cpp:198-279): lane k holds lag k (lanes > 4 shadow lag 4)
      float ac[5];
      {
        const int lag = l < 4 ? l : 4;
#if PN_FP_ROWS_PRE
        float ack = fp_chain_rows_pre<860, 5>(raw, raw, l, 0.f);
#else
        float ack = fp_chain_rows<860>(raw, raw, l, 0.f);     // lane k <= 4: sum_i raw[i] * raw[i + k]; lanes > 4 are never read
#endif
        FE_MARK(10);  // autocorrelation chain
        float d = 0;
        for (int i = lag + 860; i < 864; i++) d = d + raw[i] * raw[i - lag];
        ack += d;
#pragma unroll
        for (int k = 0; k < 5; k++) ac[k] = __shfl(ack, gb + k);
      }
      FE_MARK(11);  // autocorrelation tail + gather
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
      FE_MARK(1);   // LPC
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

      FE_MARK(2);   // whitening FIR
      // -- pitch_search (pitch.cpp:283-386): x_lp = pbuf+384, y = pbuf, len 960, max_pitch 588 ------------------------------
      // coarse (4x decimation, 147 lags x 240 steps): x_lp4[j] = pbuf[384+2j] (group-uniform), y_lp4[j] = pbuf[2j].
      // Lane l owns the 11 consecutive lags 11 l + c.  Twelve registers hold y_lp4[11 l + e] for e = j .. j+11 (register
      // e mod 12): step j uses e = j .. j+10 and then refills the register of e = j with e = j + 12, needed two steps later.
      float SyyC = 1.0f, SyyF = 1.0f;
#if PN_FP_COARSE_PK
      // Packed form (round 5).  The 11 chains of a lane run as five v_pk_mul_f32 / v_pk_add_f32 pairs (lags c and c + 5,
      // c = 0..4) and one scalar chain (lag 10): 12 VALU instructions per step instead of 22 — each chain is still the
      // reference's sequence of one rounded multiply and one rounded add per j (the packed instructions round each half
      // like v_mul_f32 / v_add_f32; -ffp-contract=off keeps them unfused).  The sliding window holds PAIRS
      // W[e] = (y_lp4[11 l + e], y_lp4[11 l + e + 5]), e = j .. j + 11 in a ring of 12 register pairs (pair e mod 12):
      // step j multiplies W[j + c] by x_lp4[j] for c = 0..4, the scalar chain takes W[j + 5].y, and the pair of e = j is
      // then refilled with e = j + 12, first needed seven steps later.  A pair is ONE ds_read2_b32 (two dwords ten floats
      // apart).  x_lp4[j] = pbuf[384 + 2 j] is group-uniform: pair v of an operand set holds steps 2 v and 2 v + 1 (one
      // ds_read2_b32) and the multiply broadcasts the half it needs through op_sel.
      // Every LDS read of this loop is issued from inline assembly: written in C++ the compiler recognises that
      // W[e].y == W[e + 5].x, loads each element once and rebuilds the pairs with ~4 v_mov per step, and it only folds the
      // broadcast of a LOW half.  Assembly loads are invisible to the compiler's s_waitcnt pass, so the loop waits itself,
      // every third step, with lgkmcnt(4) — LDS returns in order and at most the four most recent reads may still be in
      // flight: a ring pair is requested >= 5 reads before the wait that releases it, a trip's operand set a whole trip
      // (20 reads) before — in a statement that takes what it releases as operands, which keeps every first use behind
      // the wait.  Everything is drained (lgkmcnt(0)) before any of these registers can be given to something else.
      {
        typedef float fp_v2 __attribute__((ext_vector_type(2)));
        fp_v2 accp[5]; float acc10 = 0.f;
#pragma unroll
        for (int c = 0; c < 5; c++) accp[c] = fp_v2{0.f, 0.f};
        const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)pbuf;   // LDS byte address of the slice
        const unsigned ya = pa + 8 * (FP_NCH * l);        // y_lp4[11 l] = pbuf[2 (11 l)]
        const unsigned xa0 = pa + 4 * 384;                // x_lp4[0]
        const unsigned ca = pa + 8 * l, fa = pa + 4 * l;  // the energies' lane-distributed operands: pbuf[2 (j + l)], pbuf[2 j + l]
        fp_v2 W[12], xa[6], xc[6], yf; float yc;
#define FP_PK_LOAD(u, e) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(W[u]) : "v"(yaj), "n"(2 * (e)), "n"(2 * (e) + 10))
        // operand set xs_ <- steps jn .. jn + 11, energy operands <- the squares the trip after next adds
#define FP_PK_OPERANDS(xs_, jn_) do { const unsigned xj_ = xa0 + 8 * (jn_), cj_ = ca + 8 * (jn_), fj_ = fa + 8 * (jn_); \
          asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:2" : "=&v"(xs_[0]) : "v"(xj_));      \
          asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:6" : "=&v"(xs_[1]) : "v"(xj_));      \
          asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:10" : "=&v"(xs_[2]) : "v"(xj_));     \
          asm volatile("ds_read2_b32 %0, %1 offset0:12 offset1:14" : "=&v"(xs_[3]) : "v"(xj_));    \
          asm volatile("ds_read2_b32 %0, %1 offset0:16 offset1:18" : "=&v"(xs_[4]) : "v"(xj_));    \
          asm volatile("ds_read2_b32 %0, %1 offset0:20 offset1:22" : "=&v"(xs_[5]) : "v"(xj_));    \
          asm volatile("ds_read_b32 %0, %1" : "=&v"(yc) : "v"(cj_));                               \
          asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:16" : "=&v"(yf) : "v"(fj_)); } while (0)
#define FP_PK_TIE_SET(xs_) "+v"(xs_[0]), "+v"(xs_[1]), "+v"(xs_[2]), "+v"(xs_[3]), "+v"(xs_[4]), "+v"(xs_[5])
        {
          const unsigned yaj = ya;
          FP_PK_LOAD(0, 0); FP_PK_LOAD(1, 1); FP_PK_LOAD(2, 2); FP_PK_LOAD(3, 3); FP_PK_LOAD(4, 4); FP_PK_LOAD(5, 5);
          FP_PK_LOAD(6, 6); FP_PK_LOAD(7, 7); FP_PK_LOAD(8, 8); FP_PK_LOAD(9, 9); FP_PK_LOAD(10, 10); FP_PK_LOAD(11, 11);
          FP_PK_OPERANDS(xa, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7]));
        }
        // the energies stay scalar: paired by the SLP vectoriser their DPP operands would be materialised by a v_mov each
#define FP_CO_ENERGY(u) {                                                                        \
            SyyC = fp_add_bc<u>(SyyC, qc); asm("" : "+v"(SyyC));                                  \
            SyyF = ((u) < 8 ? fp_add_bc<2 * (u)>(SyyF, qf0) : fp_add_bc<2 * (u) - 16>(SyyF, qf1)); asm("" : "+v"(SyyF)); \
            SyyF = ((u) < 8 ? fp_add_bc<2 * (u) + 1>(SyyF, qf0) : fp_add_bc<2 * (u) - 15>(SyyF, qf1)); asm("" : "+v"(SyyF)); }
#define FP_CO_WAIT(u) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(W[((u) + 5) % 12]), "+v"(W[((u) + 6) % 12]), "+v"(W[((u) + 7) % 12]))
#define FP_CO_STEP(u) {                                                                         \
            if ((u) % 3 == 0 && (u)) FP_CO_WAIT(u);       /* releases the pairs the next three steps start to use */ \
            _Pragma("unroll") for (int c = 0; c < 5; c++) {                                       \
              fp_v2 pr;                                                                          \
              if ((u) & 1) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(pr) : "v"(xq_[(u) >> 1]), "v"(W[((u) + c) % 12])); \
              else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(pr) : "v"(xq_[(u) >> 1]), "v"(W[((u) + c) % 12])); \
              accp[c] = accp[c] + pr; }                                                          \
            { float p10 = (((u) & 1) ? xq_[(u) >> 1].y : xq_[(u) >> 1].x) * W[((u) + 5) % 12].y; asm("" : "+v"(p10)); acc10 = acc10 + p10; } \
            FP_PK_LOAD(u, (u) + 12);                                                             \
            FP_CO_ENERGY(u) }
        // 12 steps j0_ .. j0_ + 11 on operand set xcur_ while the operands of the trip after (steps jn_ ..) arrive in xn_.
        // The trip's first wait also releases its own operand set and energy operands (requested one trip ago).
#define FP_CO_TRIP(xcur_, xn_, j0_, jn_) {                                                        \
          fp_v2 (&xq_)[6] = xcur_;                                                                  \
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(W[5]), "+v"(W[6]), "+v"(W[7]), FP_PK_TIE_SET(xcur_), "+v"(yc), "+v"(yf)); \
          float qc = yc * yc, qf0 = yf.x * yf.x, qf1 = yf.y * yf.y;                                 \
          FP_DPP_SETTLE("+v"(qc), "+v"(qf0), "+v"(qf1));                                            \
          FP_PK_OPERANDS(xn_, jn_);                                                                 \
          const unsigned yaj = ya + 8 * (j0_);                 /* byte address of y_lp4[11 l + j0] */ \
          FP_REP12(FP_CO_STEP) }
#pragma unroll 1
        for (int j0 = 0; j0 < 240; j0 += 24) {
          FP_CO_TRIP(xa, xc, j0, j0 + 12)
          const int jn = j0 + 24 < 240 ? j0 + 24 : j0;       // the last trip re-reads a block it does not use
          FP_CO_TRIP(xc, xa, j0 + 12, jn)
        }
#undef FP_CO_TRIP
#undef FP_CO_STEP
#undef FP_CO_WAIT
#undef FP_CO_ENERGY
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]),
                     "+v"(W[6]), "+v"(W[7]), "+v"(W[8]), "+v"(W[9]), "+v"(W[10]), "+v"(W[11]), FP_PK_TIE_SET(xa), FP_PK_TIE_SET(xc), "+v"(yc), "+v"(yf));
#undef FP_PK_LOAD
#undef FP_PK_OPERANDS
#undef FP_PK_TIE_SET
#pragma unroll
        for (int c = 0; c < 5; c++) { xcs[FP_NCH * l + c] = accp[c].x; xcs[FP_NCH * l + c + 5] = accp[c].y; }
        xcs[FP_NCH * l + 10] = acc10;
      }
#else
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
        // Beside the 11 multiply-adds of a step ride the two initial energies of find_best_pitch (pitch.cpp:62-63), which
        // depend on the signal only: SyyC = 1 + sum_{j<240} y_lp4[j]^2 (one term per step) and SyyF = 1 + sum_{j<480} y[j]^2
        // (two terms per step), their squares lane-distributed (yc: elements j0 + lane; yf0 / yf1: 2 j0 + lane, + 16).
        float yc = pbuf[2 * l], yf0 = pbuf[l], yf1 = pbuf[16 + l];
#pragma unroll 1
        for (int j0 = 0; j0 < 240; j0 += 12) {
          float4 xn[6];
          const int jn = j0 + 12 < 240 ? j0 + 12 : j0;       // the last block re-reads itself
          const float qc = yc * yc, qf0 = yf0 * yf0, qf1 = yf1 * yf1;
#pragma unroll
          for (int v = 0; v < 6; v++) xn[v] = *reinterpret_cast<const float4 *>(xb + 2 * jn + 4 * v);
          yc = pbuf[2 * (jn + l)]; yf0 = pbuf[2 * jn + l]; yf1 = pbuf[2 * jn + 16 + l];
#define FP_CO_STEP(u) {                                                                         \
            const float xj = ((u) & 1) ? xq[(u) >> 1].z : xq[(u) >> 1].x;      /* pbuf[384 + 2 (j0 + u)] */ \
            _Pragma("unroll") for (int c = 0; c < FP_NCH; c++) acc[c] = acc[c] + xj * R[((u) + c) % 12]; \
            R[u] = yb[2 * (j0 + (u) + 12)];                                                      \
            SyyC = SyyC + fp_bc<u>(qc);                                                          \
            SyyF = SyyF + ((u) < 8 ? fp_bc<2 * (u)>(qf0) : fp_bc<2 * (u) - 16>(qf1));              \
            SyyF = SyyF + ((u) < 8 ? fp_bc<2 * (u) + 1>(qf0) : fp_bc<2 * (u) - 15>(qf1)); }
          FP_REP12(FP_CO_STEP)
#undef FP_CO_STEP
#pragma unroll
          for (int v = 0; v < 6; v++) xq[v] = xn[v];
        }
#pragma unroll
        for (int c = 0; c < FP_NCH; c++) xcs[FP_NCH * l + c] = acc[c];
      }
#endif
      PN_WAVE_SYNC();
      FE_MARK(3);   // coarse cross-correlation
      int bp0, bp1;
      fp_coarse_best_pitch(xcs, pbuf, l, SyyC, bp0, bp1);
      PN_WAVE_SYNC();
      FE_MARK(4);   // coarse best-pitch scan
      // fine (2x decimation): only lags within +-2 of 2*best (pitch.cpp:344-361); every other xcorr entry is 0
      int cidx; float cval; bool cact;
      const int dup_lo = 2 * bp0 - 2;
      {
        cidx = (l < 5) ? (2 * bp0 - 2 + l) : (2 * bp1 - 2 + (l - 5));
        cact = l < 10 && cidx >= 0 && cidx < 294;
        const float sum = fp_fine_chain_scan(pbuf + 384, pbuf + (cact ? cidx : 0), pbuf, scr, l, SyyF);
        cval = (-1 > sum) ? -1 : sum;
      }
      FE_MARK(5);   // fine cross-correlation
      fp_fine_best_pitch(scr, l, gb, cidx, cval, cact, dup_lo, bp0, bp1);
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

      FE_MARK(6);   // fine best-pitch scan + interpolation
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
        fp_chain2_pairs<480>(x, x - lag1, (lag1 & 1) != 0, x - lag2, (lag2 & 1) != 0, l, dot1, dot2);
        FE_MARK(7);   // remove_doubling: 29 inner products
        const float xx = __shfl(dot1, gb);
        float xy = __shfl(dot1, gb + 1);
        // yy_lookup (pitch.cpp:449-455): strictly sequential running energy, group-uniform: yy += x[-i]^2 - x[N-i]^2.  The
        // squares are formed lane-parallel, 64 at a time, and stay lane-distributed in registers (fp_bc); the clamp
        // MAX32(0, yy) of the table does not feed back into the recurrence, so the chain is an add and a subtract per
        // step; lane u keeps the steps == u (mod 16), the 384 raw values go to LDS once and every lane reads its two.
        float y1, y2;
        {
          float *ybuf = scr;                     // the unclamped running energy of i = 1 .. 384 at [i - 1]
          float yy = xx;
          float an[4], cn[4];
#define FP_YY_LOAD(blk) do {                                                                      \
    _Pragma("unroll") for (int w = 0; w < 4; w++) { const int i = 1 + 64 * (blk) + l + L * w; an[w] = x[-i]; cn[w] = x[480 - i]; } \
  } while (0)
          FP_YY_LOAD(0);
#pragma unroll 1
          for (int blk = 0; blk < 6; blk++) {        // i = 1 + 64 blk + 16 w + u  (384 = 6 * 64)
            float pa[4], qa[4];
#pragma unroll
            for (int w = 0; w < 4; w++) { pa[w] = an[w] * an[w]; qa[w] = cn[w] * cn[w]; }
            FP_YY_LOAD(blk < 5 ? blk + 1 : blk);       // next block's operands arrive under this block's chain
            FP_FENCE();
            FP_DPP_SETTLE("+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]), "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]));
#pragma unroll
            for (int w = 0; w < 4; w++) {
              const float pw_ = pa[w], qw_ = qa[w];
              float cw = 0.f;
#if PN_FP_DPP_ASM >= 2
              // the 16 steps of a block as ONE assembly statement (48 instructions, no padding between them: around single-
              // instruction statements the compiler still leaves a wait state for an unknown producer); lane u of every
              // row captures step u through a v_cndmask under a loop-invariant lane mask
#define FP_YY_ASM(u, m) "v_add_f32_dpp %0, %2, %0 row_newbcast:" #u " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"        \
                        "v_subrev_f32_dpp %0, %3, %0 row_newbcast:" #u " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"     \
                        "v_cndmask_b32_e64 %1, %1, %0, %" #m "\n\t"
#define FP_LM(u) (0x0001000100010001ull << (u))    /* lane u of each of the four rows */
              asm(FP_YY_ASM(0, 4) FP_YY_ASM(1, 5) FP_YY_ASM(2, 6) FP_YY_ASM(3, 7) FP_YY_ASM(4, 8) FP_YY_ASM(5, 9) FP_YY_ASM(6, 10) FP_YY_ASM(7, 11)
                  FP_YY_ASM(8, 12) FP_YY_ASM(9, 13) FP_YY_ASM(10, 14) FP_YY_ASM(11, 15) FP_YY_ASM(12, 16) FP_YY_ASM(13, 17) FP_YY_ASM(14, 18) FP_YY_ASM(15, 19)
                  : "+v"(yy), "+v"(cw) : "v"(pw_), "v"(qw_), "s"(FP_LM(0)), "s"(FP_LM(1)), "s"(FP_LM(2)), "s"(FP_LM(3)), "s"(FP_LM(4)), "s"(FP_LM(5)),
                    "s"(FP_LM(6)), "s"(FP_LM(7)), "s"(FP_LM(8)), "s"(FP_LM(9)), "s"(FP_LM(10)), "s"(FP_LM(11)), "s"(FP_LM(12)), "s"(FP_LM(13)),
                    "s"(FP_LM(14)), "s"(FP_LM(15)));
#undef FP_YY_ASM
#undef FP_LM
#else
#define FP_YY_STEP(u) yy = fp_add_bc<u>(yy, pw_); yy = fp_sub_bc<u>(yy, qw_); cw = (l == (u)) ? yy : cw;
              FP_REP16(FP_YY_STEP)
#undef FP_YY_STEP
#endif
              ybuf[64 * blk + L * w + l] = cw;
            }
          }
#undef FP_YY_LOAD
          PN_WAVE_SYNC();
          // yy_lookup[i] = MAX32(0, yy) for i >= 1, yy_lookup[0] = xx; lane 1 looks up T0 twice
          const float r1 = ybuf[lag1 > 0 ? lag1 - 1 : 0], r2 = ybuf[lag2 > 0 ? lag2 - 1 : 0];
          y1 = lag1 > 0 ? ((0 > r1) ? 0 : r1) : xx;
          y2 = lag2 > 0 ? ((0 > r2) ? 0 : r2) : xx;
        }
        PN_WAVE_SYNC();
        FE_MARK(8);   // remove_doubling: yy_lookup
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
#if PN_FP_ROWS_PRE
        const float xc = fp_chain_rows_pre<480, 3>(x, x - (Tsel + 1), l, 0.f);
#else
        const float xc = fp_chain_rows<480>(x, x - (Tsel + 1), l, 0.f);
#endif
        FE_MARK(12);  // remove_doubling: decision + the three final inner products
        const float xc0 = __shfl(xc, gb + 2), xc1 = __shfl(xc, gb + 1), xc2 = __shfl(xc, gb);
        int off2;
        if ((xc2 - xc0) > .7f * (xc1 - xc0)) off2 = 1;
        else if ((xc0 - xc2) > .7f * (xc1 - xc2)) off2 = -1;
        else off2 = 0;
        if (pg > g) pg = g;
        pitch_index = 2 * Tsel + off2;
        if (pitch_index < PN_PITCH_MIN) pitch_index = PN_PITCH_MIN;
      }
      FE_MARK(9);   // remove_doubling: refinement + outputs
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
