// Shadow-operand network kernels for gfx950: the GEMMs of the gain network on the fp16 matrix cores, operands taken
// from fragment-order fp16 shadows of the activations, fp32 accumulation and fp32 state.  Two instantiations:
//   NP = 2  split precision (nn_mode PN_NN_MFMA_X3): fp32 GEMMs with error compensation, described below;
//   NP = 1  fp16 operands (nn_mode PN_NN_MFMA_F16, BASELINE configs[4]): the hi plane only, one MFMA per operand pair.
//
// Every GEMM operand x (activation or weight, fp32) is carried as two fp16 numbers, hi = fp16(x) and
// lo = fp16(x - hi) (x - hi is exact in fp32; v_mfma_f32_32x32x16_f16 honours fp16 subnormals on gfx950 —
// tools/probes/mfma_f16_denorm_probe.hip — so lo keeps an absolute precision of 2^-25 for |x| <= 1 and a relative
// precision of 2^-22 above), and every product a*b is formed as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  by three MFMAs into
// the same fp32 accumulator; the dropped a_lo*b_lo term is below 2^-22 |a b|.  The operand error of a length-1024 dot
// product is then ~5x SMALLER than the rounding error the reference's own sequential fp32 accumulation makes
// (measured against float64 arithmetic: tools/x3_vs_f64.py — max |g,r - exact| 2.0e-6 for this mode, 2.6e-6 for the CPU
// reference itself), so the deviation of this mode from the CPU path is, like PN_NN_MFMA's, the summation
// order — measured against the same bounds (+-1 LSB PCM, 2e-5 on g/r; tests/test_gpu_x3.py).  The fp16 matrix
// cores run 16x the fp32 MFMA rate: three products still leave 5.3x, which moves these GEMMs from MFMA-bound to
// LDS/L2-bound.  Bias preload, table tanh/sigmoid, gating, the state blend and every stored state value stay fp32.
//
// Tiling (different from pn_nn.hip because the limiter is different):
//   block = 4 waves x 64 rows = 256 streams (RG = 2; RG = 1: 32 rows per wave, 128-row blocks, for batches below 32 768
//   streams), NT column tiles of 32 (the three gate tiles of one n-tile for a GRU);
//   A (activations) is NOT staged through LDS: a wave's rows are private to it, so its MFMA fragments are loaded
//   straight from global memory into registers from a FRAGMENT-ORDER shadow of the producing layer's output,
//     shadow[M tile of 128][column tile of 32][plane hi|lo][k-group of 8][row 0..127][8 halfs]        (8 KB per plane)
//   — lane (row r, k-half kh) of k-step s reads the 16 bytes (k-group 2s+kh, row r): 512 contiguous bytes per 32 lanes;
//   the producer block writes its 256 x 32 output tile as whole 8 KB runs.  Four k-steps of A stay in flight per wave.
//   B (weights, packed per (column tile, 32 k) as [k-step][plane][lane][8 halfs] = 4 KB) is copied linearly into a
//   double-buffered LDS image through registers and read back with conflict-free ds_read_b128 (lane-linear).
//   Per wave and k-step of 16: 4 global loads, 2*NT ds_read_b128, 6*NT MFMAs of 32 cycles -> LDS reads take ~1/3 of
//   the MFMA time on a CU (with 32-row wave tiles they would take 2/3, with A through LDS more than all of it).
#include "pn_nn_common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float fvec4 __attribute__((ext_vector_type(4)));

#define XM 256                         // rows per block
#define X3_PLANE 512                   // uint4 per (M tile of 128, column tile, plane): 4 k-groups x 128 rows

struct X3Shared {
  uint4 B[2][4][4][64];                // [buffer][column tile][2*kstep + plane][lane]: 2 x 16 KB
  float tansig[208];
};
#define X3_TLD 36                      // epilogue stage: 32 rows x 32 columns per wave, rows padded to 36 floats
static_assert(4 * 32 * X3_TLD * sizeof(float) <= sizeof(uint4) * 2 * 4 * 4 * 64, "stage aliases the weight buffers");

template <int RG> struct X3A { fvec4 h[RG], l[RG]; };      // one k-step of A fragments: [row group of 32] hi / lo

__device__ __forceinline__ half8 x3_h8(const fvec4 &v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ half8 x3_h8(const uint4 &v) { return __builtin_bit_cast(half8, v); }

// Debug counter of operand saturation (PERCEPNET_X3_SATCOUNT=1 at context creation; per device, cumulative): activations
// beyond the fp16 range are CLAMPED to +-65504 when they become GEMM operands of the fp16-operand / split-precision modes
// (weights beyond it are refused at context creation).  conv1 is a ReLU and unbounded, so a model can get there; the
// counter makes it visible (pn_ctx_describe: x3_saturated=<values clamped so far>) instead of silent.
__device__ int pn_x3_sat_enable = 0;
__device__ unsigned long long pn_x3_sat_count = 0;
int pn_x3_sat_set(int enable) {
  const unsigned long long zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(pn_x3_sat_count), &zero, sizeof(zero)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(pn_x3_sat_enable), &enable, sizeof(enable)) == hipSuccess ? 0 : -1;
}
long long pn_x3_sat_read() {
  unsigned long long n = 0;
  return hipMemcpyFromSymbol(&n, HIP_SYMBOL(pn_x3_sat_count), sizeof(n)) == hipSuccess ? (long long)n : -1;
}

// hi/lo planes of eight consecutive fp32 values (operands beyond the fp16 range saturate instead of turning into NaNs)
__device__ __forceinline__ void x3_split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
  half8 h, l;
  if (pn_x3_sat_enable) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) n += (v[j] > 65504.f || v[j] < -65504.f) ? 1 : 0;
    if (n) atomicAdd(&pn_x3_sat_count, (unsigned long long)n);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float c = __builtin_fminf(__builtin_fmaxf(v[j], -65504.f), 65504.f);
    const _Float16 hh = (_Float16)c;
    h[j] = hh;
    l[j] = (_Float16)(c - (float)hh);
  }
  hi = __builtin_bit_cast(uint4, h); lo = __builtin_bit_cast(uint4, l);
}

// A fragments of k-step s (0/1) of the 32-k tile at pt (= plane-0 pointer of the lane; plane 1 is X3_PLANE further)
template <int RG, int NP>
__device__ __forceinline__ void x3_load_A(X3A<RG> &q, const uint4 *__restrict__ pt, int s) {
#pragma unroll
  for (int rg = 0; rg < RG; rg++) q.h[rg] = *reinterpret_cast<const fvec4 *>(pt + s * 256 + 32 * rg);
  if constexpr (NP == 2) {
#pragma unroll
    for (int rg = 0; rg < RG; rg++) q.l[rg] = *reinterpret_cast<const fvec4 *>(pt + X3_PLANE + s * 256 + 32 * rg);
  }
}

// one 32-k tile (two k-steps): acc[rg][IDX[t]] += A * B[t] for both row groups, three products each (small terms
// first).  The B fragments of group (k-step, column tile) i+1 are read while the six MFMAs of group i run; the A
// registers of a k-step are refilled (tile + 2) as soon as its last MFMA has issued — PF = the lane's pointer to that tile.
struct X3B { fvec4 h, l; };
// the j-th 4-register group of the hi planes of four A-fragment sets (the paired-phase kernel parks other values there)
__device__ __forceinline__ fvec4 &x3_q4(X3A<2> &q0, X3A<2> &q1, X3A<2> &q2, X3A<2> &q3, int j) {
  X3A<2> &q = (j >> 1) == 0 ? q0 : ((j >> 1) == 1 ? q1 : ((j >> 1) == 2 ? q2 : q3));
  return q.h[j & 1];
}
template <int NP>
__device__ __forceinline__ X3B x3_read_B(const uint4 (*Bs)[4][64], int t, int s, int lane) {
  X3B f;
  f.h = *reinterpret_cast<const fvec4 *>(&Bs[t][NP * s][lane]);
  if constexpr (NP == 2) f.l = *reinterpret_cast<const fvec4 *>(&Bs[t][2 * s + 1][lane]);
  else f.l = f.h;
  return f;
}
template <int RG, int NP>
__device__ __forceinline__ void x3_mma6(const X3A<RG> &q, const X3B &f, floatx16 (&acc)[RG][4], int idx) {
  const half8 bh = x3_h8(f.h);
  if constexpr (NP == 2) {
    const half8 bl = x3_h8(f.l);
#pragma unroll
    for (int rg = 0; rg < RG; rg++) acc[rg][idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_h8(q.l[rg]), bh, acc[rg][idx], 0, 0, 0);
#pragma unroll
    for (int rg = 0; rg < RG; rg++) acc[rg][idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_h8(q.h[rg]), bl, acc[rg][idx], 0, 0, 0);
  }
#pragma unroll
  for (int rg = 0; rg < RG; rg++) acc[rg][idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x3_h8(q.h[rg]), bh, acc[rg][idx], 0, 0, 0);
}
template <int RG, int NP, int NT, int I0, int I1, int I2, int I3>
__device__ __forceinline__ void x3_tile(X3A<RG> &qa, X3A<RG> &qb, const uint4 (*Bs)[4][64], const uint4 *__restrict__ pf, int lane,
                                        floatx16 (&acc)[RG][4]) {
  constexpr int IDX[4] = {I0, I1, I2, I3};
  X3B f0 = x3_read_B<NP>(Bs, 0, 0, lane), f1;
#pragma unroll
  for (int i = 0; i < 2 * NT; i++) {
    const int s = i / NT, t = i % NT;
    X3B &cur = (i & 1) ? f1 : f0, &nxt = (i & 1) ? f0 : f1;
    if (i + 1 < 2 * NT) nxt = x3_read_B<NP>(Bs, (i + 1) % NT, (i + 1) / NT, lane);
    __builtin_amdgcn_sched_barrier(0);                   // keep the read ahead of the MFMAs it overlaps (the scheduler sinks it)
    x3_mma6<RG, NP>(s ? qb : qa, cur, acc, IDX[t]);
    __builtin_amdgcn_sched_barrier(0);
#if !(defined(PN_XP_ABL) && (PN_XP_ABL & 4))   // timing ablation (results wrong): no A refills
    if (t == NT - 1) {
      x3_load_A<RG, NP>(s ? qb : qa, pf, s);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  }
}

// x3_tile for a wave that is the ONLY matrix-pipe user of its SIMD (paired-phase kernel): nothing fills the bubbles of its
// instruction stream, so (a) the B fragments are read D-1 groups ahead (a group is only 2 MFMAs = 64 cycles in the fp16-
// operand instantiation, less than an LDS round trip), (b) the read-ahead runs across the tile boundary into the NEXT tile's
// LDS image Bn (complete since the previous barrier: the weight tiles live in a ring of three), so a tile opens with its
// first fragments already in registers, fr[0 .. D-2], and (c) the tile's other work — the weight-tile stash and load (mid1,
// after group 1) and the next A pointer (mid3, after group 3) — is issued between MFMA groups instead of around the tile.
template <int RG, int NP, int NT, int D, int I0, int I1, int I2, int I3, class M1, class M3>
__device__ __forceinline__ void x3_tile_r(X3A<RG> &qa, X3A<RG> &qb, const uint4 (*Bc)[4][64], const uint4 (*Bn)[4][64],
                                          const uint4 *__restrict__ pf, int lane, floatx16 (&acc)[RG][4], X3B (&fr)[D],
                                          M1 &&mid1, M3 &&mid3) {
  constexpr int IDX[4] = {I0, I1, I2, I3};
  constexpr int G = 2 * NT;
  static_assert(G % D == 0 && D >= 2, "the fragment window must tile the groups of a k-tile");
#pragma unroll
  for (int i = 0; i < G; i++) {
    const int s = i / NT, t = i % NT;
    const int j = i + D - 1;                             // the group whose fragments are fetched now
    if (j < G) fr[j % D] = x3_read_B<NP>(Bc, j % NT, j / NT, lane);
    else fr[j % D] = x3_read_B<NP>(Bn, (j - G) % NT, (j - G) / NT, lane);
    __builtin_amdgcn_sched_barrier(0);
    x3_mma6<RG, NP>(s ? qb : qa, fr[i % D], acc, IDX[t]);
    __builtin_amdgcn_sched_barrier(0);
    if (t == NT - 1) {
      x3_load_A<RG, NP>(s ? qb : qa, pf, s);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (i == 1) { mid1(); __builtin_amdgcn_sched_barrier(0); }
    if (i == 3) { mid3(); __builtin_amdgcn_sched_barrier(0); }
  }
}

// Stores one 32 x 32 output tile of a wave (rows row0.., v[i] = value of row (i&3) + 8(i>>2) + 4(lane>>5), column
// lane&31) through the wave's private LDS stage: fp32 rows as 16-byte stores (out may be null) and the hi/lo shadow
// planes as one 16-byte store per (row, k-group) (S may be null; Srow = row within the M tile of 128).  Two halves, so
// that the paired-phase kernel can spread them over its epilogue steps: x3_stage_write puts the tile into the stage,
// x3_stage_store<NP>(p = 0 | 1) stores rows 0..15 / 16..31 of it.
__device__ __forceinline__ void x3_stage_write(float *T, const float (&v)[16], int lane) {
#pragma unroll
  for (int i = 0; i < 16; i++) T[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * X3_TLD + (lane & 31)] = v[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int NP>
__device__ __forceinline__ void x3_stage_store(const float *T, int p, float *__restrict__ out, int ldo, int col0, int n_cols,
                                               int grow0, int n_rows, uint4 *__restrict__ S, int srow0, int lane) {
  const int idx = lane + 64 * p, row = idx >> 2, kg = idx & 3;
  float f[8];
  *reinterpret_cast<float4 *>(&f[0]) = *reinterpret_cast<const float4 *>(&T[row * X3_TLD + 8 * kg]);
  *reinterpret_cast<float4 *>(&f[4]) = *reinterpret_cast<const float4 *>(&T[row * X3_TLD + 8 * kg + 4]);
  if (out && grow0 + row < n_rows) {
    float *o = out + (size_t)(grow0 + row) * ldo + col0 + 8 * kg;
    if (col0 + 32 <= n_cols && (ldo & 3) == 0) {
      *reinterpret_cast<float4 *>(o) = *reinterpret_cast<const float4 *>(&f[0]);
      *reinterpret_cast<float4 *>(o + 4) = *reinterpret_cast<const float4 *>(&f[4]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) if (col0 + 8 * kg + j < n_cols) o[j] = f[j];
    }
  }
  if (S) {
    uint4 hi, lo;
    x3_split8(f, hi, lo);
    S[kg * 128 + srow0 + row] = hi;
    if constexpr (NP == 2) S[X3_PLANE + kg * 128 + srow0 + row] = lo;
  }
}
template <int NP>
__device__ __forceinline__ void x3_store_tile(float *T, const float (&v)[16], float *__restrict__ out, int ldo, int col0,
                                              int n_cols, int grow0, int n_rows, uint4 *__restrict__ S, int srow0, int lane) {
  x3_stage_write(T, v, lane);
#pragma unroll
  for (int p = 0; p < 2; p++) x3_stage_store<NP>(T, p, out, ldo, col0, n_cols, grow0, n_rows, S, srow0, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                      // the stage is reused by the next tile of this wave
}

// one packed weight tile (NP x 2 KB) is copied linearly into its LDS slot: 16 bytes per thread (NP = 2) or 8 (NP = 1)
typedef float fvec2 __attribute__((ext_vector_type(2)));
template <int NP> struct X3BStage { typedef fvec4 T; };
template <> struct X3BStage<1> { typedef fvec2 T; };
#define X3_BVEC typename X3BStage<NP>::T
#define X3_BLOAD(dst, src) (dst) = reinterpret_cast<const X3_BVEC *>(src)[tid]
#define X3_BSTASH(buf, t, v) reinterpret_cast<X3_BVEC *>(&S.B[buf][t][0][0])[tid] = (v)
#define X3_BTILE (128 * NP)            // uint4 per packed weight tile

// ---- dense / conv layer: out = act(bias + A W), NT column tiles per block ----------------------------------------
// A: shadow panels (PnSegs pointers carry uint4* shadows; width = logical columns of each panel, all equal)
template <int RG, int NP, int NT>
__global__ __launch_bounds__(NN_THREADS, 4 - RG) void pn_dense_x3_kernel(
    PnSegs A, const uint4 *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, uint4 *__restrict__ outS, int nts_out,
    int n_rows, int n_mtiles, int n_cblocks) {
  __shared__ X3Shared S;
  int mt, cb;
  if (!pn_tile_of_block(n_mtiles, n_cblocks, mt, cb)) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // the activation table (201 entries): one value per thread, loaded together with the first operand tiles and written to
  // LDS before the first barrier — staged up front it cost a full memory latency before any operand load was issued
  const float ts_v = tansig[tid < 201 ? tid : 200];
  floatx16 acc[RG][4];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int col = (cb * NT + t) * 32 + (lane & 31);
    const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
    for (int rg = 0; rg < RG; rg++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[rg][t][i] = bv;
  }
  constexpr int XMB = 128 * RG;                          // rows per block
  const int mt128 = (mt * XMB + 32 * RG * wave) >> 7, srow = (32 * RG * wave) & 127;
  const int lane_off = (lane >> 5) * 128 + srow + (lane & 31);
  const uint4 *wbase = Wp + (size_t)(cb * NT) * KT * X3_BTILE;
  PN_PANEL_LOCALS(A);
  (void)pld;
  // A tiles are asked for strictly in order (0, 1, 2, ...): a cursor (panel, tile within panel) instead of a division
  // per tile; past the last tile it keeps returning the last one (loaded, never used)
  int c_sg = 0, c_kt = 0, c_g = 0;
  const uint4 *c_last = nullptr;
#define XD_APTR(gg, pt) \
    const uint4 *pt; { if (c_g < KT) { \
        c_last = reinterpret_cast<const uint4 *>(pn_seg_ptr(PN_PANEL_PASS, c_sg)) + ((size_t)mt128 * tps + c_kt) * (NP * X3_PLANE) + lane_off; \
        c_g++; c_kt++; if (c_kt == tps) { c_kt = 0; c_sg++; } } \
      pt = c_last; }
#define XD_BLOAD(gg) do { int g_ = (gg); g_ = g_ < KT ? g_ : KT - 1; \
    _Pragma("unroll") for (int t = 0; t < NT; t++) X3_BLOAD(rb[t], wbase + ((size_t)t * KT + g_) * X3_BTILE); } while (0)
#define XD_BSTASH(buf) do { _Pragma("unroll") for (int t = 0; t < NT; t++) X3_BSTASH(buf, t, rb[t]); } while (0)
  X3A<RG> q0, q1, q2, q3;
  X3_BVEC rb[NT];
  { XD_APTR(0, p0); x3_load_A<RG, NP>(q0, p0, 0); x3_load_A<RG, NP>(q1, p0, 1); }
  { XD_APTR(1, p1); x3_load_A<RG, NP>(q2, p1, 0); x3_load_A<RG, NP>(q3, p1, 1); }
  XD_BLOAD(0); XD_BSTASH(0); XD_BLOAD(1);
  if (tid < 201) S.tansig[tid] = ts_v;
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < KT; g += 2) {
    { XD_APTR(g + 2, pa); x3_tile<RG, NP, NT, 0, 1, 2, 3>(q0, q1, S.B[0], pa, lane, acc); }
    XD_BSTASH(1); XD_BLOAD(g + 2);
    __syncthreads();
    { XD_APTR(g + 3, pb); x3_tile<RG, NP, NT, 0, 1, 2, 3>(q2, q3, S.B[1], pb, lane, acc); }
    XD_BSTASH(0); XD_BLOAD(g + 3);
    __syncthreads();
  }
#undef XD_APTR
#undef XD_BLOAD
#undef XD_BSTASH
  float *T = reinterpret_cast<float *>(&S.B[0][0][0][0]) + wave * 32 * X3_TLD;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int ct = cb * NT + t, col0 = ct * 32;
    if (col0 >= N) break;
#pragma unroll
    for (int rg = 0; rg < RG; rg++) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = pn_act(acc[rg][t][i], act, S.tansig);
      uint4 *Sx = outS ? outS + ((size_t)mt128 * nts_out + ct) * (NP * X3_PLANE) : nullptr;
      x3_store_tile<NP>(T, v, out, ldo, col0, N, mt * XMB + 32 * RG * wave + 32 * rg, n_rows, Sx, srow + 32 * rg, lane);
    }
  }
}

// Tuning aid (-DPN_X3_CLOCKS): shader-clock stamps of wave 0 of every block of the last N=512 launch
#ifdef PN_X3_CLOCKS
__device__ unsigned long long pn_x3_trace[4096 * 8];
extern "C" PN_EXPORT int pn_x3_trace_read(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pn_x3_trace), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#define X3_STAMP(i) do { if (N == 512) ck_[i] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif

// ---- GRU step (reset-after, nnet.cpp:122-180): acc z, r, hx (W_h x), tmp (U_h h) ---------------------------------
template <int RG, int NP>
__global__ __launch_bounds__(NN_THREADS, 4 - RG) void pn_gru_x3_kernel(
    PnSegs X, const float *__restrict__ h_old, const uint4 *__restrict__ h_oldS, const uint4 *__restrict__ Wp,
    const uint4 *__restrict__ Up, const float *__restrict__ b, int N, int KTx, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ h_new, uint4 *__restrict__ h_newS, int n_rows, int n_mtiles) {
  __shared__ X3Shared S;
  const int NTn = N >> 5;
  int mt, nt;
  if (!pn_tile_of_block(n_mtiles, NTn, mt, nt)) return;
#ifdef PN_X3_CLOCKS
  unsigned long long ck_[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long w0_ = wall_clock64();
  X3_STAMP(0);
#endif
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int KTh = NTn, T1 = KTx, TT = KTx + KTh;
  const int col = nt * 32 + (lane & 31);
  // the activation table (201 entries): one value per thread, loaded together with the first operand tiles and written to
  // LDS before the first barrier — staged up front it cost a full memory latency before any operand load was issued
  const float ts_v = tansig[tid < 201 ? tid : 200];
  floatx16 acc[RG][4];
  {
    float bz = b[col]; bz += b[3 * N + col];
    float br = b[N + col]; br += b[4 * N + col];
    const float bt = b[5 * N + col];
#pragma unroll
    for (int rg = 0; rg < RG; rg++)
#pragma unroll
      for (int i = 0; i < 16; i++) { acc[rg][0][i] = bz; acc[rg][1][i] = br; acc[rg][2][i] = 0.f; acc[rg][3][i] = bt; }
  }
  constexpr int XMB = 128 * RG;                          // rows per block
  const int mt128 = (mt * XMB + 32 * RG * wave) >> 7, srow = (32 * RG * wave) & 127;
  const int lane_off = (lane >> 5) * 128 + srow + (lane & 31);
  const uint4 *Wz = Wp + (size_t)(0 * NTn + nt) * KTx * X3_BTILE, *Wr = Wp + (size_t)(1 * NTn + nt) * KTx * X3_BTILE,
              *Wh = Wp + (size_t)(2 * NTn + nt) * KTx * X3_BTILE;
  const uint4 *Uz = Up + (size_t)(0 * NTn + nt) * KTh * X3_BTILE, *Ur = Up + (size_t)(1 * NTn + nt) * KTh * X3_BTILE,
              *Uh = Up + (size_t)(2 * NTn + nt) * KTh * X3_BTILE;
  PN_PANEL_LOCALS(X);
  (void)pld;
#define XG_SEL(gg) int g_ = (gg); g_ = g_ < TT ? g_ : TT - 1; const bool p1_ = g_ < T1; \
    const int kx_ = p1_ ? g_ : 0, kh_ = p1_ ? 0 : g_ - T1
  // A tiles are asked for strictly in order: a cursor over (x panels, then the recurrent operand) instead of a division
  // per tile; past the last tile it keeps returning the last one (loaded, never used)
  int c_sg = 0, c_kt = 0, c_g = 0;
  const uint4 *c_last = nullptr;
#define XG_APTR(gg, pt) \
    const uint4 *pt; { if (c_g < T1) { \
        c_last = reinterpret_cast<const uint4 *>(pn_seg_ptr(PN_PANEL_PASS, c_sg)) + ((size_t)mt128 * tps + c_kt) * (NP * X3_PLANE) + lane_off; \
        c_kt++; if (c_kt == tps) { c_kt = 0; c_sg++; } \
      } else if (c_g < TT) { c_last = h_oldS + ((size_t)mt128 * NTn + (c_g - T1)) * (NP * X3_PLANE) + lane_off; } \
      c_g++; pt = c_last; }
#define XG_BLOAD(gg) do { XG_SEL(gg); \
    X3_BLOAD(rb[0], p1_ ? Wz + (size_t)kx_ * X3_BTILE : Uz + (size_t)kh_ * X3_BTILE); \
    X3_BLOAD(rb[1], p1_ ? Wr + (size_t)kx_ * X3_BTILE : Ur + (size_t)kh_ * X3_BTILE); \
    X3_BLOAD(rb[2], p1_ ? Wh + (size_t)kx_ * X3_BTILE : Uh + (size_t)kh_ * X3_BTILE); } while (0)
#define XG_BSTASH(buf) do { X3_BSTASH(buf, 0, rb[0]); X3_BSTASH(buf, 1, rb[1]); X3_BSTASH(buf, 2, rb[2]); } while (0)
#define XG_PAIR(g, I2)                                                                                   \
    { XG_APTR((g) + 2, pa); x3_tile<RG, NP, 3, 0, 1, I2, 0>(q0, q1, S.B[0], pa, lane, acc); }                    \
    XG_BSTASH(1); XG_BLOAD((g) + 2);                                                                     \
    __syncthreads();                                                                                     \
    { XG_APTR((g) + 3, pb); x3_tile<RG, NP, 3, 0, 1, I2, 0>(q2, q3, S.B[1], pb, lane, acc); }                    \
    XG_BSTASH(0); XG_BLOAD((g) + 3);                                                                     \
    __syncthreads()
  X3A<RG> q0, q1, q2, q3;
  X3_BVEC rb[3];
  { XG_APTR(0, p0); x3_load_A<RG, NP>(q0, p0, 0); x3_load_A<RG, NP>(q1, p0, 1); }
  { XG_APTR(1, p1); x3_load_A<RG, NP>(q2, p1, 0); x3_load_A<RG, NP>(q3, p1, 1); }
  XG_BLOAD(0); XG_BSTASH(0); XG_BLOAD(1);
  if (tid < 201) S.tansig[tid] = ts_v;
  __syncthreads();
  X3_STAMP(1);
#pragma unroll 1
  for (int g = 0; g < T1; g += 2) { XG_PAIR(g, 2); }
  X3_STAMP(2);
#pragma unroll 1
  for (int g = T1; g < TT; g += 2) { XG_PAIR(g, 3); }
  X3_STAMP(3);
#undef XG_PAIR
#undef XG_BSTASH
#undef XG_BLOAD
#undef XG_APTR
#undef XG_SEL
  {
    const float bh = b[2 * N + col];
    float *T = reinterpret_cast<float *>(&S.B[0][0][0][0]) + wave * 32 * X3_TLD;
    uint4 *Sx = h_newS ? h_newS + ((size_t)mt128 * NTn + nt) * (NP * X3_PLANE) : nullptr;
    // previous state for the blend: the loads of all row groups in flight before any activation arithmetic
    float ho[RG][16];
#pragma unroll
    for (int rg = 0; rg < RG; rg++)
#pragma unroll
      for (int i = 0; i < 16; i++)
        ho[rg][i] = h_old[(size_t)(mt * XMB + 32 * RG * wave + 32 * rg + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * N + col];
#pragma unroll
    for (int rg = 0; rg < RG; rg++) {
      const int grow0 = mt * XMB + 32 * RG * wave + 32 * rg;
      float v[16];
      pn_gru_gate16(acc[rg][0], acc[rg][1], acc[rg][2], acc[rg][3], ho[rg], bh, act, S.tansig, v);
      x3_store_tile<NP>(T, v, h_new, N, nt * 32, N, grow0, n_rows, Sx, srow + 32 * rg, lane);
    }
  }
#ifdef PN_X3_CLOCKS
  X3_STAMP(4);
  if (tid == 0 && N == 512 && blockIdx.x < 4096) {
    unsigned long long *t = pn_x3_trace + (size_t)blockIdx.x * 8;
    for (int i = 0; i < 5; i++) t[i] = ck_[i];
    t[5] = wall_clock64(); t[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[7] = w0_;
  }
#endif
}

// ---- GRU step, paired-phase form (large batches) ------------------------------------------------------------------
// Why: in pn_gru_x3_kernel the two 4-wave blocks of a CU drift into lock-step — both in their K loop (the SIMD's matrix
// pipe shared), then both in the gating epilogue (its VALU shared), then both waiting for their first operands: rocprofv3
// showed the matrix pipe 32 % busy and the VALU 33 % busy over the fp16-operand launch, never at the same time
// (profiles/r03m_pmc_per_launch_fp16.csv).  This kernel pins the complementary schedule instead: one persistent
// 8-wave block per CU = two groups of four waves (one wave of each group per SIMD), each group owning its own 256-row
// x 32-neuron tile; while group A runs the K loop of its tile (matrix pipe), group B runs the gating epilogue of its
// previous tile and the prologue of its next one (VALU, LDS, VMEM), then the roles swap.  The phases are kept aligned by
// the block barrier the K loop needs anyway (one per 32-k tile): the epilogue is cut into steps, one per barrier.
//   phase p:      group p&1 = K loop of its tile (TT barriers)   |   the other group = epilogue + prologue (TT barriers)
// Tiles: XCD x owns the M tiles mt = x + 8 i; its 2 * (blocks per XCD) groups walk the (mt, nt) list of that XCD with
// stride 2 * blocks-per-XCD, so that all column tiles of an M tile are in flight together (its A shadow stays in that
// XCD's L2) and every group keeps the same weight tiles for its whole walk when the stride is a multiple of N/32.
// Same MFMAs, same k order per accumulator, same gating arithmetic as pn_gru_x3_kernel (the epilogue is written on
// plain f32 instructions, each separately rounded).
struct X3PShared {
  uint4 B[2][3][3][4][64];             // [group][ring slot][gate tile][2 * kstep + plane][lane]: 72 KB
  float T[4][32 * X3_TLD];             // epilogue stage, shared by wave w and wave w + 4 (never both in an E phase): 18 KB
  float H[8][64 * 32];                 // previous state of each wave's 64 x 32 tile (for the blend), row-major: 64 KB
  float tansig[208];
};

// tansig_approx (vec.h:53-75) in two halves around the table read (pn_tansig_arg / pn_tansig_fin), written for the
// epilogue wave of the paired-phase kernel.  PLAIN f32 instructions on purpose: v_pk_mul_f32 / v_pk_add_f32 issued beside a
// wave that keeps the SIMD's matrix pipe busy take 41 cycles each instead of 9.5 (tools/probes/mfma_valu_pair_probe.hip,
// profiles/r04_mfma_valu_pair_probe.log: plain f32, integer and conversion instructions are unaffected) — the file is built
// with -fno-slp-vectorize so that the compiler does not pack them either.  The reference's table index is
// (int)floor(.5f + 25 |x|) by cvttss2si — out of range or NaN gives INT_MIN — clamped to [0, 200]; here: v_cvt_i32_f32 of
// the NEGATED value (saturates at INT_MIN, NaN -> 0), negated back (INT_MIN stays INT_MIN), then the clamp — the same
// index for every input.  The sign is carried as the sign BIT of x instead of a +-1 factor (the interpolated value is
// never negative), so x = -0 returns -0 where the reference returns +0.
struct X3Ts { float x; int sb, i; };
__device__ __forceinline__ int x3_tab_index(float v) {
  int c;
  asm("v_cvt_i32_f32_e64 %0, -%1" : "=v"(c) : "v"(v));
  int i = (int)(0u - (unsigned)c);
  i = i > 200 ? 200 : i;
  i = i < 0 ? 0 : i;
  return i;
}
__device__ __forceinline__ X3Ts x3_ts_arg(float x) {
  X3Ts a;
  const int xb = __builtin_bit_cast(int, x);
  a.sb = xb & (int)0x80000000;
  const float ax = __builtin_bit_cast(float, xb & 0x7fffffff);
  a.i = x3_tab_index(__builtin_floorf(.5f + 25.f * ax));
  a.x = ax - .04f * (float)a.i;
  return a;
}
__device__ __forceinline__ float x3_ts_fin(const X3Ts &a, float y) {
  const float dy = 1.f - y * y;
  const float r = y + a.x * dy * (1.f - y * a.x);
  return __builtin_bit_cast(float, __builtin_bit_cast(int, r) | a.sb);
}

// Pins a value where it is computed: without it the optimiser sinks every stage's arithmetic across the step barriers down
// to its last use (the stores), which keeps all intermediate values alive (spills) and undoes the step balance.
__device__ __forceinline__ void x3_pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void x3_pin(int &v) { asm volatile("" : "+v"(v)); }

#ifdef PN_X3_CLOCKS
__device__ unsigned long long pn_x3p_trace[256 * 2 * 8];
__device__ unsigned pn_x3p_hwid[256 * 8];              // HW_ID of every wave of every block of the last N = 512 launch
extern "C" PN_EXPORT int pn_x3p_trace_read(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pn_x3p_trace), sizeof(unsigned long long) * 256 * 2 * 8) == hipSuccess ? 0 : -1;
}
extern "C" PN_EXPORT int pn_x3p_hwid_read(unsigned *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pn_x3p_hwid), sizeof(unsigned) * 256 * 8) == hipSuccess ? 0 : -1;
}
#endif

#ifndef PN_XP_PRIO
#define PN_XP_PRIO 2
#endif
#define XP_EPI_STEPS 32                // barriers the epilogue + prologue steps of a phase use; the K loop must have more

// KTx = k-tiles of the input panels, NTn = N / 32 = k-tiles of the recurrent operand = column tiles: compile-time, so that
// every step of a phase is a straight-line piece of code with its own constants (register sets, ring slots, tile numbers)
template <int NP, int KTx, int NTn>
__global__ __launch_bounds__(512, 1) void pn_gru_x3p_kernel(
    PnSegs X, const float *__restrict__ h_old, const uint4 *__restrict__ h_oldS, const uint4 *__restrict__ Wp,
    const uint4 *__restrict__ Up, const float *__restrict__ b, int tps,
    const float *__restrict__ tansig, float *__restrict__ h_new, uint4 *__restrict__ h_newS, int n_rows, int n_mtiles) {
  constexpr int RG = 2, XMB = 256, N = 32 * NTn;
  static_assert(KTx % 2 == 0 && NTn % 2 == 0 && KTx + NTn >= XP_EPI_STEPS, "k-tiles come in pairs; a phase needs its epilogue steps");
  __shared__ X3PShared S;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
  constexpr int KTh = NTn, T1 = KTx, TT = KTx + KTh;
  const int xcd = blockIdx.x & 7, bpx = gridDim.x >> 3;
  const int n_mtx = n_mtiles > xcd ? (n_mtiles - xcd + 7) >> 3 : 0;      // M tiles of this XCD
  const int Wx = n_mtx * NTn, NS = 2 * bpx, n_it = (Wx + NS - 1) / NS;   // tiles of this XCD, groups walking them, tiles per group
  const int slot = (blockIdx.x >> 3) * 2 + grp;
  if (n_it == 0) return;               // an XCD without M tiles (every wave of the block takes this exit)
  if (tid < 201) S.tansig[tid] = tansig[tid];
  const int srow = (64 * gw) & 127;
  float *T = S.T[gw];

  floatx16 acc[RG][4];
  constexpr int XP_D = NP == 1 ? 3 : 2;                   // fragment groups in flight (a group = 2 MFMAs with fp16 operands, 6 split)
  constexpr int XP_BD = NP == 1 ? 4 : 2;                  // weight tiles in flight global -> registers (a k-tile = 384 / 1152 matrix-pipe
                                                         // cycles; a weight tile out of L2 / MALL takes one to two thousand)
  X3A<RG> q0, q1, q2, q3;
  X3_BVEC rb[3];                                         // own prologue: weight tiles 0, 1, 2 on their way into the group's ring
  X3_BVEC pb[XP_BD][3];                                  // E phase: the partner group's weight tiles on their way into its ring
  const uint4 *pWz = Wp, *pUz = Up;                      // ... and the z-gate bases of the partner's tile
  X3B fr[XP_D];
  const uint4 *pfA = nullptr, *pfB = nullptr;            // A-operand pointers of the tiles the two register pairs are refilled from
  int mt = 0, nt = 0, mt128 = 0, lane_off = 0;           // the tile being (or about to be) accumulated
  float bh = 0.f;
  const uint4 *Wz = Wp, *Uz = Up;                        // z-gate weight tiles of column tile nt; r and h follow at gate strides
  constexpr size_t gsW = (size_t)NTn * KTx * X3_BTILE, gsU = (size_t)NTn * KTh * X3_BTILE;
  int c_sg = 0, c_kt = 0, c_g = 0;
  const uint4 *c_last = nullptr;
  PN_PANEL_LOCALS(X);
  (void)pld;

#define XP_SEL(gg) int g_ = (gg); g_ = g_ < TT ? g_ : TT - 1; const bool p1_ = g_ < T1; \
    const int kx_ = p1_ ? g_ : 0, kh_ = p1_ ? 0 : g_ - T1
#define XP_APTR(pt) \
    const uint4 *pt; { if (c_g < T1) { \
        c_last = reinterpret_cast<const uint4 *>(pn_seg_ptr(PN_PANEL_PASS, c_sg)) + ((size_t)mt128 * tps + c_kt) * (NP * X3_PLANE) + lane_off; \
        c_kt++; if (c_kt == tps) { c_kt = 0; c_sg++; } \
      } else if (c_g < TT) { c_last = h_oldS + ((size_t)mt128 * NTn + (c_g - T1)) * (NP * X3_PLANE) + lane_off; } \
      c_g++; pt = c_last; }
#define XP_BLD(dst, src) (dst) = reinterpret_cast<const X3_BVEC *>(src)[gtid]
  // weight tile gg (clamped to the last one) of the walk position whose z-gate bases are wz_ / uz_ -> register set dst[3]
#define XP_BLOADP(dst, wz_, uz_, gg) do { XP_SEL(gg); \
    const uint4 *t0_ = p1_ ? (wz_) + (size_t)kx_ * X3_BTILE : (uz_) + (size_t)kh_ * X3_BTILE; const size_t gs_ = p1_ ? gsW : gsU; \
    XP_BLD(dst[0], t0_); XP_BLD(dst[1], t0_ + gs_); XP_BLD(dst[2], t0_ + 2 * gs_); } while (0)
#define XP_BST(g_, slot, t, v) reinterpret_cast<X3_BVEC *>(&S.B[g_][slot][t][0][0])[gtid] = (v)
#define XP_BSTASHP(g_, slot, src) do { XP_BST(g_, slot, 0, src[0]); XP_BST(g_, slot, 1, src[1]); XP_BST(g_, slot, 2, src[2]); } while (0)
  // Weight staging for the PARTNER group's K phase, one call per barrier interval s of that phase (the partner wave of a
  // SIMD issues them because a global load issued by the wave that feeds the matrix pipe stalls its MFMA stream, a load
  // issued by the other wave of the SIMD does not: tools/probes/mfma_vmem_probe.hip).  The K group's own prologue has
  // put tiles 0, 1, 2 into its ring; interval s >= 1 adds tile s + 2 to slot (s + 2) % 3 — free since barrier s - 1 — from
  // register set (s + 2) % XP_BD, loaded XP_BD intervals earlier (tiles 3 .. 2 + XP_BD: at interval 0), and refills the set
  // with tile s + 2 + XP_BD.
#if defined(PN_XP_ABL) && (PN_XP_ABL & 8)       // timing ablation (results wrong): no weight staging during K phases
#define XP_STAGE(s_) do { } while (0)
#else
#define XP_STAGE(s_) do {                                                                                        \
    if ((s_) >= 1 && (s_) + 2 < TT) XP_BSTASHP(grp ^ 1, ((s_) + 2) % 3, pb[((s_) + 2) % XP_BD]);                 \
    if ((s_) == 0) { _Pragma("unroll") for (int j_ = 0; j_ < XP_BD; j_++) if (3 + j_ < TT) XP_BLOADP(pb[(3 + j_) % XP_BD], pWz, pUz, 3 + j_); } \
    else if ((s_) + 2 + XP_BD < TT) XP_BLOADP(pb[((s_) + 2) % XP_BD], pWz, pUz, (s_) + 2 + XP_BD);               \
  } while (0)
#endif
  // z-gate weight bases of the partner group's tile with walk index it_
#define XP_PARTNER(it_) do {                                                                                     \
    int w_ = (slot ^ 1) + (it_) * NS; w_ = w_ < Wx ? w_ : Wx - 1;                                                \
    const int nt_ = w_ % NTn;                                                                                    \
    pWz = Wp + (size_t)nt_ * KTx * X3_BTILE; pUz = Up + (size_t)nt_ * KTh * X3_BTILE;                             \
  } while (0)
  // the K loop's barrier: the fragment reads of the next tile issued at the end of this one (XP_D - 1 groups of NP reads)
  // stay in flight across it
#define XP_KBAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F | (((XP_D - 1) * NP) << 8)); \
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#if defined(PN_XP_ABL) && (PN_XP_ABL & 2)       // timing ablation (results wrong): the K phase is only its barriers
#define XP_PAIR(g, I2) XP_KBAR(); XP_KBAR()
#else
  // two k-tiles: tile g from ring slot s0 with the A registers q0/q1 (refilled from pfA = tile g + 2), tile g + 1 from s1
  // with q2/q3 (pfB); each tile's hook advances the A cursor for the other tile's next pointer
#define XP_PAIR(g, I2)                                                                                           \
    x3_tile_r<RG, NP, 3, XP_D, 0, 1, I2, 0>(q0, q1, S.B[grp][s0], S.B[grp][s1], pfA, lane, acc, fr,               \
        [&]() { }, [&]() { XP_APTR(pn_); pfB = pn_; });                                                           \
    XP_KBAR();                                                                                                   \
    x3_tile_r<RG, NP, 3, XP_D, 0, 1, I2, 0>(q2, q3, S.B[grp][s1], S.B[grp][s2], pfB, lane, acc, fr,               \
        [&]() { }, [&]() { XP_APTR(pn_); pfA = pn_; });                                                           \
    XP_KBAR();                                                                                                   \
    { const int s_ = s0; s0 = s2; s2 = s1; s1 = s_; }
#endif
  // prologue of the tile with walk index it_: coordinates, the first two A tiles and the first weight tile in flight ...
#define XP_PRO0(it_) do {                                                                                        \
    int w_ = slot + (it_) * NS; w_ = w_ < Wx ? w_ : Wx - 1;    /* past the end: a valid tile, loaded and never used */ \
    const int mtx_ = w_ / NTn;                                                                                   \
    nt = w_ - mtx_ * NTn; mt = xcd + 8 * mtx_;                                                                   \
    mt128 = mt * 2 + (gw >> 1); lane_off = (lane >> 5) * 128 + srow + (lane & 31);                               \
    Wz = Wp + (size_t)nt * KTx * X3_BTILE; Uz = Up + (size_t)nt * KTh * X3_BTILE;                                 \
    c_sg = 0; c_kt = 0; c_g = 0;                                                                                 \
    { XP_APTR(p0_); x3_load_A<RG, NP>(q0, p0_, 0); x3_load_A<RG, NP>(q1, p0_, 1); }                               \
    { XP_APTR(p1_); x3_load_A<RG, NP>(q2, p1_, 0); x3_load_A<RG, NP>(q3, p1_, 1); }                               \
    { XP_APTR(p2_); pfA = p2_; }                                                                                 \
    XP_BLOADP(rb, Wz, Uz, 0);                                                                                    \
  } while (0)
  // ... weight tiles 0, 1, 2 into the group's ring slots 0, 1, 2 (one per call) ...
#define XP_PRO1(j_) do { XP_BSTASHP(grp, j_, rb); if ((j_) < 2) XP_BLOADP(rb, Wz, Uz, (j_) + 1); } while (0)
  // ... and the accumulators = biases
#define XP_PRO2() do {                                                                                           \
    const int col_ = nt * 32 + (lane & 31);                                                                      \
    float bz_ = b[col_]; bz_ += b[3 * N + col_];                                                                 \
    float br_ = b[N + col_]; br_ += b[4 * N + col_];                                                             \
    const float bt_ = b[5 * N + col_];                                                                           \
    bh = b[2 * N + col_];                                                                                        \
    _Pragma("unroll") for (int rg = 0; rg < RG; rg++)                                                            \
      _Pragma("unroll") for (int i = 0; i < 16; i++) { acc[rg][0][i] = bz_; acc[rg][1][i] = br_; acc[rg][2][i] = 0.f; acc[rg][3][i] = bt_; } \
  } while (0)
  // previous state of the tile just prepared by XP_PRO0: 8 coalesced 16-byte loads per lane (row 8 j + lane / 8) ...
#define XP_HOLOAD() do {                                                                                         \
    const float *hp_ = h_old + (size_t)(mt * XMB + 64 * gw + (lane >> 3)) * N + nt * 32 + (lane & 7) * 4;        \
    _Pragma("unroll") for (int j = 0; j < 8; j++) hq[j] = *reinterpret_cast<const fvec4 *>(hp_ + (size_t)(8 * j) * N); \
  } while (0)
  // ... into the wave's LDS slice (read back in accumulator layout by the blend)
#define XP_HOSTASH() do {                                                                                        \
    float *hd_ = S.H[wave] + (lane >> 3) * 32 + (lane & 7) * 4;                                                  \
    _Pragma("unroll") for (int j = 0; j < 8; j++) *reinterpret_cast<fvec4 *>(hd_ + 8 * j * 32) = hq[j];           \
  } while (0)
  // barrier of an E-phase / idle step: everything but the last n_ LDS instructions of the wave (its table reads for the next
  // step) has completed — in particular the stash of the partner's weight tile, which the K waves read after this barrier
#define XP_BARN(n_) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F | ((n_) << 8)); \
    __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define XP_BAR() XP_BARN(0)
  // a phase in which the group only stages the partner's weight tiles (group 1 before its first tile, group 0 after its last)
#define XP_IDLE_PHASE(it_) do {                                                                                  \
    XP_PARTNER(it_);                                                                                             \
    _Pragma("unroll") for (int s = 0; s < TT; s++) { XP_STAGE(s); XP_BAR(); }                                     \
  } while (0)

  // Every group runs the same straight sequence — prologue, then per tile a K phase and an E phase — group 1 one phase
  // behind group 0 (an idle phase before its first tile, one after group 0's last).  A group whose walk runs past the
  // end of the XCD's list clamps to the last tile and recomputes it (identical stores): no data-dependent control flow.
  fvec4 hq[8];
  XP_PRO0(0); XP_HOLOAD(); XP_PRO1(0); XP_PRO1(1); XP_PRO1(2); XP_PRO2(); XP_HOSTASH();
  __syncthreads();
#ifdef PN_X3_CLOCKS
  unsigned long long ck_k = 0, ck_e = 0; const unsigned long long ck_0 = __builtin_readcyclecounter();
#endif
  if (grp == 1) XP_IDLE_PHASE(0);
#pragma unroll 1
  for (int it = 0; it < n_it; it++) {
#ifdef PN_X3_CLOCKS
    const unsigned long long ck_a = __builtin_readcyclecounter();
#endif
    // ---- K phase: TT barriers -----------------------------------------------------------------------------------
    {
      int s0 = 0, s1 = 1, s2 = 2;                              // ring slots of tiles g, g + 1, g + 2
#pragma unroll
      for (int j = 0; j < XP_D - 1; j++) fr[j] = x3_read_B<NP>(S.B[grp][0], j % 3, j / 3, lane);
      __builtin_amdgcn_s_setprio(PN_XP_PRIO);                  // the matrix-pipe stream outranks the partner wave's VALU work
#pragma unroll 1
      for (int g = 0; g < T1; g += 2) { XP_PAIR(g, 2); }
#pragma unroll 1
      for (int g = T1; g < TT; g += 2) { XP_PAIR(g, 3); }
      __builtin_amdgcn_s_setprio(0);
    }
#ifdef PN_X3_CLOCKS
    const unsigned long long ck_b = __builtin_readcyclecounter();
    ck_k += ck_b - ck_a;
#endif
    // ---- E phase: epilogue of this tile, prologue of the next one: TT barriers -------------------------------------
    {
      const int col = nt * 32 + (lane & 31);
      const int row0 = mt * XMB + 64 * gw;
      const int nt_e = nt;
      const float bh_e = bh;
      uint4 *Sx = h_newS ? h_newS + ((size_t)mt128 * NTn + nt) * (NP * X3_PLANE) : nullptr;
      // the previous state for the blend waits in the wave's LDS slice (XP_HOLOAD / XP_HOSTASH of the prologue), and the new
      // state overwrites the W_h x accumulator (acc[rg][2]) as it is formed: nothing extra is live beside the accumulators
      const float *Hl = S.H[wave] + (4 * (lane >> 5)) * 32 + (lane & 31);
      // The gating of the tile's 32 outputs per lane (value v = 16 rg + i) in three stages — A: arguments of z and r + their
      // table reads; B: z, r, candidate pre-activation, its argument + table read; C: candidate, blend — spread EVENLY over
      // the steps: stage A of value v runs in step (3 v) >> 2, B one step later, C two steps later (every table value is read
      // one barrier interval before it is used), i.e. four stage-tasks (~130 instructions) per step over steps 0..25.  The
      // first version ran a whole pair per step in 18 steps: those steps took three K-loop intervals each while the K
      // group idled at the barrier, and the 14 remaining steps ran at the K loop's pace with the epilogue wave idle — the
      // two phases ADDED (35 k cycles) instead of overlapping.
      X3Ts za[32], ra[32], ha[32];
      float zy[32], ry[32], hy[32], zv[32];
      XP_PARTNER(it + grp);                                    // group 0's partner runs its tile `it`, group 1's already `it + 1`
      // step s of the phase = barrier interval s of the partner's K phase: its weight staging first, then this group's own
      // work:  0..25 gating | 22 next tile's first operands | 25, 28, 31 own weight tiles 0, 1, 2 into the ring (each loaded
      // three steps before) | 26..29 the tile leaves through the LDS stage (26: next tile's previous state requested) |
      // 30 biases | 31 previous state into its LDS slice
#pragma unroll
      for (int s = 0; s < TT; s++) {
        XP_STAGE(s);
        int n_reads = 0;                                       // table reads issued in this step for the next one
#if !(defined(PN_XP_ABL) && (PN_XP_ABL & 1))       // timing ablation (results wrong): no gating arithmetic
#pragma unroll
        for (int v = 0; v < 32; v++) {
          if (((3 * v) >> 2) + 2 != s) continue;
          const int rg = v >> 4, i = v & 15;
          const float hc = x3_ts_fin(ha[v], hy[v]);
          const float hov = Hl[(32 * rg + (i & 3) + 8 * (i >> 2)) * 32];
          float o = zv[v] * hov + (1.f - zv[v]) * hc;
          x3_pin(o);
          acc[rg][2][i] = o;
        }
#pragma unroll
        for (int v = 0; v < 32; v++) {
          if (((3 * v) >> 2) + 1 != s) continue;
          const int rg = v >> 4, i = v & 15;
          const float z = .5f + .5f * x3_ts_fin(za[v], zy[v]);
          const float r = .5f + .5f * x3_ts_fin(ra[v], ry[v]);
          float hp = bh_e + acc[rg][3][i] * r;
          hp = hp + acc[rg][2][i];
          ha[v] = x3_ts_arg(hp);
          hy[v] = S.tansig[ha[v].i];
          zv[v] = z;
          x3_pin(ha[v].x); x3_pin(ha[v].sb); x3_pin(zv[v]);
          n_reads += 1;
        }
#pragma unroll
        for (int v = 0; v < 32; v++) {
          if (((3 * v) >> 2) != s) continue;
          const int rg = v >> 4, i = v & 15;
          za[v] = x3_ts_arg(.5f * acc[rg][0][i]);
          ra[v] = x3_ts_arg(.5f * acc[rg][1][i]);
          zy[v] = S.tansig[za[v].i];
          ry[v] = S.tansig[ra[v].i];
          x3_pin(za[v].x); x3_pin(za[v].sb); x3_pin(ra[v].x); x3_pin(ra[v].sb);
          n_reads += 2;
        }
#endif
        if (s == 22) XP_PRO0(it + 1);
        if (s == 26 || s == 28) {
          const int rg = (s - 26) >> 1;
          float vo[16];
#pragma unroll
          for (int i = 0; i < 16; i++) vo[i] = acc[rg][2][i];
          x3_stage_write(T, vo, lane);
          x3_stage_store<NP>(T, 0, h_new, N, nt_e * 32, N, row0 + 32 * rg, n_rows, Sx, srow + 32 * rg, lane);
          if (s == 26) XP_HOLOAD();                            // every blend has read the wave's slice
        }
        if (s == 27 || s == 29) {
          const int rg = (s - 27) >> 1;
          x3_stage_store<NP>(T, 1, h_new, N, nt_e * 32, N, row0 + 32 * rg, n_rows, Sx, srow + 32 * rg, lane);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        if (s == 25) XP_PRO1(0);
        if (s == 28) XP_PRO1(1);
        if (s == 30) XP_PRO2();
        if (s == 31) { XP_PRO1(2); XP_HOSTASH(); }
        // the barrier: the table reads issued for the next step stay in flight across it
        if (s == TT - 1) __syncthreads();                      // the group's ring slots 0, 1, 2 are complete for its K phase
        else if (n_reads == 0) XP_BAR(); else if (n_reads == 1) XP_BARN(1); else if (n_reads == 2) XP_BARN(2);
        else if (n_reads == 3) XP_BARN(3); else if (n_reads == 4) XP_BARN(4); else if (n_reads == 5) XP_BARN(5); else XP_BARN(6);
      }
    }
#ifdef PN_X3_CLOCKS
    ck_e += __builtin_readcyclecounter() - ck_b;
#endif
  }
  if (grp == 0) XP_IDLE_PHASE(n_it - 1);
#ifdef PN_X3_CLOCKS
  if (lane == 0 && N == 512 && blockIdx.x < 256) pn_x3p_hwid[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
  if (lane == 0 && gw == 0 && N == 512 && blockIdx.x < 256) {
    unsigned long long *t = pn_x3p_trace + ((size_t)blockIdx.x * 2 + grp) * 8;
    t[0] = ck_k; t[1] = ck_e; t[2] = __builtin_readcyclecounter() - ck_0; t[3] = n_it; t[4] = TT;
  }
#endif
#undef XP_SEL
#undef XP_APTR
#undef XP_BLD
#undef XP_BST
#undef XP_PAIR
#undef XP_PRO0
#undef XP_PRO1
#undef XP_PRO2
#undef XP_KBAR
#undef XP_BLOADP
#undef XP_BSTASHP
#undef XP_STAGE
#undef XP_PARTNER
#undef XP_BARN
#undef XP_IDLE_PHASE
#undef XP_HOLOAD
#undef XP_HOSTASH
#undef XP_BAR
}

// ---- fp32 rows -> fragment-order hi/lo shadow (the first layer's output; RNN state loaded from the host) ---------
// one thread per (row, k-group of 8): reads 32 bytes, writes 2 x 16
template <int NP>
__global__ __launch_bounds__(256) void pn_split_x3_kernel(const float *__restrict__ src, int ld, int width, uint4 *__restrict__ S,
                                                          int n_rows_padded) {
  const int kgs = width >> 3;                                    // k-groups per row
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = idx / kgs;
  const int kgi = (int)(idx - row * kgs);
  if (row >= (size_t)n_rows_padded) return;
  float f[8];
  *reinterpret_cast<float4 *>(&f[0]) = *reinterpret_cast<const float4 *>(src + row * ld + 8 * kgi);
  *reinterpret_cast<float4 *>(&f[4]) = *reinterpret_cast<const float4 *>(src + row * ld + 8 * kgi + 4);
  uint4 hi, lo;
  x3_split8(f, hi, lo);
  uint4 *chunk = S + ((row >> 7) * (width >> 5) + (kgi >> 2)) * (NP * X3_PLANE);
  chunk[(kgi & 3) * 128 + (row & 127)] = hi;
  if constexpr (NP == 2) chunk[X3_PLANE + (kgi & 3) * 128 + (row & 127)] = lo;
}

// ---- host: weight packing W[K][ncols] -> [CT][ceil(K/32)][k-step 2][plane np][lane 64][8 halfs] ------------------
static inline int x3_ct_padded(int ncols, int ct_round) {
  const int CT = (ncols + 31) / 32;
  return ((CT + ct_round - 1) / ct_round) * ct_round;
}
size_t pn_packed_halfs_x3(int k_alloc, int ncols, int ct_round, int np) {
  return (size_t)x3_ct_padded(ncols, ct_round) * ((k_alloc + 31) / 32) * 1024 * np;
}
// returns 0, or -1 if a weight is outside the fp16 range (the mode cannot represent it)
int pn_pack_weights_x3(const float *W, int K, int k_alloc, int ncols, int ct_round, int np, void *out) {
  _Float16 *Wp = (_Float16 *)out;
  const int CT = x3_ct_padded(ncols, ct_round), KT = (k_alloc + 31) / 32;
  for (int ct = 0; ct < CT; ct++)
    for (int kt = 0; kt < KT; kt++) {
      _Float16 *tile = Wp + ((size_t)ct * KT + kt) * 1024 * np;
      for (int s = 0; s < 2; s++)
        for (int lane = 0; lane < 64; lane++)
          for (int j = 0; j < 8; j++) {
            const int k = kt * 32 + 16 * s + 8 * (lane >> 5) + j, c = ct * 32 + (lane & 31);
            const float w = (k < K && c < ncols) ? W[(size_t)k * ncols + c] : 0.f;
            if (!(w > -65504.f && w < 65504.f)) return -1;
            const _Float16 hi = (_Float16)w;
            tile[((np * s + 0) * 64 + lane) * 8 + j] = hi;
            if (np == 2) tile[((2 * s + 1) * 64 + lane) * 8 + j] = (_Float16)(w - (float)hi);
          }
    }
  return 0;
}

int pn_dense_x3_nt(int N) { return N >= 128 ? 4 : 2; }

// rows per wave: 2 row groups of 32 (256-row blocks, two per CU: fewest operand bytes per MFMA, best when the grid fills
// the chip several times over) or 1 (128-row blocks, three per CU: twice the blocks, shorter chains — measured 0.45 vs 0.60 ms
// per frame at 1024 streams, 0.60 vs 0.68 at 4096, equal at 16 384, 0.60 vs 0.585 per GRU step at 65 536).  The context
// fixes the choice at creation (and its self-test runs the same instantiation); PERCEPNET_X3_RG=1|2 overrides.
int pn_x3_rg_for(int n_rows) {
  const char *e = getenv("PERCEPNET_X3_RG");           // read at every context creation (tests switch it between contexts)
  const int env = e ? atoi(e) : 0;
  if (env >= 1 && env <= 3) return env;
  // 3 = 64 rows per wave with the GRUs on the paired-phase kernel (pn_gru_x3p_kernel): opt-in only — measured at parity
  // with the one-tile-per-block kernel (0.305 vs 0.291 ms fp16 operands, 0.57 vs 0.59 ms split precision)
  return n_rows >= 32768 ? 2 : 1;
}

// A: panels carry the uint4* shadows of equally wide buffers (width = logical columns, a multiple of 32);
// out (fp32, optional) / outS (shadow of a buffer nts_out column tiles wide, optional)
int pn_launch_dense_x3(hipStream_t st, const PnSegs &A, const void *Wp, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, void *outS, int nts_out, int n_rows, int rg, int np) {
  if (rg == 3) rg = 2;                 // the paired-phase form exists for the GRUs only
  const int tps = A.width[0] / 32, KT = tps * A.n;
  // the K loop consumes k-tiles in pairs and clamps its prefetch to the last tile: an odd count would accumulate that
  // tile twice; panels must be whole 32-column tiles of equal width (every layer of the fixed topology is: 20 / 48 / 80)
  if (pn_check_dense_geometry("pn_launch_dense_x3", A.n, A.width, 1)) return -1;
  const int NT = pn_dense_x3_nt(N);
  const int n_mtiles = (n_rows + 128 * rg - 1) / (128 * rg);
  const int n_cblocks = x3_ct_padded(N, NT) / NT;
  const int grid = 8 * ((n_mtiles + 7) / 8) * n_cblocks;
#define XD_LAUNCH(NT_) do { if (np == 2) { if (rg == 2) XD_LAUNCH2(2, 2, NT_); else XD_LAUNCH2(1, 2, NT_); } \
                            else { if (rg == 2) XD_LAUNCH2(2, 1, NT_); else XD_LAUNCH2(1, 1, NT_); } } while (0)
#define XD_LAUNCH2(RG_, NP_, NT_)                                                                                \
  hipLaunchKernelGGL((pn_dense_x3_kernel<RG_, NP_, NT_>), dim3(grid), dim3(NN_THREADS), 0, st, A, (const uint4 *)Wp, bias, N, \
                     KT, tps, act, tansig, out, ldo, (uint4 *)outS, nts_out, n_rows, n_mtiles, n_cblocks)
  if (NT == 4) XD_LAUNCH(4); else XD_LAUNCH(2);
#undef XD_LAUNCH
#undef XD_LAUNCH2
  return 0;
}

// compute units of the current device (the paired-phase kernel runs one persistent block per CU)
static int x3_cu_count() {
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

int pn_launch_gru_x3(hipStream_t st, const PnSegs &X, const float *h_old, const void *h_oldS, const void *Wp,
                      const void *Up, const float *b, int N, int act, const float *tansig, float *h_new, void *h_newS,
                      int n_rows, int rg, int np) {
  const int tps = X.width[0] / 32, KTx = tps * X.n;
  const int NTn = N / 32;
  if (pn_check_gru_geometry("pn_launch_gru_x3", X.n, X.width, N)) return -1;   // k-tiles are consumed in pairs (x: 16 / 32, h: 16 / 4)
  // rg 3: paired-phase kernel (one 8-wave block per CU, K loop of one wave group beside the epilogue of the other); it
  // is written for the tanh candidate and instantiated for the two GRU geometries of the network (512 -> 512 and
  // 1024 -> 128); anything else runs on the 64-rows-per-wave kernel
  if (rg == 3 && act == ACT_TANH && ((KTx == 16 && NTn == 16) || (KTx == 32 && NTn == 4))) {
    const int n_mtiles = (n_rows + 255) / 256;
    const int grid = 8 * (x3_cu_count() / 8);
#define XP_LAUNCH(NP_, KTX_, NTN_)                                                                                  \
    hipLaunchKernelGGL((pn_gru_x3p_kernel<NP_, KTX_, NTN_>), dim3(grid), dim3(512), 0, st, X, h_old, (const uint4 *)h_oldS, \
                       (const uint4 *)Wp, (const uint4 *)Up, b, tps, tansig, h_new, (uint4 *)h_newS, n_rows, n_mtiles)
    if (NTn == 16) { if (np == 2) XP_LAUNCH(2, 16, 16); else XP_LAUNCH(1, 16, 16); }
    else { if (np == 2) XP_LAUNCH(2, 32, 4); else XP_LAUNCH(1, 32, 4); }
#undef XP_LAUNCH
    return 0;
  }
  if (rg == 3) rg = 2;
  const int n_mtiles = (n_rows + 128 * rg - 1) / (128 * rg);
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
#define XG_LAUNCH(RG_, NP_)                                                                                           \
  hipLaunchKernelGGL((pn_gru_x3_kernel<RG_, NP_>), dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, (const uint4 *)h_oldS,  \
                     (const uint4 *)Wp, (const uint4 *)Up, b, N, KTx, tps, act, tansig, h_new, (uint4 *)h_newS, n_rows,   \
                     n_mtiles)
  if (np == 2) { if (rg == 2) XG_LAUNCH(2, 2); else XG_LAUNCH(1, 2); }
  else { if (rg == 2) XG_LAUNCH(2, 1); else XG_LAUNCH(1, 1); }
#undef XG_LAUNCH
  return 0;
}

// refuses (-1, pn_set_error, nothing launched) a width that is not whole groups of 8 columns or a plane count other than 1 / 2
int pn_launch_split_x3(hipStream_t st, const float *src, int ld, int width, void *S, int n_rows_padded, int np) {
  if (width < 8 || (width & 7) || ld < width || n_rows_padded < 1 || (np != 1 && np != 2) || !src || !S) {
    pn_set_error("pn_launch_split_x3: width %d (whole groups of 8), row stride %d, %d rows, %d plane(s)", width, ld, n_rows_padded, np);
    return -1;
  }
  const size_t n = (size_t)n_rows_padded * (width >> 3);
  if (np == 2)
    hipLaunchKernelGGL(pn_split_x3_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, width, (uint4 *)S,
                       n_rows_padded);
  else
    hipLaunchKernelGGL(pn_split_x3_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, width, (uint4 *)S,
                       n_rows_padded);
  return 0;
}
