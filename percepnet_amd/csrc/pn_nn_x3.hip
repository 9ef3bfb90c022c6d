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

// hi/lo planes of eight consecutive fp32 values (operands beyond the fp16 range saturate instead of turning into NaNs)
__device__ __forceinline__ void x3_split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
  half8 h, l;
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
    if (t == NT - 1) {
      x3_load_A<RG, NP>(s ? qb : qa, pf, s);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Stores one 32 x 32 output tile of a wave (rows row0.., v[i] = value of row (i&3) + 8(i>>2) + 4(lane>>5), column
// lane&31) through the wave's private LDS stage: fp32 rows as 16-byte stores (out may be null) and the hi/lo shadow
// planes as one 16-byte store per (row, k-group) (S may be null; Srow = row within the M tile of 128).
template <int NP>
__device__ __forceinline__ void x3_store_tile(float *T, const float (&v)[16], float *__restrict__ out, int ldo, int col0,
                                              int n_cols, int grow0, int n_rows, uint4 *__restrict__ S, int srow0, int lane) {
#pragma unroll
  for (int i = 0; i < 16; i++) T[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * X3_TLD + (lane & 31)] = v[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int p = 0; p < 2; p++) {
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
extern "C" int pn_x3_trace_read(unsigned long long *out) {
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
  static const int env = getenv("PERCEPNET_X3_RG") ? atoi(getenv("PERCEPNET_X3_RG")) : 0;
  if (env == 1 || env == 2) return env;
  return n_rows >= 32768 ? 2 : 1;
}

// A: panels carry the uint4* shadows of equally wide buffers (width = logical columns, a multiple of 32);
// out (fp32, optional) / outS (shadow of a buffer nts_out column tiles wide, optional)
void pn_launch_dense_x3(hipStream_t st, const PnSegs &A, const void *Wp, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, void *outS, int nts_out, int n_rows, int rg, int np) {
  const int tps = A.width[0] / 32, KT = tps * A.n;
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
}

void pn_launch_gru_x3(hipStream_t st, const PnSegs &X, const float *h_old, const void *h_oldS, const void *Wp,
                      const void *Up, const float *b, int N, int act, const float *tansig, float *h_new, void *h_newS,
                      int n_rows, int rg, int np) {
  const int tps = X.width[0] / 32, KTx = tps * X.n;
  const int n_mtiles = (n_rows + 128 * rg - 1) / (128 * rg), NTn = N / 32;
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
#define XG_LAUNCH(RG_, NP_)                                                                                           \
  hipLaunchKernelGGL((pn_gru_x3_kernel<RG_, NP_>), dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, (const uint4 *)h_oldS,  \
                     (const uint4 *)Wp, (const uint4 *)Up, b, N, KTx, tps, act, tansig, h_new, (uint4 *)h_newS, n_rows,   \
                     n_mtiles)
  if (np == 2) { if (rg == 2) XG_LAUNCH(2, 2); else XG_LAUNCH(1, 2); }
  else { if (rg == 2) XG_LAUNCH(2, 1); else XG_LAUNCH(1, 1); }
#undef XG_LAUNCH
}

void pn_launch_split_x3(hipStream_t st, const float *src, int ld, int width, void *S, int n_rows_padded, int np) {
  const size_t n = (size_t)n_rows_padded * (width >> 3);
  if (np == 2)
    hipLaunchKernelGGL(pn_split_x3_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, width, (uint4 *)S,
                       n_rows_padded);
  else
    hipLaunchKernelGGL(pn_split_x3_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, width, (uint4 *)S,
                       n_rows_padded);
}
