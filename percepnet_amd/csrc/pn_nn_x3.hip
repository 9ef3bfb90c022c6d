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
    if (t == NT - 1) {
      x3_load_A<RG, NP>(s ? qb : qa, pf, s);
      __builtin_amdgcn_sched_barrier(0);
    }
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
// float pairs so that it compiles to v_pk_mul_f32 / v_pk_add_f32 — separately rounded like the scalar form).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

struct X3PShared {
  uint4 B[2][2][3][4][64];             // [group][buffer][gate tile][2 * kstep + plane][lane]: 48 KB
  float T[8][32 * X3_TLD];             // epilogue stage of each wave: 36 KB
  float H[8][64 * 32];                 // previous state of each wave's 64 x 32 tile (for the blend), row-major: 64 KB
  float tansig[208];
};

// tansig_approx (vec.h:53-75) on a pair, in two halves around the table read (pn_tansig_arg / pn_tansig_fin).  The
// reference's index is (int)floor(.5f + 25 |x|) by cvttss2si — out of range or NaN gives INT_MIN — clamped to
// [0, 200]; here: v_cvt_i32_f32 of the NEGATED value (saturates at INT_MIN, NaN -> 0), negated back (INT_MIN stays
// INT_MIN), then the clamp — the same index for every input.  The sign is carried as the sign BIT of x instead of a
// +-1 factor (the interpolated value is never negative), so x = -0 returns -0 where the reference returns +0.
struct X3Ts2 { v2f x; v2i sb, i; };
__device__ __forceinline__ int x3_tab_index(float v) {
  int c;
  asm("v_cvt_i32_f32_e64 %0, -%1" : "=v"(c) : "v"(v));
  int i = (int)(0u - (unsigned)c);
  i = i > 200 ? 200 : i;
  i = i < 0 ? 0 : i;
  return i;
}
__device__ __forceinline__ X3Ts2 x3_ts_arg2(v2f x) {
  X3Ts2 a;
  const v2i xb = __builtin_bit_cast(v2i, x);
  a.sb = xb & (int)0x80000000;
  const v2f ax = __builtin_bit_cast(v2f, xb & 0x7fffffff);
  v2f v = .5f + 25.f * ax;
  v.x = __builtin_floorf(v.x); v.y = __builtin_floorf(v.y);
  a.i.x = x3_tab_index(v.x); a.i.y = x3_tab_index(v.y);
  const v2f fi = {(float)a.i.x, (float)a.i.y};
  a.x = ax - .04f * fi;
  return a;
}
__device__ __forceinline__ v2f x3_ts_fin2(const X3Ts2 &a, v2f y) {
  const v2f dy = 1.f - y * y;
  const v2f r = y + a.x * dy * (1.f - y * a.x);
  return __builtin_bit_cast(v2f, __builtin_bit_cast(v2i, r) | a.sb);
}

// Pins a value where it is computed: without it the optimiser sinks every stage's arithmetic across the step barriers down
// to its last use (the stores), which keeps all intermediate values alive (spills) and undoes the step balance.
__device__ __forceinline__ void x3_pin(v2f &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void x3_pin(v2i &v) { asm volatile("" : "+v"(v)); }

#ifdef PN_X3_CLOCKS
__device__ unsigned long long pn_x3p_trace[256 * 2 * 8];
extern "C" int pn_x3p_trace_read(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pn_x3p_trace), sizeof(unsigned long long) * 256 * 2 * 8) == hipSuccess ? 0 : -1;
}
#endif

#define XP_EPI_STEPS 24                // barriers the epilogue + prologue steps of a phase use; the K loop must have more

template <int NP>
__global__ __launch_bounds__(512, 1) void pn_gru_x3p_kernel(
    PnSegs X, const float *__restrict__ h_old, const uint4 *__restrict__ h_oldS, const uint4 *__restrict__ Wp,
    const uint4 *__restrict__ Up, const float *__restrict__ b, int N, int KTx, int tps,
    const float *__restrict__ tansig, float *__restrict__ h_new, uint4 *__restrict__ h_newS, int n_rows, int n_mtiles) {
  constexpr int RG = 2, XMB = 256;
  __shared__ X3PShared S;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
  const int NTn = N >> 5, KTh = NTn, T1 = KTx, TT = KTx + KTh;
  const int xcd = blockIdx.x & 7, bpx = gridDim.x >> 3;
  const int n_mtx = n_mtiles > xcd ? (n_mtiles - xcd + 7) >> 3 : 0;      // M tiles of this XCD
  const int Wx = n_mtx * NTn, NS = 2 * bpx, n_it = (Wx + NS - 1) / NS;   // tiles of this XCD, groups walking them, tiles per group
  const int slot = (blockIdx.x >> 3) * 2 + grp;
  if (n_it == 0) return;               // an XCD without M tiles (every wave of the block takes this exit)
  if (tid < 201) S.tansig[tid] = tansig[tid];
  const int srow = (64 * gw) & 127;
  float *T = S.T[wave];

  floatx16 acc[RG][4];
  X3A<RG> q0, q1, q2, q3;
  X3_BVEC rb[3];
  int mt = 0, nt = 0, mt128 = 0, lane_off = 0;           // the tile being (or about to be) accumulated
  float bh = 0.f;
  const uint4 *Wz = Wp, *Uz = Up;                        // z-gate weight tiles of column tile nt; r and h follow at gate strides
  const size_t gsW = (size_t)NTn * KTx * X3_BTILE, gsU = (size_t)NTn * KTh * X3_BTILE;
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
#define XP_BLOAD(gg) do { XP_SEL(gg); \
    const uint4 *t0_ = p1_ ? Wz + (size_t)kx_ * X3_BTILE : Uz + (size_t)kh_ * X3_BTILE; const size_t gs_ = p1_ ? gsW : gsU; \
    XP_BLD(rb[0], t0_); XP_BLD(rb[1], t0_ + gs_); XP_BLD(rb[2], t0_ + 2 * gs_); } while (0)
#define XP_BST(buf, t, v) reinterpret_cast<X3_BVEC *>(&S.B[grp][buf][t][0][0])[gtid] = (v)
#define XP_BSTASH(buf) do { XP_BST(buf, 0, rb[0]); XP_BST(buf, 1, rb[1]); XP_BST(buf, 2, rb[2]); } while (0)
#define XP_PAIR(g, I2)                                                                                           \
    { XP_APTR(pa); x3_tile<RG, NP, 3, 0, 1, I2, 0>(q0, q1, S.B[grp][0], pa, lane, acc); }                         \
    XP_BSTASH(1); XP_BLOAD((g) + 2);                                                                             \
    __syncthreads();                                                                                             \
    { XP_APTR(pb); x3_tile<RG, NP, 3, 0, 1, I2, 0>(q2, q3, S.B[grp][1], pb, lane, acc); }                         \
    XP_BSTASH(0); XP_BLOAD((g) + 3);                                                                             \
    __syncthreads()
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
    XP_BLOAD(0);                                                                                                 \
  } while (0)
  // ... then the first weight tile into the group's LDS buffer 0, the second into registers, accumulators = biases
#define XP_PRO1() do {                                                                                           \
    XP_BSTASH(0); XP_BLOAD(1);                                                                                   \
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
#define XP_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

  // Every group runs the same straight sequence — prologue, then per tile a K phase and an E phase — group 1 one phase
  // behind group 0 (an idle phase before its first tile, one after group 0's last).  A group whose walk runs past the
  // end of the XCD's list clamps to the last tile and recomputes it (identical stores): no data-dependent control flow.
  fvec4 hq[8];
  XP_PRO0(0); XP_HOLOAD(); XP_PRO1(); XP_HOSTASH();
  __syncthreads();
#ifdef PN_X3_CLOCKS
  unsigned long long ck_k = 0, ck_e = 0; const unsigned long long ck_0 = __builtin_readcyclecounter();
#endif
  if (grp == 1) {
#pragma unroll 1
    for (int g = 0; g < TT; g++) XP_BAR();
  }
#pragma unroll 1
  for (int it = 0; it < n_it; it++) {
#ifdef PN_X3_CLOCKS
    const unsigned long long ck_a = __builtin_readcyclecounter();
#endif
    // ---- K phase: TT barriers -----------------------------------------------------------------------------------
#pragma unroll 1
    for (int g = 0; g < T1; g += 2) { XP_PAIR(g, 2); }
#pragma unroll 1
    for (int g = T1; g < TT; g += 2) { XP_PAIR(g, 3); }
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
      // chunk c = (row group c >> 3, output pair c & 7): stage A = arguments of z and r + their table reads,
      // B = z, r, candidate pre-activation, its argument + table read, C = candidate, blend.  Step s runs
      // C(s-2), B(s-1), A(s): every table value is read one barrier interval before it is used.
      X3Ts2 za[16], ra[16], ha[16];
      v2f zy[16], ry[16], hy[16], zv[16];
#pragma unroll
      for (int s = 0; s < 18; s++) {
        if (s >= 2) {
          const int c = s - 2, rg = c >> 3, i0 = 2 * (c & 7);
          const v2f hc = x3_ts_fin2(ha[c], hy[c]);
          const int hr = 32 * rg + (i0 & 3) + 8 * (i0 >> 2);
          const v2f hov = {Hl[hr * 32], Hl[(hr + 1) * 32]};
          v2f o = zv[c] * hov + (1.f - zv[c]) * hc;
          x3_pin(o);
          acc[rg][2][i0] = o.x; acc[rg][2][i0 + 1] = o.y;
        }
        if (s >= 1 && s <= 16) {
          const int c = s - 1, rg = c >> 3, i0 = 2 * (c & 7);
          const v2f z = .5f + .5f * x3_ts_fin2(za[c], zy[c]);
          const v2f r = .5f + .5f * x3_ts_fin2(ra[c], ry[c]);
          const v2f tmp = {acc[rg][3][i0], acc[rg][3][i0 + 1]}, hx = {acc[rg][2][i0], acc[rg][2][i0 + 1]};
          v2f hp = bh_e + tmp * r;
          hp = hp + hx;
          ha[c] = x3_ts_arg2(hp);
          hy[c] = v2f{S.tansig[ha[c].i.x], S.tansig[ha[c].i.y]};
          zv[c] = z;
          x3_pin(ha[c].x); x3_pin(ha[c].sb); x3_pin(zv[c]);
        }
        if (s <= 15) {
          const int c = s, rg = c >> 3, i0 = 2 * (c & 7);
          za[c] = x3_ts_arg2(.5f * v2f{acc[rg][0][i0], acc[rg][0][i0 + 1]});
          ra[c] = x3_ts_arg2(.5f * v2f{acc[rg][1][i0], acc[rg][1][i0 + 1]});
          zy[c] = v2f{S.tansig[za[c].i.x], S.tansig[za[c].i.y]};
          ry[c] = v2f{S.tansig[ra[c].i.x], S.tansig[ra[c].i.y]};
          x3_pin(za[c].x); x3_pin(za[c].sb); x3_pin(ra[c].x); x3_pin(ra[c].sb);
        }
        if (s == 17) XP_PRO0(it + 1);                          // the next tile's first operands (after the last use of the parked state)
        XP_BAR();
      }
#pragma unroll
      for (int rg = 0; rg < RG; rg++) {                        // steps 18..21
        float vo[16];
#pragma unroll
        for (int i = 0; i < 16; i++) vo[i] = acc[rg][2][i];
        x3_stage_write(T, vo, lane);
        x3_stage_store<NP>(T, 0, h_new, N, nt_e * 32, N, row0 + 32 * rg, n_rows, Sx, srow + 32 * rg, lane);
        if (rg == 0) XP_HOLOAD();                              // step 18: the next tile's previous state (every blend has read the slice)
        XP_BAR();
        x3_stage_store<NP>(T, 1, h_new, N, nt_e * 32, N, row0 + 32 * rg, n_rows, Sx, srow + 32 * rg, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        XP_BAR();
      }
      XP_PRO1();                                               // steps 22, 23
      XP_BAR();
      XP_HOSTASH();
      XP_BAR();
#pragma unroll 1
      for (int g = XP_EPI_STEPS; g < TT - 1; g++) XP_BAR();
      __syncthreads();                                         // the group's LDS weight buffer 0 is complete for its K phase
    }
#ifdef PN_X3_CLOCKS
    ck_e += __builtin_readcyclecounter() - ck_b;
#endif
  }
  if (grp == 0) {
#pragma unroll 1
    for (int g = 0; g < TT; g++) XP_BAR();
  }
#ifdef PN_X3_CLOCKS
  if (lane == 0 && gw == 0 && N == 512 && blockIdx.x < 256) {
    unsigned long long *t = pn_x3p_trace + ((size_t)blockIdx.x * 2 + grp) * 8;
    t[0] = ck_k; t[1] = ck_e; t[2] = __builtin_readcyclecounter() - ck_0; t[3] = n_it; t[4] = TT;
  }
#endif
#undef XP_SEL
#undef XP_APTR
#undef XP_BLD
#undef XP_BLOAD
#undef XP_BST
#undef XP_BSTASH
#undef XP_PAIR
#undef XP_PRO0
#undef XP_PRO1
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
  return n_rows >= 32768 ? 3 : 1;      // 3 = 64 rows per wave with the GRUs on the paired-phase kernel (pn_gru_x3p_kernel)
}

// A: panels carry the uint4* shadows of equally wide buffers (width = logical columns, a multiple of 32);
// out (fp32, optional) / outS (shadow of a buffer nts_out column tiles wide, optional)
void pn_launch_dense_x3(hipStream_t st, const PnSegs &A, const void *Wp, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, void *outS, int nts_out, int n_rows, int rg, int np) {
  if (rg == 3) rg = 2;                 // the paired-phase form exists for the GRUs only
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

void pn_launch_gru_x3(hipStream_t st, const PnSegs &X, const float *h_old, const void *h_oldS, const void *Wp,
                      const void *Up, const float *b, int N, int act, const float *tansig, float *h_new, void *h_newS,
                      int n_rows, int rg, int np) {
  const int tps = X.width[0] / 32, KTx = tps * X.n;
  const int NTn = N / 32;
  // rg 3: paired-phase kernel (one 8-wave block per CU, K loop of one wave group beside the epilogue of the other); it
  // is written for the tanh candidate and needs more K tiles than epilogue steps, otherwise the 64-rows-per-wave kernel
  if (rg == 3 && act == ACT_TANH && (KTx & 1) == 0 && (NTn & 1) == 0 && KTx + NTn > XP_EPI_STEPS + 1) {
    const int n_mtiles = (n_rows + 255) / 256;
    const int grid = 8 * (x3_cu_count() / 8);
#define XP_LAUNCH(NP_)                                                                                              \
    hipLaunchKernelGGL((pn_gru_x3p_kernel<NP_>), dim3(grid), dim3(512), 0, st, X, h_old, (const uint4 *)h_oldS,        \
                       (const uint4 *)Wp, (const uint4 *)Up, b, N, KTx, tps, tansig, h_new, (uint4 *)h_newS, n_rows, n_mtiles)
    if (np == 2) XP_LAUNCH(2); else XP_LAUNCH(1);
#undef XP_LAUNCH
    return;
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
