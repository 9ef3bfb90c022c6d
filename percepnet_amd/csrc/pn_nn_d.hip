// Direct-operand fp32 GRU kernels for gfx950: the GRU steps of large PN_NN_MFMA contexts (round 6; default from 24 576 streams).
//
// Same arithmetic as pn_gru_mfma_p_kernel (pn_nn.hip) — a GEMM on v_mfma_f32_32x32x2_f32 with the streams on the M axis, bias
// preload, one k-ascending chain per output element, the reset-after GRU's four accumulators (compute_gru, reference
// nnet.cpp:120-180; order of sgemv_accum, vec.h:102-135), the same gating epilogue (pn_gru_gate16) — so the results are
// bit-identical to the batch and small-batch families.  What differs is how the operands reach the matrix pipe.  In pn_nn.hip a
// wave owns 32 rows and both operands go through LDS: per 48 MFMAs (one 32-k tile) a wave issues 16 ds_read_b128, 11 ds_write,
// 7 global loads and one block barrier.  But a wave's activation rows are private to it — LDS only TRANSPOSES them into fragment
// order.  Here:
//   * the layers that feed a GRU step (conv2 through pn_dense_mfma_ps_kernel, the GRU steps themselves) also write their output
//     as a FRAGMENT-ORDER fp32 shadow
//       shadow[M tile of 128][column tile of 32][q 0..3][kh 0..1][row 0..127][4 floats: k = 32 ct + 8 q + 2 s + kh, s = 0..3]
//     (16 KB per (M tile, column tile); the same slab indexing j * 128 + row, j = 0..7, as the fp16 hi/lo shadows of
//     pn_nn_x3.hip, so the per-stream reset / active-set code is shared), and a wave loads its A fragments from it
//     straight into registers: lane (row r, k-half kh) takes the 16 bytes (q, kh, r) — 512 contiguous bytes per 32 lanes —
//     one float4 = four consecutive MFMA k-steps;
//   * a wave owns 64 rows (two 32-row groups): every weight fragment read from LDS feeds 8 MFMAs instead of 4;
//   * the weight tiles (packed in fragment order by pn_pack_weights) are copied linearly into a double-buffered LDS
//     image and read back lane-linear (conflict-free ds_read_b128).
// Per wave and 32-k tile: 96 MFMAs, 12 ds_read_b128, 3 ds_write_b128, 11 global loads, one barrier — per MFMA 2.7x fewer LDS
// reads, 7x fewer LDS writes and half the barriers of the batch kernel.  Block = 4 waves x 64 rows = 256 streams x the three gate
// tiles of one 32-neuron column tile.  Measured (profiles/r06_direct_operand_gru.log): the 512 -> 512 step 1.552 -> 1.523 ms at
// 65 536 streams; what is left above the matrix-pipe time is the activation loads (3.8 %: VMEM issued by the wave that feeds the
// pipe) and the gating epilogue (2.7 %).  The dense layers were built in this form too and gained nothing: they stay in pn_nn.hip.
#include "pn_nn_common.h"
#include <stdlib.h>

// Timing ablations (tools only; results WRONG): 1 no A refills, 2 no weight staging, 4 no K-loop barriers, 8 no gating epilogue,
// 16 no weight-fragment reads
#ifndef PN_D_ABL
#define PN_D_ABL 0
#endif
#ifndef PN_D_REFILL_IDLE
#define PN_D_REFILL_IDLE 1              // 64-row waves: a tile's activation loads go to the register set the current tile does not read
#endif

#if PN_D_ABL & 4
#define D_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define D_SYNC() __syncthreads()
#endif

#define D_CHUNK PN_SHADOW_CHUNK

struct DShared {
  fvec4 B[2][4][256];                  // [buffer][column tile][float4 number q * 64 + lane of the packed tile]: 2 x 16 KB
  float tansig[208];
};
static_assert(4 * 32 * PN_TLD * sizeof(float) <= sizeof(fvec4) * 2 * 4 * 256, "the epilogue stage aliases the weight buffers");

// a wave-uniform pointer, pinned into scalar registers (folds away when the compiler already keeps it there)
// (rebuilt as a GLOBAL-address-space pointer: through a plain integer round trip the loads become flat_load)
typedef const __attribute__((address_space(1))) char *d_gptr;
__device__ __forceinline__ d_gptr d_uniform(const uint4 *p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (d_gptr)(((unsigned long long)hi << 32) | lo);
}

template <int RG> struct DA { fvec4 v[4][RG]; };         // the A fragments of one 32-k tile: [q][row group of 32]

// slab q of the tile whose chunk starts at pt (wave-uniform); lb = the lane's byte offset inside a slab pair (slab kh, row
// srow + r).  Uniform base + 32-bit per-lane offset: hipcc emits `global_load_dwordx4 v, v_off, s[base:base+1] offset:512 rg` —
// the tile advance lives in scalar registers, no 64-bit VALU address arithmetic between the MFMAs
template <int RG>
__device__ __forceinline__ void d_load_A(DA<RG> &a, d_gptr pt, unsigned lb, int q) {
  typedef const __attribute__((address_space(1))) fvec4 *gv4;
  d_gptr base = pt + q * 4096;
#pragma unroll
  for (int rg = 0; rg < RG; rg++) a.v[q][rg] = *(gv4)(base + (lb + 512u * rg));
}

// One 32-k tile: 4 NT groups (q, column tile t) of 4 RG MFMAs — the four k-steps of slab q for every row group, into
// acc[rg][IDX[t]].  The weight fragment of group i + 1 is read while the MFMAs of group i run; slab q of the A registers is
// refilled from the tile at pf (two tiles ahead) as soon as its last MFMA has issued; mid(i) runs after group i (the
// caller's weight staging rides there, in the shadow of the MFMAs, instead of after the tile).
template <int RG, int NT, int I0, int I1, int I2, int I3, class Mid>
__device__ __forceinline__ void d_tile(DA<RG> &a, DA<RG> &fill, const fvec4 (*Bs)[256], d_gptr pf, unsigned lb, int lane,
                                       floatx16 (&acc)[RG][4], Mid &&mid) {
  constexpr int IDX[4] = {I0, I1, I2, I3};
  fvec4 f0 = Bs[0][lane], f1;
#pragma unroll
  for (int i = 0; i < 4 * NT; i++) {
    const int q = i / NT, t = i % NT;
    fvec4 &cur = (i & 1) ? f1 : f0, &nxt = (i & 1) ? f0 : f1;
#if !(PN_D_ABL & 16)
    if (i + 1 < 4 * NT) nxt = Bs[(i + 1) % NT][((i + 1) / NT) * 64 + lane];
#else
    nxt = cur;
#endif
    __builtin_amdgcn_sched_barrier(0);                   // keep the read ahead of the MFMAs it overlaps (the scheduler sinks it)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int rg = 0; rg < RG; rg++)
        acc[rg][IDX[t]] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[q][rg][c], cur[c], acc[rg][IDX[t]], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#if !(PN_D_ABL & 1)
    if (t == NT - 1) {
      d_load_A<RG>(fill, pf, lb, q);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
    mid(i);
  }
}

#define D_BLOAD(dst, src) (dst) = reinterpret_cast<const fvec4 *>(src)[tid]
#define D_BSTASH(buf, t, v) S.B[buf][t][tid] = (v)

// ---- GRU step (reset-after, nnet.cpp:122-180): acc z, r, hx (W_h x), tmp (b_rh + U_h h) -------------------------------
template <int RG>
__global__ __launch_bounds__(NN_THREADS, 4 - RG) void pn_gru_d_kernel(
    PnSegs X, const float *__restrict__ h_old, const uint4 *__restrict__ h_oldS, const float *__restrict__ Wp,
    const float *__restrict__ Up, const float *__restrict__ b, int N, int KTx, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ h_new, uint4 *__restrict__ h_newS, int n_rows, int n_mtiles) {
  __shared__ DShared S;
  const int NTn = N >> 5;
  int mt, nt;
  if (!pn_tile_of_block(n_mtiles, NTn, mt, nt)) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int KTh = NTn, T1 = KTx, TT = KTx + KTh;
  const int col = nt * 32 + (lane & 31);
  const float ts_v = tansig[tid < 201 ? tid : 200];
  floatx16 acc[RG][4];
  {
    float bz = b[col]; bz += b[3 * N + col];             // nnet.cpp:135-141
    float br = b[N + col]; br += b[4 * N + col];         // 147-153
    const float bt = b[5 * N + col];                     // 164
#pragma unroll
    for (int rg = 0; rg < RG; rg++)
#pragma unroll
      for (int i = 0; i < 16; i++) { acc[rg][0][i] = bz; acc[rg][1][i] = br; acc[rg][2][i] = 0.f; acc[rg][3][i] = bt; }
  }
  constexpr int XMB = 128 * RG;                          // rows per block
  const int mt128 = (mt * XMB + 32 * RG * wave) >> 7, srow = (32 * RG * wave) & 127;
  const unsigned lb = (unsigned)(((lane >> 5) * 128 + srow + (lane & 31)) * 16);
  const float *Wz = Wp + (size_t)(0 * NTn + nt) * KTx * 1024, *Wr = Wp + (size_t)(1 * NTn + nt) * KTx * 1024,
              *Wh = Wp + (size_t)(2 * NTn + nt) * KTx * 1024;
  const float *Uz = Up + (size_t)(0 * NTn + nt) * KTh * 1024, *Ur = Up + (size_t)(1 * NTn + nt) * KTh * 1024,
              *Uh = Up + (size_t)(2 * NTn + nt) * KTh * 1024;
  PN_PANEL_LOCALS(X);
  (void)pld;
  // A tiles are asked for strictly in order: a cursor over (x panels, then the recurrent operand) instead of a division
  // per tile; past the last tile it keeps returning the last one (loaded, never used)
  int c_sg = 0, c_kt = 0, c_g = 0;
  const uint4 *c_last = nullptr;
#define DG_APTR(pt) \
    d_gptr pt; { if (c_g < T1) { \
        c_last = reinterpret_cast<const uint4 *>(pn_seg_ptr(PN_PANEL_PASS, c_sg)) + ((size_t)mt128 * tps + c_kt) * D_CHUNK; \
        c_kt++; if (c_kt == tps) { c_kt = 0; c_sg++; } \
      } else if (c_g < TT) { c_last = h_oldS + ((size_t)mt128 * NTn + (c_g - T1)) * D_CHUNK; } \
      c_g++; pt = d_uniform(c_last); }
#define DG_BLOAD(gg) do { int g_ = (gg); g_ = g_ < TT ? g_ : TT - 1; const bool p1_ = g_ < T1; \
    const size_t bo_ = (size_t)(p1_ ? g_ : g_ - T1) * 1024; \
    D_BLOAD(rb[0], (p1_ ? Wz : Uz) + bo_); D_BLOAD(rb[1], (p1_ ? Wr : Ur) + bo_); D_BLOAD(rb[2], (p1_ ? Wh : Uh) + bo_); } while (0)
#define DG_BSTASH(buf) do { D_BSTASH(buf, 0, rb[0]); D_BSTASH(buf, 1, rb[1]); D_BSTASH(buf, 2, rb[2]); } while (0)
  // Where the activation loads issued during a tile land.  64-row waves: in the register set the current tile does NOT read — tile
  // g + 1, one tile (>= 6144 matrix-pipe cycles) ahead; no load ever targets a register an in-flight MFMA still reads: gru512 -0.4 %
  // (profiles/r06_direct_operand_gru.log G).  32-row waves (a tile is half as long, three waves per SIMD): the slab just consumed is
  // refilled with tile g + 2, two tiles ahead, as in pn_nn_x3.hip.  (A raised wave priority for the K loop: +0.8 %, not kept.)
  constexpr bool IDLE = PN_D_REFILL_IDLE && RG == 2;
#define DG_FILL(cur, other) (IDLE ? other : cur)
#define DG_PAIR(g, I2)                                                                                             \
    { DG_APTR(pa); d_tile<RG, 3, 0, 1, I2, 0>(qa, DG_FILL(qa, qb), S.B[0], pa, lb, lane, acc, [&](int i) { if (!(PN_D_ABL & 2)) { if (i == 1) DG_BSTASH(1); if (i == 3) DG_BLOAD((g) + 2); } }); } \
    D_SYNC();                                                                                                      \
    { DG_APTR(pb); d_tile<RG, 3, 0, 1, I2, 0>(qb, DG_FILL(qb, qa), S.B[1], pb, lb, lane, acc, [&](int i) { if (!(PN_D_ABL & 2)) { if (i == 1) DG_BSTASH(0); if (i == 3) DG_BLOAD((g) + 3); } }); } \
    D_SYNC()
  DA<RG> qa, qb;
  fvec4 rb[3];
  { DG_APTR(p0); _Pragma("unroll") for (int q = 0; q < 4; q++) d_load_A<RG>(qa, p0, lb, q); }
  if (!IDLE) { DG_APTR(p1); _Pragma("unroll") for (int q = 0; q < 4; q++) d_load_A<RG>(qb, p1, lb, q); }
  DG_BLOAD(0); DG_BSTASH(0); DG_BLOAD(1);
  if (tid < 201) S.tansig[tid] = ts_v;
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < T1; g += 2) { DG_PAIR(g, 2); }
#pragma unroll 1
  for (int g = T1; g < TT; g += 2) { DG_PAIR(g, 3); }

#undef DG_PAIR
#undef DG_BSTASH
#undef DG_BLOAD
#undef DG_APTR
  // gates, candidate, blend (nnet.cpp:144,156,161-179)
  {
    const float bh = b[2 * N + col];
    float *T = reinterpret_cast<float *>(&S.B[0][0][0]) + wave * 32 * PN_TLD;
    uint4 *Sx = h_newS ? h_newS + ((size_t)mt128 * NTn + nt) * D_CHUNK : nullptr;
    // previous state for the blend: the loads of all row groups in flight before any activation arithmetic (the state
    // buffers are allocated with their row count rounded up to the tile: rows past n_rows are readable, only stores are guarded)
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
#if PN_D_ABL & 8
      _Pragma("unroll") for (int i = 0; i < 16; i++) v[i] = acc[rg][0][i] + acc[rg][1][i] + acc[rg][2][i] + acc[rg][3][i] + ho[rg][i] + bh;
#else
      pn_gru_gate16(acc[rg][0], acc[rg][1], acc[rg][2], acc[rg][3], ho[rg], bh, act, S.tansig, v);
#endif
      pn_store_tile_frag(T, v, h_new, N, nt * 32, N, grow0, n_rows, Sx, srow + 32 * rg, lane);
    }
  }
}

// ---- fp32 rows -> fragment-order fp32 shadow (the first layer's output; RNN state loaded from the host) -----------
// one thread per (row, slab pair q): reads 32 bytes (k = 8q .. 8q + 7 of a column tile), writes the kh = 0 and kh = 1 entries
__global__ __launch_bounds__(256) void pn_split_d_kernel(const float *__restrict__ src, int ld, int width, uint4 *__restrict__ S,
                                                         int n_rows_padded) {
  const int kgs = width >> 3;                                    // 8-column groups per row
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = idx / kgs;
  const int kgi = (int)(idx - row * kgs);
  if (row >= (size_t)n_rows_padded) return;
  const fvec4 a = *reinterpret_cast<const fvec4 *>(src + row * ld + 8 * kgi);
  const fvec4 c = *reinterpret_cast<const fvec4 *>(src + row * ld + 8 * kgi + 4);
  const fvec4 e = {a.x, a.z, c.x, c.z}, o = {a.y, a.w, c.y, c.w};
  uint4 *chunk = S + ((row >> 7) * (width >> 5) + (kgi >> 2)) * D_CHUNK;
  chunk[(2 * (kgi & 3)) * 128 + (row & 127)] = __builtin_bit_cast(uint4, e);
  chunk[(2 * (kgi & 3) + 1) * 128 + (row & 127)] = __builtin_bit_cast(uint4, o);
}

// ---- launchers -----------------------------------------------------------------------------------------------------
// Which PN_NN_MFMA contexts run their GRU steps here (read at every context creation; tests switch it between contexts):
// PERCEPNET_NN_DIRECT=0|1 overrides the batch-size rule.  Measured against the batch family, same box, default chain rule
// (profiles/r06_direct_operand_gru.log): frame time -0.6 % at 24 576 streams, -1.1 % at 32 768, -1.2 % at 61 440 / 66 560, -1.4 % at
// 69 632 (the 512 -> 512 step at 65 536: 1.552 -> 1.523 ms), even at 16 384, +1.4 % at 8192 (too few blocks per launch).
#ifndef PN_DIRECT_ROWS
#define PN_DIRECT_ROWS 24576
#endif
int pn_direct_for(int n_rows) {
  const char *e = getenv("PERCEPNET_NN_DIRECT");
  if (e) return atoi(e) ? 1 : 0;
  return n_rows >= PN_DIRECT_ROWS;
}
int pn_direct_rg_for(int n_rows) {
  const char *e = getenv("PERCEPNET_NN_DIRECT_RG");
  const int env = e ? atoi(e) : 0;
  if (env == 1 || env == 2) return env;
  return n_rows >= 32768 ? 2 : 1;
}
// X panels / h_oldS / h_newS: the uint4* fragment-order fp32 shadows; Wp / Up: the fp32 packed tiles of pn_pack_weights.
// rg: row groups of 32 per wave (2: 256-row blocks, two per CU; 1: 128-row blocks, three)
int pn_launch_gru_d(hipStream_t st, const PnSegs &X, const float *h_old, const void *h_oldS, const float *Wp,
                    const float *Up, const float *b, int N, int act, const float *tansig, float *h_new, void *h_newS,
                    int n_rows, int rg) {
  const int tps = X.width[0] / 32, KTx = tps * X.n;
  const int NTn = N / 32;
  if (pn_check_gru_geometry("pn_launch_gru_d", X.n, X.width, N)) return -1;   // k-tiles are consumed in pairs (x: 16 / 32, h: 16 / 4)
  const int n_mtiles = (n_rows + 128 * rg - 1) / (128 * rg);
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
#define DG_LAUNCH(RG_)                                                                                             \
  hipLaunchKernelGGL((pn_gru_d_kernel<RG_>), dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, (const uint4 *)h_oldS, Wp, Up, \
                     b, N, KTx, tps, act, tansig, h_new, (uint4 *)h_newS, n_rows, n_mtiles)
  if (rg == 2) DG_LAUNCH(2); else DG_LAUNCH(1);
#undef DG_LAUNCH
  return 0;
}

// refuses (-1, pn_set_error, nothing launched) a width that is not whole groups of 8 columns
int pn_launch_split_d(hipStream_t st, const float *src, int ld, int width, void *S, int n_rows_padded) {
  if (width < 8 || (width & 7) || ld < width || n_rows_padded < 1 || !src || !S) {
    pn_set_error("pn_launch_split_d: width %d (whole groups of 8), row stride %d, %d rows", width, ld, n_rows_padded);
    return -1;
  }
  const size_t n = (size_t)n_rows_padded * (width >> 3);
  hipLaunchKernelGGL(pn_split_d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, width, (uint4 *)S, n_rows_padded);
  return 0;
}
