// Gain-network kernels (compute_rnn, reference rnn.cpp:42-81) batched over streams for gfx950.
//
// MFMA path (PN_NN_MFMA): every layer is a GEMM  [B streams] x [K inputs] x [N neurons]  on
// v_mfma_f32_32x32x2_f32 (exact fp32, a k-ascending fmaf chain per output element), streams on
// the M axis, with bias preload and the table activation / GRU gating fused in the epilogue.
// The summation order is the reference's (sgemv_accum, nnet.cpp:59-72 / vec.h:102-135): start
// from the bias, add input contributions k = 0..K-1, then recurrent ones; the reset-after GRU
// (compute_gru, nnet.cpp:120-180) keeps four accumulators per output: z and r (bias -> W x -> U h, the reference's
// chains), tmp = b_rh + U_h h, and hx = W_h x summed as its OWN k-ascending chain from 0 in the same sweep over x;
// the candidate is b_h + tmp * r + hx.  The reference adds the W_h x products onto (b_h + tmp * r) one by one
// (nnet.cpp:166-167): same terms, one different association — the single documented deviation from the
// reference's order.  Other than that: fused instead of separate rounding of each multiply-add.
//
// Tiling: 256-thread blocks (4 waves); a block owns 128 streams x (NT x 32) output columns,
// wave w owns rows [32w, 32w+32).  Per K-tile of 32 the A tile (activations, row-major in HBM)
// and the B tiles (weights, pre-packed at context creation) are staged in LDS in a
// k-interleaved order [row][q][kh][s]  (k = 8q + 2s + kh) so that one ds_read_b128 feeds four
// consecutive MFMA k-steps of a lane ((lane>>5) = kh); rows are padded to 36 floats, which
// makes the 16-lane b128 groups bank-conflict-free.  Blocks are numbered XCD-major so the
// column tiles that share an activation panel run on one XCD's L2.
//
// STRICT path (PN_NN_STRICT): one lane per (stream, neuron), separate v_mul/v_add in the
// reference's order (file compiled with -ffp-contract=off) — bit-identical to the CPU reference.
#include "pn_nn_common.h"
#include <stdlib.h>

#define BK 32
#define LDT 36            // padded LDS row stride (floats)
// =============================== STRICT kernels ==================================================
// W in the reference layout [K][ncols] (nnet_data.h); thread = (stream, neuron); the stream index is folded into
// grid.x (block = nbx * stream + neuron block) because grid.y stops at 65535 and a batch may be larger.
__global__ void pn_dense_strict_kernel(PnSegs A, const float *__restrict__ W, const float *__restrict__ bias,
                                       int N, int act, const float *__restrict__ tansig, float *__restrict__ out,
                                       int ldo, int nbx) {
  const int m = blockIdx.x / nbx, n = (blockIdx.x - m * nbx) * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc = bias[n];
  int koff = 0;
  for (int sg = 0; sg < A.n; sg++) {
    const float *x = A.p[sg] + (size_t)m * A.ld[sg];
    for (int k = 0; k < A.width[sg]; k++) acc = acc + W[(size_t)(koff + k) * N + n] * x[k];
    koff += A.width[sg];
  }
  out[(size_t)m * ldo + n] = pn_act(acc, act, tansig);
}

__global__ void pn_gru_strict_kernel(PnSegs X, const float *__restrict__ h_old, const float *__restrict__ W,
                                     const float *__restrict__ U, const float *__restrict__ b, int N, int act,
                                     const float *__restrict__ tansig, float *__restrict__ h_new, int nbx) {
  const int m = blockIdx.x / nbx, n = (blockIdx.x - m * nbx) * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int st = 3 * N;
  const float *h = h_old + (size_t)m * N;
  float z = b[n]; z += b[3 * N + n];
  float r = b[N + n]; r += b[4 * N + n];
  int koff = 0;
  for (int sg = 0; sg < X.n; sg++) {
    const float *x = X.p[sg] + (size_t)m * X.ld[sg];
    for (int k = 0; k < X.width[sg]; k++) z = z + W[(size_t)(koff + k) * st + n] * x[k];
    koff += X.width[sg];
  }
  for (int k = 0; k < N; k++) z = z + U[(size_t)k * st + n] * h[k];
  z = pn_sigmoid(z, tansig);
  koff = 0;
  for (int sg = 0; sg < X.n; sg++) {
    const float *x = X.p[sg] + (size_t)m * X.ld[sg];
    for (int k = 0; k < X.width[sg]; k++) r = r + W[(size_t)(koff + k) * st + N + n] * x[k];
    koff += X.width[sg];
  }
  for (int k = 0; k < N; k++) r = r + U[(size_t)k * st + N + n] * h[k];
  r = pn_sigmoid(r, tansig);
  float hh = b[2 * N + n];
  float tmp = b[5 * N + n];
  for (int k = 0; k < N; k++) tmp = tmp + U[(size_t)k * st + 2 * N + n] * h[k];
  hh += tmp * r;
  koff = 0;
  for (int sg = 0; sg < X.n; sg++) {
    const float *x = X.p[sg] + (size_t)m * X.ld[sg];
    for (int k = 0; k < X.width[sg]; k++) hh = hh + W[(size_t)(koff + k) * st + 2 * N + n] * x[k];
    koff += X.width[sg];
  }
  hh = pn_act(hh, act, tansig);
  h_new[(size_t)m * N + n] = z * h[n] + (1 - z) * hh;
}

// =============================== MFMA kernels ====================================================
struct NnShared {
  float A[2][BM][LDT];       // 2 x 18432 B   double-buffered K-tiles
  float B[2][4 * 32][LDT];   // 2 x 18432 B   (up to 4 column tiles: dense NT<=4, GRU 3 gates)
  float tansig[208];
#ifdef PN_NN_LDS_PAD
  char pad[PN_NN_LDS_PAD];   // experiment (tools/gpu_overlap.sh): > 7.3 KB caps these kernels at ONE block per CU, leaving
                             // 81 KB of LDS and 264 registers per lane for a DSP block of another half-batch
#endif
};

// ---- software-pipelined staging: global -> registers (issued before the MFMAs of the current
// K-tile, so L2/HBM latency hides under them) -> LDS (written after them, into the other buffer).
// A tile: 128 rows x 32 k of a row-major activation panel = 4 float4 per thread.  Every
// activation buffer is allocated with its row count rounded up to 128 and its width to 32
// (pn_context.cpp), so loads need no predication; only the final stores are row-guarded.
__device__ __forceinline__ void pn_load_A(float4 (&ra)[4], const float *__restrict__ p, int ld, int k0, int m0) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + NN_THREADS * it;
    const int row = idx >> 3, c = idx & 7;        // 8 float4 per row
    ra[it] = *reinterpret_cast<const float4 *>(p + (size_t)(m0 + row) * ld + k0 + 4 * c);
  }
}
// k-interleave while writing: k_local = 4c + {0,1,2,3} -> q = c>>1, (kh,s) = (0,2(c&1)), (1,2(c&1)),
// (0,2(c&1)+1), (1,2(c&1)+1); position in the row = q*8 + kh*4 + s
__device__ __forceinline__ void pn_store_A(float (*As)[LDT], const float4 (&ra)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + NN_THREADS * it;
    const int row = idx >> 3, c = idx & 7;
    float *dst = &As[row][(c >> 1) * 8 + 2 * (c & 1)];
    *reinterpret_cast<float2 *>(dst) = make_float2(ra[it].x, ra[it].z);
    *reinterpret_cast<float2 *>(dst + 4) = make_float2(ra[it].y, ra[it].w);
  }
}
// one packed 32(col) x 32(k) weight tile = 1024 contiguous floats, already k-interleaved
__device__ __forceinline__ float4 pn_load_B(const float *__restrict__ tile) {
  const int tid = threadIdx.x;
  return *reinterpret_cast<const float4 *>(tile + (tid >> 3) * 32 + 4 * (tid & 7));
}
__device__ __forceinline__ void pn_store_B(float (*Bs)[LDT], const float4 &v) {
  // float4 #tid of a packed tile = chunk (q, kh) = tid >> 5 of column tid & 31 (pn_pack_weights) -> row-major padded LDS
  // image Bs[col][q*8 + kh*4 ..]; 8 consecutive lanes hit rows 36 floats apart = 8 distinct bank quads: conflict-free
  const int tid = threadIdx.x;
  *reinterpret_cast<float4 *>(&Bs[tid & 31][4 * (tid >> 5)]) = v;
}

// Register set holding one prefetched K-tile (A: 4 float4, B: up to NB float4 per thread)
template <int NB> struct PnTileRegs { float4 a[4]; float4 b[NB]; };

template <int NT>
__device__ __forceinline__ void pn_dense_fetch(PnTileRegs<NT> &R, int g, int KT, int tps, PN_PANEL_ARGS,
                                               const float *__restrict__ wbase, int m0) {
  g = g < KT ? g : KT - 1;                     // past-the-end prefetches re-read the last tile (unused)
  const int sg = g / tps, k0 = (g - sg * tps) * BK;
  pn_load_A(R.a, pn_seg_ptr(PN_PANEL_PASS, sg), pld, k0, m0);
#pragma unroll
  for (int t = 0; t < NT; t++) R.b[t] = pn_load_B(wbase + ((size_t)t * KT + g) * 1024);
}
template <int NT>
__device__ __forceinline__ void pn_tile_stash(float (*As)[LDT], float (*Bs)[LDT], const PnTileRegs<NT> &R) {
  pn_store_A(As, R.a);
#pragma unroll
  for (int t = 0; t < NT; t++) pn_store_B(&Bs[32 * t], R.b[t]);
}

// =============================== half-tile software pipeline ========================================
// The K loop above leaves two kinds of bubbles per K-tile in every wave: the two batches of LDS operand
// reads are waited for right before the MFMAs that use them, and the global prefetch / LDS stash /
// barrier sit between MFMA blocks; two co-resident blocks run in lock-step, so the second wave of the
// SIMD does not fill them (measured: operands hot in cache change nothing, removing the staging
// instructions gains 17 %).  The kernels below issue every non-MFMA instruction in the shadow of an
// MFMA: per K-tile g ("interval", between two block barriers)
//     read half 0 of tile g from LDS            | interleaved with the MFMAs of half 1 of tile g-1
//     global loads of tile g+2 -> registers     |   (one piece after each 3-MFMA k-step)
//     read half 1 of tile g from LDS            | interleaved with the MFMAs of half 0 of tile g
//     registers of tile g+1 -> the other buffer |
//     barrier
// LDS buffer discipline: tile g is written during interval g-1 and read only during interval g.
// The order is pinned with sched_barrier(0) between pieces; numerics are unchanged (same MFMAs in the
// same k order per accumulator).
template <int NB> struct PnHalfOps { float4 a[2]; float4 b[NB][2]; };
#define PN_SB() __builtin_amdgcn_sched_barrier(0)

template <int NB, int HF, int QQ>
__device__ __forceinline__ void pn_lds_read_q(PnHalfOps<NB> &o, const float (*As)[LDT], const float (*Bs)[LDT],
                                              int wave, int lane) {
  const int r = lane & 31, kh = lane >> 5;
  constexpr int q = 2 * HF + QQ;
  o.a[QQ] = *reinterpret_cast<const float4 *>(&As[32 * wave + r][q * 8 + kh * 4]);
#pragma unroll
  for (int t = 0; t < NB; t++) o.b[t][QQ] = *reinterpret_cast<const float4 *>(&Bs[32 * t + r][q * 8 + kh * 4]);
}
// the same with the weight tiles in FRAGMENT order (the global packing, copied verbatim by LDS-DMA): tile t at Bf + 1024 t,
// float4 number q*64 + lane = the four k-steps (q, kh = lane >> 5) of column lane & 31 — consecutive lanes, consecutive 16 bytes
template <int NB, int HF, int QQ>
__device__ __forceinline__ void pn_lds_read_qf(PnHalfOps<NB> &o, const float (*As)[LDT], const float *Bf, int wave, int lane) {
  const int r = lane & 31, kh = lane >> 5;
  constexpr int q = 2 * HF + QQ;
  o.a[QQ] = *reinterpret_cast<const float4 *>(&As[32 * wave + r][q * 8 + kh * 4]);
#pragma unroll
  for (int t = 0; t < NB; t++) o.b[t][QQ] = *reinterpret_cast<const float4 *>(Bf + 1024 * t + (q * 64 + lane) * 4);
}
// one k-step (2 k values) of a GRU half tile: 3 MFMAs
#define PN_G3(o, QQ, c, I0, I1, I2) do {                                                                   \
    acc[I0] = __builtin_amdgcn_mfma_f32_32x32x2f32((o).a[QQ].c, (o).b[0][QQ].c, acc[I0], 0, 0, 0);         \
    acc[I1] = __builtin_amdgcn_mfma_f32_32x32x2f32((o).a[QQ].c, (o).b[1][QQ].c, acc[I1], 0, 0, 0);         \
    acc[I2] = __builtin_amdgcn_mfma_f32_32x32x2f32((o).a[QQ].c, (o).b[2][QQ].c, acc[I2], 0, 0, 0);         \
    PN_SB(); } while (0)
// one A-panel float4 of the staging set (piece `it` of pn_load_A / pn_store_A)
__device__ __forceinline__ float4 pn_load_A1(const float *__restrict__ p, int ld, int k0, int m0, int it) {
  const int idx = threadIdx.x + NN_THREADS * it;
  return *reinterpret_cast<const float4 *>(p + (size_t)(m0 + (idx >> 3)) * ld + k0 + 4 * (idx & 7));
}
// uniform (SGPR) base + 32-bit per-lane byte offset: hipcc emits `global_load_dwordx4 v, v_off, s[base:base+1]`, no
// 64-bit VALU address arithmetic in the K loop (the K-tile advance lives in the scalar base)
__device__ __forceinline__ float4 pn_load_so(const float *__restrict__ ubase, unsigned off_bytes) {
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(ubase) + off_bytes);
}
__device__ __forceinline__ void pn_store_A1(float (*As)[LDT], const float4 &v, int it) {
  const int idx = threadIdx.x + NN_THREADS * it;
  const int row = idx >> 3, c = idx & 7;
  float *dst = &As[row][(c >> 1) * 8 + 2 * (c & 1)];
  *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.z);
  *reinterpret_cast<float2 *>(dst + 4) = make_float2(v.y, v.w);
}

// Tuning aid (-DPN_NN_CLOCKS): per-block shader-clock (s_memtime) and constant 100 MHz (s_memrealtime) ticks of the
// pipelined GRU kernel, summed over blocks -> effective shader clock under this load, cycles per block.
#ifdef PN_NN_CLOCKS
__device__ unsigned long long pn_nn_clk[4];
__device__ unsigned long long pn_nn_trace[8192 * 4];     // per block of the last N=512 launch: start, end (100 MHz ticks), HW_ID, XCC_ID
extern "C" PN_EXPORT int pn_nn_trace_read(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pn_nn_trace), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : -1;
}
extern "C" PN_EXPORT int pn_nn_clocks_read(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pn_nn_clk), sizeof(unsigned long long) * 4) != hipSuccess) return -1;
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pn_nn_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 4;
}
#endif

__global__ __launch_bounds__(NN_THREADS) void pn_gru_mfma_p_kernel(
    PnSegs X, const float *__restrict__ h_old, const float *__restrict__ Wp, const float *__restrict__ Up,
    const float *__restrict__ b, int N, int KTx, int tps, int act, const float *__restrict__ tansig,
    float *__restrict__ h_new, int n_rows, int n_mtiles) {
  __shared__ NnShared S;
#ifdef PN_NN_SETPRIO
  __builtin_amdgcn_s_setprio(PN_NN_SETPRIO);   // experiment: the MFMA waves win every issue arbitration against co-resident DSP waves
#endif
  const int NTn = N >> 5;
  int mt, nt;
  if (!pn_tile_of_block(n_mtiles, NTn, mt, nt)) return;
#ifdef PN_NN_CLOCKS
  const long long c0_ = __builtin_readcyclecounter(), r0_ = wall_clock64();
#endif
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = mt * BM, KTh = N >> 5;
  const int T1 = KTx, TT = KTx + KTh;
  const int col = nt * 32 + (lane & 31);
  for (int i = tid; i < 201; i += (int)blockDim.x) S.tansig[i] = tansig[i];   // 201 entries whatever the block size (a 192-thread block once left 192..200 unstaged)

  floatx16 acc[4];
  {
    float bz = b[col]; bz += b[3 * N + col];    // nnet.cpp:135-141
    float br = b[N + col]; br += b[4 * N + col];// 147-153
    const float bt = b[5 * N + col];            // 164
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[0][i] = bz; acc[1][i] = br; acc[2][i] = 0.f; acc[3][i] = bt; }
  }
  const float *Wz = Wp + (size_t)(0 * NTn + nt) * KTx * 1024, *Wr = Wp + (size_t)(1 * NTn + nt) * KTx * 1024,
              *Wh = Wp + (size_t)(2 * NTn + nt) * KTx * 1024;
  const float *Uz = Up + (size_t)(0 * NTn + nt) * KTh * 1024, *Ur = Up + (size_t)(1 * NTn + nt) * KTh * 1024,
              *Uh = Up + (size_t)(2 * NTn + nt) * KTh * 1024;
  PN_PANEL_LOCALS(X);
  PnTileRegs<3> R0, R1;
  PnHalfOps<3> op0, op1;
  // per-lane byte offsets of the four A float4 (x panels: row stride pld; h_old: row stride N) and of the B float4
  unsigned aox[4], aoh[4];
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + NN_THREADS * it;
    aox[it] = (unsigned)(((idx >> 3) * pld + 4 * (idx & 7)) * 4);
    aoh[it] = (unsigned)(((idx >> 3) * N + 4 * (idx & 7)) * 4);
  }
  const unsigned bo4 = (unsigned)(tid * 16);
  // scalar operand selection for tile gg (clamped past the end: re-reads the last tile, never used)
#define GP_SEL(gg)                                                                                        \
    int g_ = (gg); g_ = g_ < TT ? g_ : TT - 1;                                                             \
    const bool p1_ = g_ < T1;                                                                              \
    const int kx_ = p1_ ? g_ : 0, kh_ = p1_ ? 0 : g_ - T1;                                                 \
    const int sg_ = kx_ / tps, k0_ = (kx_ - sg_ * tps) * BK;                                               \
    const float *ap_ = (p1_ ? pn_seg_ptr(PN_PANEL_PASS, sg_) + (size_t)m0 * pld + k0_ : h_old + (size_t)m0 * N + kh_ * BK); \
    const size_t bo_ = (size_t)(p1_ ? kx_ : kh_) * 1024;                                         \
    const float *bz_ = (p1_ ? Wz : Uz) + bo_, *br_ = (p1_ ? Wr : Ur) + bo_, *bh_ = (p1_ ? Wh : Uh) + bo_
#define GP_LA(it) pn_load_so(ap_, p1_ ? aox[it] : aoh[it])
#ifdef PN_NN_BDMA
  // Weight tiles by LDS-DMA (global_load_lds_dwordx4): tile g+1 is copied during interval g straight into the other B
  // buffer in its packed fragment order — no staging registers, no ds_write for B; each wave moves one 1 KB quarter of
  // each of the three gate tiles.  The fragment-order images alias S.B (2 x 3072 of its 9216 floats).
  float *const Bf0 = &S.B[0][0][0], *const Bf1 = Bf0 + 3 * 1024;
#define GP_BF(BUF) ((BUF) ? Bf1 : Bf0)
#define GP_SELB(gg)                                                                                       \
    int gb_ = (gg); gb_ = gb_ < TT ? gb_ : TT - 1;                                                         \
    const bool pb_ = gb_ < T1;                                                                             \
    const size_t bob_ = (size_t)(pb_ ? gb_ : gb_ - T1) * 1024;                                             \
    const float *dz_ = (pb_ ? Wz : Uz) + bob_, *dr_ = (pb_ ? Wr : Ur) + bob_, *dh_ = (pb_ ? Wh : Uh) + bob_
#define GP_DMA(src, BUF, t) __builtin_amdgcn_global_load_lds(reinterpret_cast<const char *>(src) + bo4, GP_BF(BUF) + 1024 * (t) + 256 * wave, 16, 0, 0)
#endif
#define GP_FETCH_ALL(R, gg) do { GP_SEL(gg);                                                               \
    (R).a[0] = GP_LA(0); (R).a[1] = GP_LA(1); (R).a[2] = GP_LA(2); (R).a[3] = GP_LA(3);                    \
    (R).b[0] = pn_load_so(bz_, bo4); (R).b[1] = pn_load_so(br_, bo4); (R).b[2] = pn_load_so(bh_, bo4); } while (0)
  // One interval.  RF: register set that receives tile g+2; RS: register set holding tile g+1 (stashed into
  // buffer BUF^1).  PI2 / CI2: third accumulator of the previous / current tile (2 = hx for x tiles, 3 = tmp for h).
#ifdef PN_NN_BDMA
#define GP_INTERVAL(gg, BUF, RF, RS, PI2, CI2, HAVE_PREV) do {                                             \
    GP_SEL((gg) + 2); GP_SELB((gg) + 1);                                                                                      \
    PN_SB();                                                                                               \
    pn_lds_read_qf<3, 0, 0>(op0, S.A[BUF], GP_BF(BUF), wave, lane); PN_SB();                        \
    if (HAVE_PREV) PN_G3(op1, 0, x, 0, 1, PI2);                                                            \
    pn_lds_read_qf<3, 0, 1>(op0, S.A[BUF], GP_BF(BUF), wave, lane); PN_SB();                        \
    if (HAVE_PREV) PN_G3(op1, 0, y, 0, 1, PI2);                                                            \
    (RF).a[0] = GP_LA(0); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 0, z, 0, 1, PI2);                                                            \
    (RF).a[1] = GP_LA(1); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 0, w, 0, 1, PI2);                                                            \
    (RF).a[2] = GP_LA(2); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 1, x, 0, 1, PI2);                                                            \
    (RF).a[3] = GP_LA(3); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 1, y, 0, 1, PI2);                                                            \
    GP_DMA(dz_, (BUF) ^ 1, 0); GP_DMA(dr_, (BUF) ^ 1, 1); PN_SB();                                     \
    if (HAVE_PREV) PN_G3(op1, 1, z, 0, 1, PI2);                                                            \
    GP_DMA(dh_, (BUF) ^ 1, 2); PN_SB();                                                                \
    if (HAVE_PREV) PN_G3(op1, 1, w, 0, 1, PI2);                                                            \
    PN_G3(op0, 0, x, 0, 1, CI2);                                                                           \
    pn_lds_read_qf<3, 1, 0>(op1, S.A[BUF], GP_BF(BUF), wave, lane); PN_SB();                        \
    PN_G3(op0, 0, y, 0, 1, CI2);                                                                           \
    pn_lds_read_qf<3, 1, 1>(op1, S.A[BUF], GP_BF(BUF), wave, lane); PN_SB();                        \
    PN_G3(op0, 0, z, 0, 1, CI2);                                                                           \
    pn_store_A1(S.A[(BUF) ^ 1], (RS).a[0], 0); pn_store_A1(S.A[(BUF) ^ 1], (RS).a[1], 1); PN_SB();\
    PN_G3(op0, 0, w, 0, 1, CI2);                                                                           \
    pn_store_A1(S.A[(BUF) ^ 1], (RS).a[2], 2); pn_store_A1(S.A[(BUF) ^ 1], (RS).a[3], 3); PN_SB();\
    PN_G3(op0, 1, x, 0, 1, CI2);                                                                           \
    PN_SB();                                                                                               \
    PN_G3(op0, 1, y, 0, 1, CI2);                                                                           \
    PN_SB();                                                                                               \
    PN_G3(op0, 1, z, 0, 1, CI2);                                                                           \
    PN_G3(op0, 1, w, 0, 1, CI2);                                                                           \
    __syncthreads();                                                                                             \
  } while (0)

#else
#define GP_INTERVAL(gg, BUF, RF, RS, PI2, CI2, HAVE_PREV) do {                                             \
    GP_SEL((gg) + 2);                                                                                      \
    PN_SB();                                                                                               \
    pn_lds_read_q<3, 0, 0>(op0, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                        \
    if (HAVE_PREV) PN_G3(op1, 0, x, 0, 1, PI2);                                                            \
    pn_lds_read_q<3, 0, 1>(op0, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                        \
    if (HAVE_PREV) PN_G3(op1, 0, y, 0, 1, PI2);                                                            \
    (RF).a[0] = GP_LA(0); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 0, z, 0, 1, PI2);                                                            \
    (RF).a[1] = GP_LA(1); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 0, w, 0, 1, PI2);                                                            \
    (RF).a[2] = GP_LA(2); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 1, x, 0, 1, PI2);                                                            \
    (RF).a[3] = GP_LA(3); PN_SB();                                      \
    if (HAVE_PREV) PN_G3(op1, 1, y, 0, 1, PI2);                                                            \
    (RF).b[0] = pn_load_so(bz_, bo4); (RF).b[1] = pn_load_so(br_, bo4); PN_SB();                             \
    if (HAVE_PREV) PN_G3(op1, 1, z, 0, 1, PI2);                                                            \
    (RF).b[2] = pn_load_so(bh_, bo4); PN_SB();                                                         \
    if (HAVE_PREV) PN_G3(op1, 1, w, 0, 1, PI2);                                                            \
    PN_G3(op0, 0, x, 0, 1, CI2);                                                                           \
    pn_lds_read_q<3, 1, 0>(op1, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                        \
    PN_G3(op0, 0, y, 0, 1, CI2);                                                                           \
    pn_lds_read_q<3, 1, 1>(op1, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                        \
    PN_G3(op0, 0, z, 0, 1, CI2);                                                                           \
    pn_store_A1(S.A[(BUF) ^ 1], (RS).a[0], 0); pn_store_A1(S.A[(BUF) ^ 1], (RS).a[1], 1); PN_SB();\
    PN_G3(op0, 0, w, 0, 1, CI2);                                                                           \
    pn_store_A1(S.A[(BUF) ^ 1], (RS).a[2], 2); pn_store_A1(S.A[(BUF) ^ 1], (RS).a[3], 3); PN_SB();\
    PN_G3(op0, 1, x, 0, 1, CI2);                                                                           \
    pn_store_B(&S.B[(BUF) ^ 1][0], (RS).b[0]); pn_store_B(&S.B[(BUF) ^ 1][32], (RS).b[1]); PN_SB();\
    PN_G3(op0, 1, y, 0, 1, CI2);                                                                           \
    pn_store_B(&S.B[(BUF) ^ 1][64], (RS).b[2]); PN_SB();                                         \
    PN_G3(op0, 1, z, 0, 1, CI2);                                                                           \
    PN_G3(op0, 1, w, 0, 1, CI2);                                                                           \
    __syncthreads();                                                                                             \
  } while (0)

#endif
#ifdef PN_NN_BDMA
  { GP_SEL(0); R0.a[0] = GP_LA(0); R0.a[1] = GP_LA(1); R0.a[2] = GP_LA(2); R0.a[3] = GP_LA(3);
    GP_DMA(bz_, 0, 0); GP_DMA(br_, 0, 1); GP_DMA(bh_, 0, 2); }
  { GP_SEL(1); R1.a[0] = GP_LA(0); R1.a[1] = GP_LA(1); R1.a[2] = GP_LA(2); R1.a[3] = GP_LA(3); }
  pn_store_A(S.A[0], R0.a);
#else
  GP_FETCH_ALL(R0, 0); GP_FETCH_ALL(R1, 1);
  pn_tile_stash<3>(S.A[0], S.B[0], R0);
#endif
  __syncthreads();
  // tile g lives in LDS buffer g&1 and, before that, in register set R(g&1)
  GP_INTERVAL(0, 0, R0, R1, 2, 2, false);
#pragma unroll 1
  for (int g = 1; g + 1 < T1; g += 2) {
    GP_INTERVAL(g, 1, R1, R0, 2, 2, true);
    GP_INTERVAL(g + 1, 0, R0, R1, 2, 2, true);
  }
  GP_INTERVAL(T1 - 1, 1, R1, R0, 2, 2, true);      // last x tile (T1 even)
  GP_INTERVAL(T1, 0, R0, R1, 2, 3, true);          // first h tile; its first half still runs the x tile's MFMAs
#pragma unroll 1
  for (int g = T1 + 1; g + 1 < TT; g += 2) {
    GP_INTERVAL(g, 1, R1, R0, 3, 3, true);
    GP_INTERVAL(g + 1, 0, R0, R1, 3, 3, true);
  }
  GP_INTERVAL(TT - 1, 1, R1, R0, 3, 3, true);
  PN_G3(op1, 0, x, 0, 1, 3); PN_G3(op1, 0, y, 0, 1, 3); PN_G3(op1, 0, z, 0, 1, 3); PN_G3(op1, 0, w, 0, 1, 3);
  PN_G3(op1, 1, x, 0, 1, 3); PN_G3(op1, 1, y, 0, 1, 3); PN_G3(op1, 1, z, 0, 1, 3); PN_G3(op1, 1, w, 0, 1, 3);
#undef GP_INTERVAL
#undef GP_FETCH_ALL
#undef GP_LA
#undef GP_SEL
#ifdef PN_NN_BDMA
#undef GP_DMA
#undef GP_SELB
#undef GP_BF
#endif
#ifdef PN_NN_CLOCKS
  if (tid == 0) {
    atomicAdd(&pn_nn_clk[0], (unsigned long long)(__builtin_readcyclecounter() - c0_));
    atomicAdd(&pn_nn_clk[1], (unsigned long long)(wall_clock64() - r0_));
    atomicAdd(&pn_nn_clk[2], 1ull);
  }
#endif
  // gates, candidate, blend (nnet.cpp:144,156,161-179)
  {
    const float bh = b[2 * N + col];
    // previous state for the blend: all 16 loads in flight before the activation arithmetic (the state buffers are
    // allocated with their row count rounded up to the tile, so rows past n_rows are readable; only stores are guarded)
    float ho[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
      ho[i] = h_old[(size_t)(m0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * N + col];
    pn_gru_epilogue(acc, ho, bh, act, S.tansig, h_new, nullptr, N, col, m0 + 32 * wave + 4 * (lane >> 5), n_rows);
  }
#ifdef PN_NN_CLOCKS
  if (tid == 0 && N == 512 && blockIdx.x < 8192) {
    unsigned long long *t = pn_nn_trace + (size_t)blockIdx.x * 4;
    t[0] = (unsigned long long)r0_; t[1] = (unsigned long long)wall_clock64();
    t[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
}

// (Round 5, verdict item 6: a wave-specialised form of the kernel above — four MFMA waves that issue only LDS operand reads, MFMAs and
// the barrier, one or two loader waves doing all the staging — was built, is bit-identical and SLOWER (2.2-2.4 ms); its ablation with
// loaders that move nothing takes 1.548 ms: with no staging at all this tile shape is 1.2 % faster than the kernel above.  The record is
// profiles/r05_gru_wave_specialised.log; the kernel is in the history, one commit before this note.)
// Dense / conv-as-dense with the same half-tile pipeline (KT even, >= 2).
#define PN_DN(o, QQ, c) do {                                                                               \
    _Pragma("unroll") for (int t_ = 0; t_ < NT; t_++)                                                      \
      acc[t_] = __builtin_amdgcn_mfma_f32_32x32x2f32((o).a[QQ].c, (o).b[t_][QQ].c, acc[t_], 0, 0, 0);      \
    PN_SB(); } while (0)
// SH: the layer also feeds the direct-operand GRU kernels (pn_nn_d.hip) — its output leaves through the wave's LDS stage as fp32 rows AND
// as the fragment-order shadow outS (a template parameter, not a run-time branch: the extra epilogue must not cost the plain
// instantiation its second wave per SIMD — it did: 192 -> 202 registers)
template <int NT, bool SH>
__device__ __forceinline__ void pn_dense_mfma_p_body(
    NnShared &S, const PnSegs &A, const float *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_mtiles, int n_cblocks,
    uint4 *__restrict__ outS, int nts_out) {
#ifdef PN_NN_SETPRIO
  __builtin_amdgcn_s_setprio(PN_NN_SETPRIO);   // experiment: the MFMA waves win every issue arbitration against co-resident DSP waves
#endif
  int mt, cb;
  if (!pn_tile_of_block(n_mtiles, n_cblocks, mt, cb)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = mt * BM;
  for (int i = tid; i < 201; i += (int)blockDim.x) S.tansig[i] = tansig[i];   // 201 entries whatever the block size (a 192-thread block once left 192..200 unstaged)
  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int col = (cb * NT + t) * 32 + (lane & 31);
    const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = bv;
  }
  const float *wbase = Wp + (size_t)(cb * NT) * KT * 1024;
  PN_PANEL_LOCALS(A);
  PnTileRegs<NT> R0, R1;
  PnHalfOps<NT> op0, op1;
#define DP_SEL(gg)                                                                                         \
    int g_ = (gg); g_ = g_ < KT ? g_ : KT - 1;                                                             \
    const int sg_ = g_ / tps, ak_ = (g_ - sg_ * tps) * BK;                                                 \
    const float *ap_ = pn_seg_ptr(PN_PANEL_PASS, sg_);                                                     \
    const float *bp_ = wbase + (size_t)g_ * 1024
#define DP_LOADB(R, t0, t1) do { _Pragma("unroll") for (int t_ = (t0); t_ < (t1); t_++)                    \
      (R).b[t_] = pn_load_B(bp_ + (size_t)t_ * KT * 1024); } while (0)
#define DP_STOREB(RS, BUF, t0, t1) do { _Pragma("unroll") for (int t_ = (t0); t_ < (t1); t_++)             \
      pn_store_B(&S.B[(BUF) ^ 1][32 * t_], (RS).b[t_]); } while (0)
#define DP_INTERVAL(gg, BUF, RF, RS, HAVE_PREV) do {                                                       \
    DP_SEL((gg) + 2);                                                                                      \
    PN_SB();                                                                                               \
    pn_lds_read_q<NT, 0, 0>(op0, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                                 \
    if (HAVE_PREV) PN_DN(op1, 0, x);                                                                       \
    pn_lds_read_q<NT, 0, 1>(op0, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                                 \
    if (HAVE_PREV) PN_DN(op1, 0, y);                                                                       \
    (RF).a[0] = pn_load_A1(ap_, pld, ak_, m0, 0); PN_SB();                                                 \
    if (HAVE_PREV) PN_DN(op1, 0, z);                                                                       \
    (RF).a[1] = pn_load_A1(ap_, pld, ak_, m0, 1); PN_SB();                                                 \
    if (HAVE_PREV) PN_DN(op1, 0, w);                                                                       \
    (RF).a[2] = pn_load_A1(ap_, pld, ak_, m0, 2); PN_SB();                                                 \
    if (HAVE_PREV) PN_DN(op1, 1, x);                                                                       \
    (RF).a[3] = pn_load_A1(ap_, pld, ak_, m0, 3); PN_SB();                                                 \
    if (HAVE_PREV) PN_DN(op1, 1, y);                                                                       \
    DP_LOADB(RF, 0, NT / 2); PN_SB();                                                                      \
    if (HAVE_PREV) PN_DN(op1, 1, z);                                                                       \
    DP_LOADB(RF, NT / 2, NT); PN_SB();                                                                     \
    if (HAVE_PREV) PN_DN(op1, 1, w);                                                                       \
    PN_DN(op0, 0, x);                                                                                      \
    pn_lds_read_q<NT, 1, 0>(op1, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                                 \
    PN_DN(op0, 0, y);                                                                                      \
    pn_lds_read_q<NT, 1, 1>(op1, S.A[BUF], S.B[BUF], wave, lane); PN_SB();                                 \
    PN_DN(op0, 0, z);                                                                                      \
    pn_store_A1(S.A[(BUF) ^ 1], (RS).a[0], 0); pn_store_A1(S.A[(BUF) ^ 1], (RS).a[1], 1); PN_SB();         \
    PN_DN(op0, 0, w);                                                                                      \
    pn_store_A1(S.A[(BUF) ^ 1], (RS).a[2], 2); pn_store_A1(S.A[(BUF) ^ 1], (RS).a[3], 3); PN_SB();         \
    PN_DN(op0, 1, x);                                                                                      \
    DP_STOREB(RS, BUF, 0, NT / 2); PN_SB();                                                                \
    PN_DN(op0, 1, y);                                                                                      \
    DP_STOREB(RS, BUF, NT / 2, NT); PN_SB();                                                               \
    PN_DN(op0, 1, z);                                                                                      \
    PN_DN(op0, 1, w);                                                                                      \
    __syncthreads();                                                                                             \
  } while (0)
  pn_dense_fetch<NT>(R0, 0, KT, tps, PN_PANEL_PASS, wbase, m0);
  pn_dense_fetch<NT>(R1, 1, KT, tps, PN_PANEL_PASS, wbase, m0);
  pn_tile_stash<NT>(S.A[0], S.B[0], R0);
  __syncthreads();
  DP_INTERVAL(0, 0, R0, R1, false);
#pragma unroll 1
  for (int g = 1; g + 1 < KT; g += 2) {
    DP_INTERVAL(g, 1, R1, R0, true);
    DP_INTERVAL(g + 1, 0, R0, R1, true);
  }
  DP_INTERVAL(KT - 1, 1, R1, R0, true);
  PN_DN(op1, 0, x); PN_DN(op1, 0, y); PN_DN(op1, 0, z); PN_DN(op1, 0, w);
  PN_DN(op1, 1, x); PN_DN(op1, 1, y); PN_DN(op1, 1, z); PN_DN(op1, 1, w);
#undef DP_INTERVAL
#undef DP_STOREB
#undef DP_LOADB
#undef DP_SEL
  if constexpr (SH) {
    // (the operand buffers are dead: every wave has passed the last interval's barrier)
    float *T = &S.A[0][0][0] + wave * 32 * PN_TLD;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int ct = cb * NT + t;
      if (ct * 32 >= N) break;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) v[i] = pn_act(acc[t][i], act, S.tansig);
      pn_store_tile_frag(T, v, out, ldo, ct * 32, N, m0 + 32 * wave, n_rows, outS + ((size_t)mt * nts_out + ct) * PN_SHADOW_CHUNK, 32 * wave, lane);
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int col = (cb * NT + t) * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = m0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (row < n_rows && col < N) out[(size_t)row * ldo + col] = pn_act(acc[t][i], act, S.tansig);
    }
  }
}

template <int NT>
__global__ __launch_bounds__(NN_THREADS) void pn_dense_mfma_p_kernel(
    PnSegs A, const float *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_mtiles, int n_cblocks) {
  __shared__ NnShared S;
  pn_dense_mfma_p_body<NT, false>(S, A, Wp, bias, N, KT, tps, act, tansig, out, ldo, n_rows, n_mtiles, n_cblocks, nullptr, 0);
}
// the same with the shadow output (conv2 of a context whose GRU steps run on pn_nn_d.hip); two waves per SIMD asked for explicitly
template <int NT>
__global__ __launch_bounds__(NN_THREADS, 2) void pn_dense_mfma_ps_kernel(
    PnSegs A, const float *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_mtiles, int n_cblocks,
    uint4 *__restrict__ outS, int nts_out) {
  __shared__ NnShared S;
  pn_dense_mfma_p_body<NT, true>(S, A, Wp, bias, N, KT, tps, act, tansig, out, ldo, n_rows, n_mtiles, n_cblocks, outS, nts_out);
}

// ---- launchers -----------------------------------------------------------------------------------
// Batches of at most this many streams run the small-batch kernel family (pn_nn_small.hip: one 32x32 tile and one
// accumulator chain per wave, 3-4x more blocks), larger ones the batch-GEMM kernels above.  Same numerics either way.
// Measured crossovers (profiles/r02f_small_batch_study.txt): the dense/conv kernels win up to 4096 streams, the
// gate-per-wave GRU up to ~1500.  PERCEPNET_SMALL_ROWS / PERCEPNET_SMALL_GRU_ROWS override them (0 = never).
int pn_small_rows() {                     // read at every context creation (tests switch families through it)
  const char *e = getenv("PERCEPNET_SMALL_ROWS");
  return e ? atoi(e) : 4096;
}
int pn_small_gru_rows() {
  const char *e = getenv("PERCEPNET_SMALL_GRU_ROWS");
  if (e) return atoi(e);
  const int d = pn_small_rows();
  return d < 1536 ? d : 1536;
}
int pn_launch_dense_small(hipStream_t st, const PnSegs &A, const float *Wp, const float *bias, int N, int act,
                          const float *tansig, float *out, int ldo, int n_rows, int ct_padded);
int pn_launch_gru_small(hipStream_t st, const PnSegs &X, const float *h_old, const float *Wp, const float *Up,
                        const float *b, int N, int act, const float *tansig, float *h_new, int n_rows);
int pn_launch_dense(hipStream_t st, int strict, const PnSegs &A, const float *W, const float *Wp, const float *bias,
                     int N, int act, const float *tansig, float *out, int ldo, int n_rows, int small, void *outS, int nts_out) {
  // outS: fragment-order fp32 shadow of `out` (a buffer nts_out column tiles wide) for the direct-operand GRU kernels — batch kernels only
  if (outS && (strict || small)) { pn_set_error("pn_launch_dense: a shadow output needs the batch-GEMM kernels"); return -1; }
  if (strict) {
    const int nbx = (N + 63) / 64;
    hipLaunchKernelGGL(pn_dense_strict_kernel, dim3((unsigned)nbx * (unsigned)n_rows), dim3(64), 0, st, A, W, bias, N, act, tansig, out, ldo, nbx);
    return 0;
  }
  if (small) {
    return pn_launch_dense_small(st, A, Wp, bias, N, act, tansig, out, ldo, n_rows, pn_ct_padded(N, pn_dense_nt(N)));
  }
  const int tps = (A.width[0] + 31) / 32, KT = tps * A.n;   // equal-width panels
  const int NT = pn_dense_nt(N);
  // the half-tile pipeline consumes K-tiles in pairs: every layer of the PercepNet topology (the only geometry a
  // context accepts, pn_context.cpp:check_geometry) has an even number of them (4, 20, 48, 80, 4)
  if (pn_check_dense_geometry("pn_launch_dense", A.n, A.width, 0)) return -1;
  const int n_cblocks = pn_ct_padded(N, NT) / NT;
  const int n_mtiles = (n_rows + BM - 1) / BM;
  const int grid = 8 * ((n_mtiles + 7) / 8) * n_cblocks;
  if (outS) {
    if (NT != 4) { pn_set_error("pn_launch_dense: a shadow output needs whole 128-column blocks"); return -1; }
    hipLaunchKernelGGL(pn_dense_mfma_ps_kernel<4>, dim3(grid), dim3(NN_THREADS), 0, st, A, Wp, bias, N, KT, tps, act,
                       tansig, out, ldo, n_rows, n_mtiles, n_cblocks, (uint4 *)outS, nts_out);
  } else if (NT == 4)
    hipLaunchKernelGGL(pn_dense_mfma_p_kernel<4>, dim3(grid), dim3(NN_THREADS), 0, st, A, Wp, bias, N, KT, tps, act,
                       tansig, out, ldo, n_rows, n_mtiles, n_cblocks);
  else
    hipLaunchKernelGGL(pn_dense_mfma_p_kernel<2>, dim3(grid), dim3(NN_THREADS), 0, st, A, Wp, bias, N, KT, tps, act,
                       tansig, out, ldo, n_rows, n_mtiles, n_cblocks);
  return 0;
}

int pn_launch_gru(hipStream_t st, int strict, const PnSegs &X, const float *h_old, const float *W, const float *U,
                   const float *Wp, const float *Up, const float *b, int N, int act, const float *tansig,
                   float *h_new, int n_rows, int small) {
  if (strict) {
    const int nbx = (N + 63) / 64;
    hipLaunchKernelGGL(pn_gru_strict_kernel, dim3((unsigned)nbx * (unsigned)n_rows), dim3(64), 0, st, X, h_old, W, U, b, N, act, tansig, h_new, nbx);
    return 0;
  }
  if (small) {
    return pn_launch_gru_small(st, X, h_old, Wp, Up, b, N, act, tansig, h_new, n_rows);
  }
  const int tps = (X.width[0] + 31) / 32, KTx = tps * X.n;   // equal-width panels
  const int NTn = N / 32;
  const int n_mtiles = (n_rows + BM - 1) / BM;
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
  hipLaunchKernelGGL(pn_gru_mfma_p_kernel, dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, Wp, Up, b, N, KTx, tps, act,
                       tansig, h_new, n_rows, n_mtiles);
  return 0;
}
