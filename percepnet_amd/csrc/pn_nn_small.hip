// Small-batch network kernels: the latency regime of BASELINE configs[1] (about a thousand concurrent streams).
//
// The batch-GEMM kernels of pn_nn.hip give every wave a 32-row x 96/128-column tile and walk the whole K range with
// it: 16-49 K-tiles of 48-64 dependent-pipe MFMAs, i.e. 43-85 us of matrix-pipe time per block however few blocks
// there are.  At 1024 streams that is 8 M tiles: a 512->512 GRU step runs on 128 of the 256 CUs, one wave per SIMD,
// for 61 us, and the ten layers of compute_rnn (reference rnn.cpp:42-81) take 0.62 ms of a 0.86 ms frame.
//
// Here a wave owns ONE 32x32 output tile and one accumulator chain, and a block is the set of waves that share an
// activation tile:
//   * dense / conv-as-dense: 32 rows x 4 column tiles per block (4 waves); K-tile = 16 MFMAs per wave;
//   * GRU: 32 rows x 32 neurons per block, one wave per GATE (z | r | candidate): the z and r waves walk the x tiles
//     then the h tiles, the candidate wave accumulates W_h x during the x tiles and b_rh + U_h h during the h tiles
//     (the same three chains, in the same k order, as pn_gru_mfma_p_kernel and nnet.cpp:135-167); the three
//     accumulators meet in LDS for the gating epilogue.
// A 512->512 GRU step at 1024 streams becomes 512 blocks x 3 waves with 1/3 of the per-wave chain.  The activation
// tile (shared by the block's waves) is staged through LDS as in the batch kernels; the weight tile of a wave is its
// own, pre-packed in MFMA fragment order (pn_pack_weights), so it goes global -> registers directly, one K-tile ahead.
// Numerics are identical to the batch kernels (same MFMA, same chains): results do not depend on which family ran.
#include "pn_nn_common.h"

#define SBM 32            // rows (streams) per block
#define SLDT 36           // padded LDS row stride (floats): conflict-free ds_read_b128, as in pn_nn.hip

struct SmShared {
  float A[2][SBM][SLDT];  // double-buffered activation K-tile, k-interleaved rows (k = 8q + 2s + kh at q*8 + kh*4 + s)
  float E[4][SBM][33];    // GRU: z | r | W_h x | b_rh + U_h h  accumulators of the block's three waves
  float tansig[208];
};

// activation K-tile: 32 rows x 32 k = 256 float4, loaded by the first 256 threads of the block
template <int NTHREADS>
__device__ __forceinline__ void sm_load_A(float4 (&ra)[(256 + NTHREADS - 1) / NTHREADS], const float *__restrict__ p,
                                          int ld, int k0, int m0) {
#pragma unroll
  for (int it = 0; it < (256 + NTHREADS - 1) / NTHREADS; it++) {
    const int idx = threadIdx.x + NTHREADS * it;
    if (idx < 256) ra[it] = *reinterpret_cast<const float4 *>(p + (size_t)(m0 + (idx >> 3)) * ld + k0 + 4 * (idx & 7));
  }
}
template <int NTHREADS>
__device__ __forceinline__ void sm_store_A(float (*As)[SLDT], const float4 (&ra)[(256 + NTHREADS - 1) / NTHREADS]) {
#pragma unroll
  for (int it = 0; it < (256 + NTHREADS - 1) / NTHREADS; it++) {
    const int idx = threadIdx.x + NTHREADS * it;
    if (idx < 256) {
      const int row = idx >> 3, c = idx & 7;
      float *dst = &As[row][(c >> 1) * 8 + 2 * (c & 1)];
      *reinterpret_cast<float2 *>(dst) = make_float2(ra[it].x, ra[it].z);
      *reinterpret_cast<float2 *>(dst + 4) = make_float2(ra[it].y, ra[it].w);
    }
  }
}
// this wave's packed 32(col) x 32(k) weight tile, stored in fragment order (pn_pack_weights): lane (col r, k-half kh)
// takes float4 number q*64 + lane for q = 0..3 — four fully coalesced 1 KB loads
__device__ __forceinline__ void sm_load_B(float4 (&rb)[4], const float *__restrict__ tile, int lane) {
  const float *p = tile + lane * 4;
#pragma unroll
  for (int q = 0; q < 4; q++) rb[q] = *reinterpret_cast<const float4 *>(p + q * 256);
}
__device__ __forceinline__ void sm_mma_tile(floatx16 &acc, const float (*As)[SLDT], const float4 (&rb)[4], int lane) {
  const int r = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float4 a = *reinterpret_cast<const float4 *>(&As[r][q * 8 + kh * 4]);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, rb[q].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, rb[q].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, rb[q].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, rb[q].w, acc, 0, 0, 0);
  }
}

// Dense / conv-as-dense (compute_dense, compute_conv1d: nnet.cpp:105-118,182-200): out = act(bias + A W), A = up to five
// row-major panels side by side (PnSegs), Wp packed [CT][KT][32][32] with CT padded to a multiple of the batch kernels'
// NT (pn_pack_weights).  Block = 32 rows x 4 column tiles; wave w owns column tile 4*cb + w (idle beyond ct_total).
__global__ __launch_bounds__(256) void pn_dense_small_kernel(
    PnSegs A, const float *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_cblocks, int ct_total) {
  __shared__ SmShared S;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = blockIdx.x / n_cblocks, cb = blockIdx.x - mt * n_cblocks;
  const int m0 = mt * SBM, ct = cb * 4 + wave;
  const bool live = ct < ct_total;
  for (int i = tid; i < 201; i += (int)blockDim.x) S.tansig[i] = tansig[i];   // 201 entries whatever the block size (a 192-thread block once left 192..200 unstaged)
  const int col = ct * 32 + (lane & 31);
  floatx16 acc;
  {
    const float bv = (live && col < N) ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = bv;
  }
  const float *wt = Wp + (size_t)(live ? ct : 0) * KT * 1024;
  PN_PANEL_LOCALS(A);
  // Two weight-tile register sets (rbA: even K-tiles, rbB: odd ones), no copies: the loads of tile g+1 are issued
  // before the MFMAs of tile g and first waited for a whole K-tile later.  The order is pinned with sched_barrier
  // (hipcc otherwise sinks the prefetch below the MFMA block and its latency is exposed every step).  KT is even.
  // Prefetches are UNCONDITIONAL (past-the-end ones re-read the last tile, unused): behind a branch the waitcnt pass
  // must assume the loads may not have been issued and waits vmcnt(3..0) right after them — measured 3.5x slower.
  float4 ra[1], rbA[4], rbB[4];
#define SD_STEP(g_, RB_CUR, RB_NEXT) do {                                                                  \
    sm_load_B(RB_NEXT, wt + (size_t)((g_) + 1 < KT ? (g_) + 1 : KT - 1) * 1024, lane);                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (live) sm_mma_tile(acc, S.A[(g_) & 1], RB_CUR, lane);                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if ((g_) + 1 < KT) sm_store_A<256>(S.A[((g_) + 1) & 1], ra);                                           \
    { const int g2 = (g_) + 2 < KT ? (g_) + 2 : KT - 1, sg = g2 / tps;                                     \
      sm_load_A<256>(ra, pn_seg_ptr(PN_PANEL_PASS, sg), pld, (g2 - sg * tps) * 32, m0); }                  \
    __syncthreads();                                                                                       \
  } while (0)
  sm_load_A<256>(ra, pn_seg_ptr(PN_PANEL_PASS, 0), pld, 0, m0);
  sm_load_B(rbA, wt, lane);
  sm_store_A<256>(S.A[0], ra);
  sm_load_A<256>(ra, pn_seg_ptr(PN_PANEL_PASS, 1 / tps), pld, (1 % tps) * 32, m0);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < KT; g += 2) {
    SD_STEP(g, rbA, rbB);
    SD_STEP(g + 1, rbB, rbA);
  }
#undef SD_STEP
  if (live && col < N) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (row < n_rows) out[(size_t)row * ldo + col] = pn_act(acc[i], act, S.tansig);
    }
  }
}

// Reset-after GRU step (compute_gru, nnet.cpp:120-180) for 32 streams x 32 neurons; wave = gate.
__global__ __launch_bounds__(192) void pn_gru_small_kernel(
    PnSegs X, const float *__restrict__ h_old, const float *__restrict__ Wp, const float *__restrict__ Up,
    const float *__restrict__ b, int N, int KTx, int tps, int act, const float *__restrict__ tansig,
    float *__restrict__ h_new, int n_rows) {
  __shared__ SmShared S;
  const int tid = threadIdx.x, lane = tid & 63, gate = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NTn = N >> 5, KTh = N >> 5;
  const int mt = blockIdx.x / NTn, nt = blockIdx.x - mt * NTn;
  const int m0 = mt * SBM, T1 = KTx, TT = KTx + KTh;
  const int col = nt * 32 + (lane & 31);
  for (int i = tid; i < 201; i += (int)blockDim.x) S.tansig[i] = tansig[i];   // 201 entries whatever the block size (a 192-thread block once left 192..200 unstaged)
  floatx16 acc, acc2;                       // acc: z | r | W_h x ;  acc2 (candidate wave only): b_rh + U_h h
  {
    float b0;
    if (gate == 0) { b0 = b[col]; b0 += b[3 * N + col]; }             // nnet.cpp:135-141
    else if (gate == 1) { b0 = b[N + col]; b0 += b[4 * N + col]; }    // 147-153
    else b0 = 0.f;
    const float bt = b[5 * N + col];                                  // 164
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[i] = b0; acc2[i] = bt; }
  }
  const float *wx = Wp + (size_t)(gate * NTn + nt) * KTx * 1024, *wh = Up + (size_t)(gate * NTn + nt) * KTh * 1024;
  PN_PANEL_LOCALS(X);
  // tile g: x tiles [0, T1) then h tiles [T1, TT)
#define SG_A(g_) ((g_) < T1 ? pn_seg_ptr(PN_PANEL_PASS, (g_) / tps) : h_old)
#define SG_LD(g_) ((g_) < T1 ? pld : N)
#define SG_K0(g_) (((g_) < T1 ? (g_) % tps : (g_) - T1) * 32)
#define SG_B(g_) ((g_) < T1 ? wx + (size_t)(g_) * 1024 : wh + (size_t)((g_) - T1) * 1024)
  float4 ra[2], rbA[4], rbB[4];            // weight-tile register sets for even / odd tiles (see pn_dense_small_kernel)
#define SG_STEP(g_, RB_CUR, RB_NEXT) do {                                                                  \
    { const int g1 = (g_) + 1 < TT ? (g_) + 1 : TT - 1; sm_load_B(RB_NEXT, SG_B(g1), lane); }               \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (gate == 2 && (g_) >= T1) sm_mma_tile(acc2, S.A[(g_) & 1], RB_CUR, lane);                           \
    else sm_mma_tile(acc, S.A[(g_) & 1], RB_CUR, lane);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if ((g_) + 1 < TT) sm_store_A<192>(S.A[((g_) + 1) & 1], ra);                                           \
    { const int g2 = (g_) + 2 < TT ? (g_) + 2 : TT - 1; sm_load_A<192>(ra, SG_A(g2), SG_LD(g2), SG_K0(g2), m0); } \
    __syncthreads();                                                                                       \
  } while (0)
  sm_load_A<192>(ra, SG_A(0), SG_LD(0), SG_K0(0), m0);
  sm_load_B(rbA, SG_B(0), lane);
  sm_store_A<192>(S.A[0], ra);
  sm_load_A<192>(ra, SG_A(1), SG_LD(1), SG_K0(1), m0);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < TT; g += 2) {          // TT = KTx + KTh is even for every GRU of the topology (32, 36)
    SG_STEP(g, rbA, rbB);
    SG_STEP(g + 1, rbB, rbA);
  }
#undef SG_STEP
#undef SG_A
#undef SG_LD
#undef SG_K0
#undef SG_B
  // the three waves' accumulators meet in LDS: E[0] z, E[1] r, E[2] W_h x, E[3] b_rh + U_h h
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    S.E[gate][row][lane & 31] = acc[i];
    if (gate == 2) S.E[3][row][lane & 31] = acc2[i];
  }
  __syncthreads();
  // gates, candidate, blend (nnet.cpp:144,156,161-179), one output per thread and pass; same arithmetic as pn_gru_epilogue
  for (int e = tid; e < SBM * 32; e += 192) {
    const int row = e >> 5, c = e & 31, grow = m0 + row, gcol = nt * 32 + c;
    const float z = pn_sigmoid(S.E[0][row][c], S.tansig);
    const float r = pn_sigmoid(S.E[1][row][c], S.tansig);
    float h = b[2 * N + gcol];
    h += S.E[3][row][c] * r;
    float hp = h + S.E[2][row][c];
    hp = pn_act(hp, act, S.tansig);
    if (grow < n_rows) {
      const float ho = h_old[(size_t)grow * N + gcol];
      h_new[(size_t)grow * N + gcol] = z * ho + (1 - z) * hp;
    }
  }
}


// ---- narrow dense layers (fc_gb: 2560 -> 34, fc_rb: 128 -> 34) in the latency regime -------------------------------------------
// With 32x32 tiles a 34-column layer wastes 47 % of its MFMAs on padding and, worse for one frame's latency, every
// output hangs on a K/2-long chain of 64-cycle v_mfma_f32_32x32x2_f32 (fc_gb: 1280 of them = 34 us however few rows).
// v_mfma_f32_16x16x4_f32 retires four k per 32-cycle issue (40 dependent) and is the SAME k-ascending fmaf chain bit for
// bit (tools/probes/mfma16_korder_probe.hip: 256/256 outputs), so a wave here owns one 16-row x 16-column tile and one
// accumulator: fc_gb's chain is 640 instructions.  One wave per block, nothing shared, no barrier:
//   * weights pre-packed per (16-column tile, 16-k group, lane) as float4 = the lane's b for four consecutive MFMAs
//     (pn_pack_weights_n16): one coalesced 1 KB load per four MFMAs;
//   * activations: lane (row r, k-quarter g) loads A[r][16t + 4g .. +3]; the MFMA wants lane (r, g) to feed k = 4e + g,
//     a 4x4 transpose inside each 16-float run, done through a wave-private LDS tile (one ds_write_b128, four
//     ds_read_b32 per four MFMAs);
//   * eight 16-k groups of both operands in flight (the chain is paced by L2 latency, not bandwidth).
// Chain order = bias, then k ascending: identical to the other kernel families and to sgemv_accum (nnet.cpp:59-72).
#define N16_DEPTH 8
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64, 1) void pn_dense_n16_kernel(        // (64, 1): one wave per block, up to 512 registers — no spills
    PnSegs A, const float *__restrict__ Wq, const float *__restrict__ bias, int N, int KG, int lg_gps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_ctiles) {
  __shared__ __attribute__((aligned(16))) float T[2][16][20];     // transposing tile, rows padded to 20 floats
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const int mt = blockIdx.x / n_ctiles, ct = blockIdx.x - mt * n_ctiles;
  const int m0 = mt * 16, col = ct * 16 + r;
  floatx4 acc;
  {
    const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = bv;
  }
  PN_PANEL_LOCALS(A);
  const float *wq = Wq + ((size_t)ct * KG) * 256 + lane * 4;
  const size_t arow = (size_t)(m0 + r) * pld + 4 * g;
  // group t (16 k): panel t >> lg_gps (panel widths are powers of two: 128, 512), k offset 16 (t mod groups-per-panel)
  auto a_ptr = [&](int t) { const int sg = t >> lg_gps; return pn_seg_ptr(PN_PANEL_PASS, sg) + arow + 16 * (t - (sg << lg_gps)); };
  // Two register sets of eight 16-k groups each (named registers: as arrays the compiler left them in scratch memory).
  // While one set feeds 32 MFMAs (~1300 cycles) the 16 loads of the other are in flight — more than an L2 round trip.
  // (A rolling eight-deep prefetch with one load pair per group was tried first: across the loop back-edge the compiler's
  // wait-count pass falls back to vmcnt(0) before every use, which exposes the full load latency per group.)
  float4 pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7, pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7;
  float4 qa0, qa1, qa2, qa3, qa4, qa5, qa6, qa7, qb0, qb1, qb2, qb3, qb4, qb5, qb6, qb7;
#define N16_FILL(RA, RB, t_) do { const int tt_ = (t_) < KG ? (t_) : KG - 1;    /* past the end: re-read the last group, unused */ \
    RA = *reinterpret_cast<const float4 *>(a_ptr(tt_)); RB = *reinterpret_cast<const float4 *>(wq + (size_t)tt_ * 256); } while (0)
#define N16_FILL_SET(P, t_) do { N16_FILL(P##a0, P##b0, (t_)); N16_FILL(P##a1, P##b1, (t_) + 1); N16_FILL(P##a2, P##b2, (t_) + 2);   \
    N16_FILL(P##a3, P##b3, (t_) + 3); N16_FILL(P##a4, P##b4, (t_) + 4); N16_FILL(P##a5, P##b5, (t_) + 5);                            \
    N16_FILL(P##a6, P##b6, (t_) + 6); N16_FILL(P##a7, P##b7, (t_) + 7); } while (0)
#define N16_GROUP(d_, RA, RB) do {                                                                             \
    float (*Tt)[20] = T[(d_) & 1];                                                                             \
    *reinterpret_cast<float4 *>(&Tt[r][4 * g]) = RA;                                                           \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();                    \
    const float a0_ = Tt[r][g], a1_ = Tt[r][4 + g], a2_ = Tt[r][8 + g], a3_ = Tt[r][12 + g];                   \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0_, RB.x, acc, 0, 0, 0);                                       \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1_, RB.y, acc, 0, 0, 0);                                       \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2_, RB.z, acc, 0, 0, 0);                                       \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3_, RB.w, acc, 0, 0, 0);                                       \
  } while (0)
#define N16_RUN_SET(P) do { N16_GROUP(0, P##a0, P##b0); N16_GROUP(1, P##a1, P##b1); N16_GROUP(2, P##a2, P##b2);                     \
    N16_GROUP(3, P##a3, P##b3); N16_GROUP(4, P##a4, P##b4); N16_GROUP(5, P##a5, P##b5); N16_GROUP(6, P##a6, P##b6);                  \
    N16_GROUP(7, P##a7, P##b7); } while (0)
  N16_FILL_SET(p, 0);
#pragma unroll 1
  for (int t0 = 0;; t0 += 2 * N16_DEPTH) {           // KG is a multiple of N16_DEPTH (launcher)
    N16_FILL_SET(q, t0 + N16_DEPTH);
    __builtin_amdgcn_sched_barrier(0);
    N16_RUN_SET(p);
    __builtin_amdgcn_sched_barrier(0);
    if (t0 + N16_DEPTH >= KG) break;
    N16_FILL_SET(p, t0 + 2 * N16_DEPTH);
    __builtin_amdgcn_sched_barrier(0);
    N16_RUN_SET(q);
    __builtin_amdgcn_sched_barrier(0);
    if (t0 + 2 * N16_DEPTH >= KG) break;
  }
#undef N16_RUN_SET
#undef N16_GROUP
#undef N16_FILL_SET
#undef N16_FILL
  if (col < N) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = m0 + 4 * g + i;                 // D[4 (lane / 16) + i][lane % 16]
      if (row < n_rows) out[(size_t)row * ldo + col] = pn_act(acc[i], act, tansig);
    }
  }
}

int pn_launch_dense_n16(hipStream_t st, const PnSegs &A, const float *Wq, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, int n_rows) {
  if (pn_check_n16_geometry("pn_launch_dense_n16", A.n, A.width, N16_DEPTH)) return -1;
  const int gps = A.width[0] / 16, KG = gps * A.n;      // equal-width panels, widths multiples of 16 (128, 512)
  const int n_mt = (n_rows + 15) / 16, n_ct = (N + 15) / 16;
  int lg = 0;
  while ((1 << lg) < gps) lg++;
  hipLaunchKernelGGL(pn_dense_n16_kernel, dim3(n_mt * n_ct), dim3(64), 0, st, A, Wq, bias, N, KG, lg, act, tansig, out, ldo,
                     n_rows, n_ct);
  return 0;
}

// ---- launchers (called from pn_launch_dense / pn_launch_gru when the batch is small) -------------------------------
int pn_launch_dense_small(hipStream_t st, const PnSegs &A, const float *Wp, const float *bias, int N, int act,
                          const float *tansig, float *out, int ldo, int n_rows, int ct_padded) {
  if (pn_check_dense_geometry("pn_launch_dense_small", A.n, A.width, 0)) return -1;   // K-tiles alternate between two register sets
  const int tps = (A.width[0] + 31) / 32, KT = tps * A.n;   // equal-width panels
  const int n_mt = (n_rows + SBM - 1) / SBM, ct_total = (N + 31) / 32, n_cblocks = (ct_total + 3) / 4;
  (void)ct_padded;
  hipLaunchKernelGGL(pn_dense_small_kernel, dim3(n_mt * n_cblocks), dim3(256), 0, st, A, Wp, bias, N, KT, tps, act, tansig,
                     out, ldo, n_rows, n_cblocks, ct_total);
  return 0;
}
int pn_launch_gru_small(hipStream_t st, const PnSegs &X, const float *h_old, const float *Wp, const float *Up,
                        const float *b, int N, int act, const float *tansig, float *h_new, int n_rows) {
  const int tps = (X.width[0] + 31) / 32, KTx = tps * X.n;
  for (int j = 1; j < X.n; j++) if (X.width[j] != X.width[0]) { pn_set_error("pn_launch_gru_small: unequal panel widths"); return -1; }
  if ((N & 31) || ((KTx + N / 32) & 1)) { pn_set_error("pn_launch_gru_small: %d input + %d recurrent K-tiles (the sum must be even, N whole tiles)", KTx, N / 32); return -1; }
  const int n_mt = (n_rows + SBM - 1) / SBM;
  hipLaunchKernelGGL(pn_gru_small_kernel, dim3(n_mt * (N / 32)), dim3(192), 0, st, X, h_old, Wp, Up, b, N, KTx, tps, act,
                     tansig, h_new, n_rows);
  return 0;
}
