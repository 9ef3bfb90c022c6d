// fc_gb (2560 -> 34, compute_dense on the concatenated GRU outputs, reference rnn.cpp:72-77, nnet.cpp:105-118) at LARGE batches on
// v_mfma_f32_16x16x4_f32 tiles (round 6; round-5 verdict item 8).
//
// The batch-GEMM kernels of pn_nn.hip pad the 34 output columns to two 32-column tiles: 47 % of fc_gb's MFMAs multiply zeros
// (0.178 ms at 65 536 streams = 0.40 of the fp32 matrix peak counted on useful flops).  v_mfma_f32_16x16x4_f32 retires the SAME
// k-ascending fmaf chain bit for bit (pn_nn_small.hip: pn_dense_n16_kernel, which serves these layers in the latency regime) on
// 16-column tiles: three of them cover the 34 columns with 29 % padding.  Here the batch form:
//   block = 4 waves x 32 rows = 128 streams x 48 columns; wave = two 16-row groups x three 16-column tiles = 6 accumulators;
//   per 16-k group a wave reads 2 activation + 3 weight fragments (ds_read_b128 each) for 24 MFMAs of 32 cycles.
//   A (row-major panels) is staged through LDS in the 16x16x4 fragment order [16-row group][16-k group][k quarter][row][e]
//   (lane (row r, quarter kq) feeds k = 16 t + 4 e + kq, e = 0..3: one ds_read_b128), transposed while it is written;
//   B comes packed in that order already (pn_pack_weights_n16) and is copied linearly.
// Same chain as every other family (bias, then k ascending): results bit-identical to pn_launch_dense on this layer.
#include "pn_nn_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define Q_CT 3                          // 16-column tiles per block (48 columns)
#ifndef PN_N48_RGW
#define PN_N48_RGW 2                    // 16-row groups per wave: 2 = 128-row blocks, 1 = 64-row blocks (twice the blocks, twice the weight staging)
#endif

template <int RGW> struct QShared {
  float A[2][4 * RGW * 2 * 4 * 16 * 4]; // [buffer][16-row group][t 0..1][kq][row][e]: 2 x 8 RGW KB
  fvec4 B[2][2][Q_CT][64];              // [buffer][t][column tile][lane]: 2 x 6 KB
  float tansig[208];
};

// activation K-tile: 64 RGW rows x 32 k = 512 RGW float4, 2 RGW per thread (row = idx >> 3, float4 column c = idx & 7)
template <int NA>
__device__ __forceinline__ void q_load_A(fvec4 (&ra)[NA], const float *__restrict__ p, int ld, int k0, int m0) {
#pragma unroll
  for (int it = 0; it < NA; it++) {
    const int idx = threadIdx.x + NN_THREADS * it;
    ra[it] = *reinterpret_cast<const fvec4 *>(p + (size_t)(m0 + (idx >> 3)) * ld + k0 + 4 * (idx & 7));
  }
}
// k_local = 4 c + j: group t = c >> 2, e = c & 3, quarter kq = j -> the four floats of a float4 go to the four quarter slabs
template <int NA>
__device__ __forceinline__ void q_store_A(float *As, const fvec4 (&ra)[NA]) {
#pragma unroll
  for (int it = 0; it < NA; it++) {
    const int idx = threadIdx.x + NN_THREADS * it;
    const int row = idx >> 3, c = idx & 7;
    float *dst = As + ((((row >> 4) * 2 + (c >> 2)) * 4) * 16 + (row & 15)) * 4 + (c & 3);
    dst[0] = ra[it].x; dst[64] = ra[it].y; dst[128] = ra[it].z; dst[192] = ra[it].w;
  }
}

template <int RGW>
__global__ __launch_bounds__(NN_THREADS) void pn_dense_n48_kernel(
    PnSegs A, const float *__restrict__ Wq, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows) {
  __shared__ QShared<RGW> S;
  constexpr int Q_BM = 64 * RGW, NA = 2 * RGW;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m0 = blockIdx.x * Q_BM, KG = 2 * KT;
  for (int i = tid; i < 201; i += NN_THREADS) S.tansig[i] = tansig[i];
  floatx4 acc[RGW][Q_CT];
#pragma unroll
  for (int ct = 0; ct < Q_CT; ct++) {
    const int col = 16 * ct + (lane & 15);
    const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
    for (int rg = 0; rg < RGW; rg++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[rg][ct][i] = bv;
  }
  PN_PANEL_LOCALS(A);
  // weight K-tile g = the 16-k groups 2g, 2g + 1 of the three column tiles: 6 x 64 float4; thread tid copies float4 number
  // tid (tid < 192: t = 0, ct = tid >> 6; 192..255: t = 1, ct = 0) and number 256 + (tid & 127) (t = 1, ct = 1 | 2) of the sequence
  // [t][ct][lane] — the upper two waves repeat the lower two waves' second copy (same address, same value): no branch around a load
  const int t0 = tid < 192 ? 0 : 1, c0 = tid < 192 ? (tid >> 6) : 0;
  const int t1 = 1, c1 = 1 + ((tid & 127) >> 6);
  fvec4 ra0[NA], ra1[NA], rb0[2], rb1[2];
#define Q_LOADB(rb, gg) do { int g_ = (gg); g_ = g_ < KT ? g_ : KT - 1;                                                           \
    (rb)[0] = *reinterpret_cast<const fvec4 *>(Wq + (((size_t)c0 * KG + 2 * g_ + t0) * 64 + lane) * 4);                          \
    (rb)[1] = *reinterpret_cast<const fvec4 *>(Wq + (((size_t)c1 * KG + 2 * g_ + t1) * 64 + lane) * 4); } while (0)
#define Q_STOREB(buf, rb) do { S.B[buf][t0][c0][lane] = (rb)[0]; S.B[buf][t1][c1][lane] = (rb)[1]; } while (0)
#define Q_LOADA(ra, gg) do { int g_ = (gg); g_ = g_ < KT ? g_ : KT - 1; const int sg_ = g_ / tps;                                 \
    q_load_A<NA>(ra, pn_seg_ptr(PN_PANEL_PASS, sg_), pld, (g_ - sg_ * tps) * 32, m0); } while (0)
  // one K-tile from buffer BUF: two 16-k groups x (2 row groups x 3 column tiles x 4 MFMAs); the fragments of group 1 are read
  // while the MFMAs of group 0 run
#define Q_FRAGS(BUF, t, fa, fb) do {                                                                                              \
    _Pragma("unroll") for (int rg = 0; rg < RGW; rg++)                                                                            \
      (fa)[rg] = *reinterpret_cast<const fvec4 *>(&S.A[BUF][((((RGW * wave + rg) * 2 + (t)) * 4) * 16) * 4 + lane * 4]);           \
    _Pragma("unroll") for (int ct = 0; ct < Q_CT; ct++) (fb)[ct] = S.B[BUF][t][ct][lane]; } while (0)
#define Q_MMA(fa, fb) do {                                                                                                        \
    _Pragma("unroll") for (int e = 0; e < 4; e++)                                                                                 \
      _Pragma("unroll") for (int rg = 0; rg < RGW; rg++)                                                                          \
        _Pragma("unroll") for (int ct = 0; ct < Q_CT; ct++)                                                                       \
          acc[rg][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32((fa)[rg][e], (fb)[ct][e], acc[rg][ct], 0, 0, 0); } while (0)
  // interval g: tile g from LDS buffer BUF; RF receives tile g + 2 from memory, RS (tile g + 1) goes to the other buffer
#define Q_INTERVAL(gg, BUF, RFA, RFB, RSA, RSB) do {                                                                              \
    fvec4 fa0[RGW], fb0[Q_CT], fa1[RGW], fb1[Q_CT];                                                                                   \
    Q_FRAGS(BUF, 0, fa0, fb0);                                                                                                    \
    Q_LOADA(RFA, (gg) + 2); Q_LOADB(RFB, (gg) + 2);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    Q_FRAGS(BUF, 1, fa1, fb1);                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    Q_MMA(fa0, fb0);                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    q_store_A<NA>(S.A[(BUF) ^ 1], RSA); Q_STOREB((BUF) ^ 1, RSB);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                            \
    Q_MMA(fa1, fb1);                                                                                                              \
    __syncthreads();                                                                                                              \
  } while (0)
  Q_LOADA(ra0, 0); Q_LOADB(rb0, 0);
  Q_LOADA(ra1, 1); Q_LOADB(rb1, 1);
  q_store_A<NA>(S.A[0], ra0); Q_STOREB(0, rb0);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < KT; g += 2) {                       // KT is even (launcher)
    Q_INTERVAL(g, 0, ra0, rb0, ra1, rb1);
    Q_INTERVAL(g + 1, 1, ra1, rb1, ra0, rb0);
  }
#undef Q_INTERVAL
#undef Q_MMA
#undef Q_FRAGS
#undef Q_LOADA
#undef Q_STOREB
#undef Q_LOADB
#pragma unroll
  for (int ct = 0; ct < Q_CT; ct++) {
    const int col = 16 * ct + (lane & 15);
    if (col >= N) continue;
#pragma unroll
    for (int rg = 0; rg < RGW; rg++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int row = m0 + 16 * RGW * wave + 16 * rg + 4 * (lane >> 4) + i;      // D[4 (lane / 16) + i][lane % 16]
        if (row < n_rows) out[(size_t)row * ldo + col] = pn_act(acc[rg][ct][i], act, S.tansig);
      }
  }
}

// Wq: pn_pack_weights_n16 (16-column tiles x 16-k groups x 64 lanes x 4).  N <= 48; panels of equal width, whole 32-column
// tiles, an even number of them in total (the K loop consumes tiles in pairs and clamps its prefetch to the last one)
int pn_launch_dense_n48(hipStream_t st, const PnSegs &A, const float *Wq, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, int n_rows) {
  if (pn_check_dense_geometry("pn_launch_dense_n48", A.n, A.width, 1)) return -1;
  if (N < 1 || N > 16 * Q_CT) { pn_set_error("pn_launch_dense_n48: %d output columns (1..%d)", N, 16 * Q_CT); return -1; }
  const int tps = A.width[0] / 32, KT = tps * A.n;
  constexpr int Q_BM = 64 * PN_N48_RGW;
  const int n_mt = (n_rows + Q_BM - 1) / Q_BM;
  hipLaunchKernelGGL(pn_dense_n48_kernel<PN_N48_RGW>, dim3(n_mt), dim3(NN_THREADS), 0, st, A, Wq, bias, N, KT, tps, act, tansig, out, ldo, n_rows);
  return 0;
}
