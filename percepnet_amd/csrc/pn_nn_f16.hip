// fp16-input variant of the gain-network kernels (BASELINE.json configs[4]: "fp16 weights/activations
// variant, tolerance re-stated vs CPU fp32 reference") for gfx950.
//
// Same layer graph, tiling, software pipeline and fused epilogues as pn_nn.hip; the only change is
// the GEMM operands: activations (fp32 in HBM, as in every other mode) are rounded to fp16 (RNE)
// while they are staged into LDS, weights are pre-packed as fp16, and the products are accumulated
// in fp32 by v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate).  Bias preload, table tanh/sigmoid,
// GRU gating, state blend and every stored activation stay fp32, and the DSP front/back end is
// untouched, so the deviation from the CPU reference comes only from the 11-bit operand mantissas
// (and the hardware's summation order inside a 16-wide MFMA dot).  Tolerance: see
// tests/test_gpu_parity.py::test_fp16_variant_tolerance and DESIGN.md.
//
// Tile: 128 streams x (NT x 32) columns per 256-thread block, K-tile 64 (four MFMA k-steps of 16);
// LDS rows padded to 72 halfs (144 B) -> conflict-free ds_read_b128 / ds_write_b64.
#include "pn_nn_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define HK 64             // K-tile
#define HLD 72            // padded LDS row stride (halfs)

struct NnSharedH {
  _Float16 A[2][BM][HLD];        // 2 x 18432 B
  _Float16 B[2][4 * 32][HLD];    // 2 x 18432 B
  float tansig[208];
};

struct HTileRegs { float4 a[8]; uint4 b[4]; };

// A tile: 128 rows x 64 k of a row-major fp32 panel = 8 float4 per thread (16 float4 per row)
__device__ __forceinline__ void h_load_A(float4 (&ra)[8], const float *__restrict__ p, int ld, int k0, int m0) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int idx = tid + NN_THREADS * it;
    const int row = idx >> 4, c = idx & 15;
    ra[it] = *reinterpret_cast<const float4 *>(p + (size_t)(m0 + row) * ld + k0 + 4 * c);
  }
}
__device__ __forceinline__ void h_store_A(_Float16 (*As)[HLD], const float4 (&ra)[8]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int idx = tid + NN_THREADS * it;
    const int row = idx >> 4, c = idx & 15;
    half4 h;
    h[0] = (_Float16)ra[it].x; h[1] = (_Float16)ra[it].y; h[2] = (_Float16)ra[it].z; h[3] = (_Float16)ra[it].w;
    *reinterpret_cast<half4 *>(&As[row][4 * c]) = h;
  }
}
// one packed 32(col) x 64(k) fp16 weight tile = 4 KB contiguous: 16 B per thread
__device__ __forceinline__ uint4 h_load_B(const _Float16 *__restrict__ tile) {
  const int tid = threadIdx.x;
  return *reinterpret_cast<const uint4 *>(tile + (tid >> 3) * 64 + 8 * (tid & 7));
}
__device__ __forceinline__ void h_store_B(_Float16 (*Bs)[HLD], const uint4 &v) {
  const int tid = threadIdx.x;
  *reinterpret_cast<uint4 *>(&Bs[tid >> 3][8 * (tid & 7)]) = v;
}

// acc[IDX[t]] += A * B[t] for one K-tile of 64
template <int NT, int I0, int I1, int I2, int I3>
__device__ __forceinline__ void h_mma_ktile(const _Float16 (*As)[HLD], const _Float16 (*Bs)[HLD], floatx16 *acc,
                                            int wave, int lane) {
  constexpr int IDX[4] = {I0, I1, I2, I3};
  const int r = lane & 31, kh = lane >> 5;
  __builtin_amdgcn_sched_barrier(0);   // keep the caller's prefetch loads ahead of the MFMAs
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const half8 a = *reinterpret_cast<const half8 *>(&As[32 * wave + r][16 * s + 8 * kh]);
    half8 b[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const half8 *>(&Bs[32 * t + r][16 * s + 8 * kh]);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[IDX[t]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], acc[IDX[t]], 0, 0, 0);
  }
  pn_mfma_drain();
}

template <int NT>
__global__ __launch_bounds__(NN_THREADS) void pn_dense_f16_kernel(
    PnSegs A, const _Float16 *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_mtiles, int n_cblocks) {
  __shared__ NnSharedH S;
  int mt, cb;
  if (!pn_tile_of_block(n_mtiles, n_cblocks, mt, cb)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = mt * BM;
  if (tid < 201) S.tansig[tid] = tansig[tid];
  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int col = (cb * NT + t) * 32 + (lane & 31);
    const float bv = col < N ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[t][i] = bv;
  }
  const _Float16 *wbase = Wp + (size_t)(cb * NT) * KT * 2048;
  PN_PANEL_LOCALS(A);
  HTileRegs R0, R1;
#define HD_FETCH(R, gg) do {                                                                   \
    int g_ = (gg); g_ = g_ < KT ? g_ : KT - 1;                                                 \
    const int sg_ = g_ / tps, k0_ = (g_ - sg_ * tps) * HK;                                     \
    h_load_A((R).a, pn_seg_ptr(PN_PANEL_PASS, sg_), pld, k0_, m0);                             \
    _Pragma("unroll") for (int t = 0; t < NT; t++) (R).b[t] = h_load_B(wbase + ((size_t)t * KT + g_) * 2048); \
  } while (0)
#define HD_STASH(R, buf) do {                                                                  \
    h_store_A(S.A[buf], (R).a);                                                                \
    _Pragma("unroll") for (int t = 0; t < NT; t++) h_store_B(&S.B[buf][32 * t], (R).b[t]);     \
  } while (0)
  HD_FETCH(R0, 0); HD_FETCH(R1, 1);
  HD_STASH(R0, 0);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < KT; g += 2) {
    HD_FETCH(R0, g + 2);
    h_mma_ktile<NT, 0, 1, 2, 3>(S.A[0], S.B[0], acc, wave, lane);
    HD_STASH(R1, 1);
    __syncthreads();
    if (g + 1 < KT) {
      HD_FETCH(R1, g + 3);
      h_mma_ktile<NT, 0, 1, 2, 3>(S.A[1], S.B[1], acc, wave, lane);
      HD_STASH(R0, 0);
      __syncthreads();
    }
  }
#undef HD_FETCH
#undef HD_STASH
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

// GRU step, acc[0..3] = z, r, hx, tmp; schedule as in pn_gru_mfma_kernel (x tiles then h tiles)
__global__ __launch_bounds__(NN_THREADS) void pn_gru_f16_kernel(
    PnSegs X, const float *__restrict__ h_old, const _Float16 *__restrict__ Wp, const _Float16 *__restrict__ Up,
    const float *__restrict__ b, int N, int KTx, int tps, int act, const float *__restrict__ tansig,
    float *__restrict__ h_new, int n_rows, int n_mtiles) {
  __shared__ NnSharedH S;
  const int NTn = N >> 5;
  int mt, nt;
  if (!pn_tile_of_block(n_mtiles, NTn, mt, nt)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = mt * BM, KTh = N / HK;
  const int T1 = KTx, TT = KTx + KTh;
  const int col = nt * 32 + (lane & 31);
  if (tid < 201) S.tansig[tid] = tansig[tid];
  floatx16 acc[4];
  {
    float bz = b[col]; bz += b[3 * N + col];
    float br = b[N + col]; br += b[4 * N + col];
    const float bt = b[5 * N + col];
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[0][i] = bz; acc[1][i] = br; acc[2][i] = 0.f; acc[3][i] = bt; }
  }
  const _Float16 *Wz = Wp + (size_t)(0 * NTn + nt) * KTx * 2048, *Wr = Wp + (size_t)(1 * NTn + nt) * KTx * 2048,
                 *Wh = Wp + (size_t)(2 * NTn + nt) * KTx * 2048;
  const _Float16 *Uz = Up + (size_t)(0 * NTn + nt) * KTh * 2048, *Ur = Up + (size_t)(1 * NTn + nt) * KTh * 2048,
                 *Uh = Up + (size_t)(2 * NTn + nt) * KTh * 2048;
  PN_PANEL_LOCALS(X);
  HTileRegs R0, R1;
#define HG_FETCH(R, gg) do {                                                                               \
    int g_ = (gg); g_ = g_ < TT ? g_ : TT - 1;                                                             \
    const bool p1_ = g_ < T1;                                                                              \
    const int kx_ = p1_ ? g_ : 0, kh_ = p1_ ? 0 : g_ - T1;                                                 \
    const int sg_ = kx_ / tps, k0_ = (kx_ - sg_ * tps) * HK;                                               \
    h_load_A((R).a, p1_ ? pn_seg_ptr(PN_PANEL_PASS, sg_) : h_old, p1_ ? pld : N, p1_ ? k0_ : kh_ * HK, m0); \
    (R).b[0] = h_load_B(p1_ ? Wz + (size_t)kx_ * 2048 : Uz + (size_t)kh_ * 2048);                          \
    (R).b[1] = h_load_B(p1_ ? Wr + (size_t)kx_ * 2048 : Ur + (size_t)kh_ * 2048);                          \
    (R).b[2] = h_load_B(p1_ ? Wh + (size_t)kx_ * 2048 : Uh + (size_t)kh_ * 2048);                          \
  } while (0)
#define HG_STASH(R, buf) do {                                                                              \
    h_store_A(S.A[buf], (R).a);                                                                            \
    h_store_B(&S.B[buf][0], (R).b[0]); h_store_B(&S.B[buf][32], (R).b[1]); h_store_B(&S.B[buf][64], (R).b[2]); \
  } while (0)
  HG_FETCH(R0, 0); HG_FETCH(R1, 1);
  HG_STASH(R0, 0);
  __syncthreads();
#pragma unroll 1
  for (int g = 0; g < T1; g += 2) {
    HG_FETCH(R0, g + 2);
    h_mma_ktile<3, 0, 1, 2, 0>(S.A[0], S.B[0], acc, wave, lane);
    HG_STASH(R1, 1);
    __syncthreads();
    HG_FETCH(R1, g + 3);
    h_mma_ktile<3, 0, 1, 2, 0>(S.A[1], S.B[1], acc, wave, lane);
    HG_STASH(R0, 0);
    __syncthreads();
  }
#pragma unroll 1
  for (int g = T1; g < TT; g += 2) {
    HG_FETCH(R0, g + 2);
    h_mma_ktile<3, 0, 1, 3, 0>(S.A[0], S.B[0], acc, wave, lane);
    HG_STASH(R1, 1);
    __syncthreads();
    HG_FETCH(R1, g + 3);
    h_mma_ktile<3, 0, 1, 3, 0>(S.A[1], S.B[1], acc, wave, lane);
    HG_STASH(R0, 0);
    __syncthreads();
  }
#undef HG_FETCH
#undef HG_STASH
  {
    const float bh = b[2 * N + col];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = m0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      const float z = pn_sigmoid(acc[0][i], S.tansig);
      const float r = pn_sigmoid(acc[1][i], S.tansig);
      float h = bh;
      h += acc[3][i] * r;
      h = h + acc[2][i];
      const float hv = pn_act(h, act, S.tansig);
      if (row < n_rows) {
        const float ho = h_old[(size_t)row * N + col];
        h_new[(size_t)row * N + col] = z * ho + (1 - z) * hv;
      }
    }
  }
}

// ---- host: fp16 weight packing: W[K][ncols] -> [CT][ceil(K/64)][32 cols][64 k] halfs, zero padded ----
static inline int h_ct_padded(int ncols, int ct_round) {
  const int CT = (ncols + 31) / 32;
  return ((CT + ct_round - 1) / ct_round) * ct_round;
}
size_t pn_packed_halfs(int k_alloc, int ncols, int ct_round) {
  return (size_t)h_ct_padded(ncols, ct_round) * ((k_alloc + HK - 1) / HK) * 2048;
}
void pn_pack_weights_f16(const float *W, int K, int k_alloc, int ncols, int ct_round, void *out) {
  _Float16 *Wp = (_Float16 *)out;
  const int CT = h_ct_padded(ncols, ct_round), KT = (k_alloc + HK - 1) / HK;
  for (int ct = 0; ct < CT; ct++)
    for (int kt = 0; kt < KT; kt++) {
      _Float16 *tile = Wp + ((size_t)ct * KT + kt) * 2048;
      for (int j = 0; j < 32; j++)
        for (int kl = 0; kl < HK; kl++) {
          const int k = kt * HK + kl, c = ct * 32 + j;
          tile[j * HK + kl] = (k < K && c < ncols) ? (_Float16)W[(size_t)k * ncols + c] : (_Float16)0.f;
        }
    }
}

int pn_dense_nt(int N);   // pn_nn.hip

void pn_launch_dense_f16(hipStream_t st, const PnSegs &A, const void *Wp, const float *bias, int N, int act,
                         const float *tansig, float *out, int ldo, int n_rows) {
  const int tps = (A.width[0] + HK - 1) / HK, KT = tps * A.n;   // equal-width panels, multiples of 64
  const int NT = pn_dense_nt(N);
  const int n_mtiles = (n_rows + BM - 1) / BM;
  const int n_cblocks = h_ct_padded(N, NT) / NT;
  const int grid = 8 * ((n_mtiles + 7) / 8) * n_cblocks;
  if (NT == 4)
    hipLaunchKernelGGL(pn_dense_f16_kernel<4>, dim3(grid), dim3(NN_THREADS), 0, st, A, (const _Float16 *)Wp, bias, N,
                       KT, tps, act, tansig, out, ldo, n_rows, n_mtiles, n_cblocks);
  else
    hipLaunchKernelGGL(pn_dense_f16_kernel<2>, dim3(grid), dim3(NN_THREADS), 0, st, A, (const _Float16 *)Wp, bias, N,
                       KT, tps, act, tansig, out, ldo, n_rows, n_mtiles, n_cblocks);
}

void pn_launch_gru_f16(hipStream_t st, const PnSegs &X, const float *h_old, const void *Wp, const void *Up,
                       const float *b, int N, int act, const float *tansig, float *h_new, int n_rows) {
  const int tps = (X.width[0] + HK - 1) / HK, KTx = tps * X.n;
  const int n_mtiles = (n_rows + BM - 1) / BM, NTn = N / 32;
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
  hipLaunchKernelGGL(pn_gru_f16_kernel, dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, (const _Float16 *)Wp,
                     (const _Float16 *)Up, b, N, KTx, tps, act, tansig, h_new, n_rows, n_mtiles);
}
