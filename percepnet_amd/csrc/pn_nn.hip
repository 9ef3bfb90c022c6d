// Gain-network kernels (compute_rnn, reference rnn.cpp:42-81) batched over streams for gfx950.
//
// MFMA path (PN_NN_MFMA): every layer is a GEMM  [B streams] x [K inputs] x [N neurons]  on
// v_mfma_f32_32x32x2_f32 (exact fp32, a k-ascending fmaf chain per output element), streams on
// the M axis, with bias preload and the table activation / GRU gating fused in the epilogue.
// The summation order is the reference's (sgemv_accum, nnet.cpp:59-72 / vec.h:102-135): start
// from the bias, add input contributions k = 0..K-1, then recurrent ones; the reset-after GRU
// (compute_gru, nnet.cpp:120-180) is evaluated in the same three steps as the reference
// (z,r and tmp = b_rh + U_h h  ->  h = b_h + tmp*r, then += W_h x  ->  blend), which needs a
// second sweep over x but keeps the chain order.  Only difference from the CPU path: fused
// instead of separate rounding of each multiply-add.
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
#include "pn_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BK 32
#define LDT 36            // padded LDS row stride (floats)
#define NN_THREADS 256

struct PnSegs {           // A operand = concatenation along K of up to 5 row-major panels
  const float *p[5];
  int ld[5];              // row stride (floats)
  int width[5];           // valid columns; the MFMA path requires every panel to be readable (and
                          // zero) up to the next multiple of 32 and all panels to be equally wide
  int n;
};

enum { ACT_LINEAR = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_RELU = 3 };

// tansig_approx / sigmoid_approx (reference vec.h:53-75)
__device__ __forceinline__ float pn_tansig(float x, const float *tab) {
  float sign = 1;
  if (x < 0) { x = -x; sign = -1; }
  const float v = floorf(.5f + 25 * x);
  // the reference's x86-64 build converts with cvttss2si: out-of-range / NaN -> INT_MIN, which
  // the clamp below then turns into index 0 (not 200); mirrored here so that even absurd
  // pre-activations (> 8.6e7) behave like the CPU path
  int i = (v < 2147483648.f) ? (int)v : (int)0x80000000;
  i = i > 200 ? 200 : i;
  i = i < 0 ? 0 : i;
  x -= .04f * i;
  float y = tab[i];
  const float dy = 1 - y * y;
  y = y + x * dy * (1 - y * x);
  return sign * y;
}
__device__ __forceinline__ float pn_sigmoid(float x, const float *tab) { return .5f + .5f * pn_tansig(.5f * x, tab); }
__device__ __forceinline__ float pn_act(float v, int act, const float *tab) {
  if (act == ACT_SIGMOID) return pn_sigmoid(v, tab);
  if (act == ACT_TANH) return pn_tansig(v, tab);
  if (act == ACT_RELU) return v < 0 ? 0 : v;
  return v;
}

// =============================== STRICT kernels ==================================================
// W in the reference layout [K][ncols] (nnet_data.h); thread = (stream blockIdx.y, neuron).
__global__ void pn_dense_strict_kernel(PnSegs A, const float *__restrict__ W, const float *__restrict__ bias,
                                       int N, int act, const float *__restrict__ tansig, float *__restrict__ out,
                                       int ldo) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
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
                                     const float *__restrict__ tansig, float *__restrict__ h_new) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
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
  float A[BM][LDT];        // 18432 B
  float B[4 * 32][LDT];    // 18432 B (up to 4 column tiles: dense NT<=4, GRU 3 gates)
  float tansig[208];
};

// stage a 128 x 32 activation tile: rows m0.., columns k0..k0+31 of panel p (zero beyond
// n_rows), k-interleaved into S.A.  Panels are padded to a multiple of 32 columns.
__device__ __forceinline__ void pn_stage_A(float (*As)[LDT], const float *__restrict__ p, int ld, int k0,
                                           int m0, int n_rows) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int idx = tid + NN_THREADS * it;
    const int row = idx >> 3, c = idx & 7;        // 8 float4 per row
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + row < n_rows) v = *reinterpret_cast<const float4 *>(p + (size_t)(m0 + row) * ld + k0 + 4 * c);
    // k_local = 4c + {0,1,2,3}: q = c>>1, (kh,s) = (0,2(c&1)), (1,2(c&1)), (0,2(c&1)+1), (1,2(c&1)+1)
    float *dst = &As[row][(c >> 1) * 8 + 2 * (c & 1)];
    *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.z);
    *reinterpret_cast<float2 *>(dst + 4) = make_float2(v.y, v.w);
  }
}

// stage one packed 32(col) x 32(k) weight tile (1024 contiguous floats, already k-interleaved)
__device__ __forceinline__ void pn_stage_B(float (*Bs)[LDT], const float *__restrict__ tile) {
  const int tid = threadIdx.x;
  const int j = tid >> 3, c = tid & 7;
  const float4 v = *reinterpret_cast<const float4 *>(tile + j * 32 + 4 * c);
  *reinterpret_cast<float4 *>(&Bs[j][4 * c]) = v;
}

// Toolchain hazard found on ROCm 7.2 / gfx950 (DESIGN.md "MFMA result hazard"): when a loop of
// v_mfma_f32_32x32x2_f32 exits, hipcc places the first read of the accumulator tuple (a
// v_accvgpr_mov of element 15, the register the 16th pass writes last) only `s_nop 1` after the
// final MFMA, and that read returns the value from BEFORE it: output rows 27/31 (mod 32) silently
// lose the last k-step.  The hazard recogniser does not look across the loop back-edge / exit
// copies.  Every K-tile therefore ends with an explicit drain of the matrix pipe (32 wait states
// >= the 19 a 16-pass MFMA needs), pinned in place with scheduling barriers: ~1 % of a K-tile.
__device__ __forceinline__ void pn_mfma_drain() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int NT>
__device__ __forceinline__ void pn_mma_ktile(const float (*As)[LDT], const float (*Bs)[LDT], floatx16 *acc,
                                             int wave, int lane) {
  const int r = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float4 a = *reinterpret_cast<const float4 *>(&As[32 * wave + r][q * 8 + kh * 4]);
    float4 b[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const float4 *>(&Bs[32 * t + r][q * 8 + kh * 4]);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
  }
  pn_mfma_drain();
}

// XCD-aware block numbering: hardware places block b on XCD b % 8; give each XCD whole
// activation panels (all column tiles of an M tile run on the same XCD's L2).
__device__ __forceinline__ bool pn_tile_of_block(int n_mtiles, int n_ctiles, int &mt, int &ct) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  mt = (idx / n_ctiles) * 8 + xcd;
  ct = idx % n_ctiles;
  return mt < n_mtiles;
}

// Dense / conv-as-dense: out[m][n] = act(bias[n] + sum_k A[m][k] W[k][n]); Wp packed
// [ctile][ktile][32 cols][32 k-interleaved]; NT column tiles per block.  tps = K-tiles per panel.
template <int NT>
__global__ __launch_bounds__(NN_THREADS) void pn_dense_mfma_kernel(
    PnSegs A, const float *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, int n_rows, int n_mtiles, int n_cblocks) {
  __shared__ NnShared S;
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
#pragma unroll 1
  for (int kt = 0; kt < KT; kt++) {
    const int sg = kt / tps, k0 = (kt - sg * tps) * BK;
    __syncthreads();
    pn_stage_A(S.A, A.p[sg], A.ld[sg], k0, m0, n_rows);
#pragma unroll
    for (int t = 0; t < NT; t++) pn_stage_B(&S.B[32 * t], Wp + ((size_t)(cb * NT + t) * KT + kt) * 1024);
    __syncthreads();
    pn_mma_ktile<NT>(S.A, S.B, acc, wave, lane);
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

// Reset-after GRU step for a 128-stream x 32-neuron tile.
// Wp: packed input weights  [3N/32 ctiles][KTx][32][32]; Up: packed recurrent [3N/32][N/32][32][32].
// acc[0..3] = z, r, tmp (= b_rh + U_h h), h.
__global__ __launch_bounds__(NN_THREADS) void pn_gru_mfma_kernel(
    PnSegs X, const float *__restrict__ h_old, const float *__restrict__ Wp, const float *__restrict__ Up,
    const float *__restrict__ b, int N, int KTx, int tps, int act, const float *__restrict__ tansig,
    float *__restrict__ h_new, int n_rows, int n_mtiles) {
  __shared__ NnShared S;
  const int NTn = N >> 5;                       // neuron tiles
  int mt, nt;
  if (!pn_tile_of_block(n_mtiles, NTn, mt, nt)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = mt * BM, KTh = N >> 5;
  const int col = nt * 32 + (lane & 31);
  if (tid < 201) S.tansig[tid] = tansig[tid];

  floatx16 acc[4];
  {
    float bz = b[col]; bz += b[3 * N + col];    // nnet.cpp:135-141
    float br = b[N + col]; br += b[4 * N + col];// 147-153
    const float bt = b[5 * N + col];            // 164
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[0][i] = bz; acc[1][i] = br; acc[2][i] = bt; }
  }
  // z,r += W_{z,r} x
#pragma unroll 1
  for (int kt = 0; kt < KTx; kt++) {
    const int sg = kt / tps, k0 = (kt - sg * tps) * BK;
    __syncthreads();
    pn_stage_A(S.A, X.p[sg], X.ld[sg], k0, m0, n_rows);
    pn_stage_B(&S.B[0], Wp + ((size_t)(0 * NTn + nt) * KTx + kt) * 1024);
    pn_stage_B(&S.B[32], Wp + ((size_t)(1 * NTn + nt) * KTx + kt) * 1024);
    __syncthreads();
    pn_mma_ktile<2>(S.A, S.B, acc, wave, lane);
  }
  // z,r,tmp += U_{z,r,h} h_old
#pragma unroll 1
  for (int kt = 0; kt < KTh; kt++) {
    __syncthreads();
    pn_stage_A(S.A, h_old, N, kt * BK, m0, n_rows);
#pragma unroll
    for (int g = 0; g < 3; g++) pn_stage_B(&S.B[32 * g], Up + ((size_t)(g * NTn + nt) * KTh + kt) * 1024);
    __syncthreads();
    pn_mma_ktile<3>(S.A, S.B, acc, wave, lane);
  }
  // gates; h = b_h + tmp * r  (nnet.cpp:144,156,161-166)
  {
    const float bh = b[2 * N + col];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      acc[0][i] = pn_sigmoid(acc[0][i], S.tansig);
      acc[1][i] = pn_sigmoid(acc[1][i], S.tansig);
      float h = bh;
      h += acc[2][i] * acc[1][i];
      acc[3][i] = h;
    }
  }
  // h += W_h x  (nnet.cpp:167)
#pragma unroll 1
  for (int kt = 0; kt < KTx; kt++) {
    const int sg = kt / tps, k0 = (kt - sg * tps) * BK;
    __syncthreads();
    pn_stage_A(S.A, X.p[sg], X.ld[sg], k0, m0, n_rows);
    pn_stage_B(&S.B[0], Wp + ((size_t)(2 * NTn + nt) * KTx + kt) * 1024);
    __syncthreads();
    pn_mma_ktile<1>(S.A, S.B, acc + 3, wave, lane);
  }
  // activation + blend (nnet.cpp:175-179)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int row = m0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    if (row < n_rows) {
      const float hv = pn_act(acc[3][i], act, S.tansig);
      const float z = acc[0][i];
      const float ho = h_old[(size_t)row * N + col];
      h_new[(size_t)row * N + col] = z * ho + (1 - z) * hv;
    }
  }
}

// ---- host: weight packing ---------------------------------------------------------------------
// W[K][ncols] (reference layout) -> Wp[CT][ceil(K/32)][32 cols][32 k-interleaved], zero padded,
// CT = ceil(ncols/32) rounded up to a multiple of ct_round (the kernel's column tiles per block).
// position of k_local = 8q + 2s + kh inside a tile row is q*8 + kh*4 + s.
static inline int pn_ct_padded(int ncols, int ct_round) {
  const int CT = (ncols + 31) / 32;
  return ((CT + ct_round - 1) / ct_round) * ct_round;
}
size_t pn_packed_floats(int K, int ncols, int ct_round) {
  return (size_t)pn_ct_padded(ncols, ct_round) * ((K + 31) / 32) * 1024;
}
void pn_pack_weights(const float *W, int K, int ncols, int ct_round, float *Wp) {
  const int CT = pn_ct_padded(ncols, ct_round), KT = (K + 31) / 32;
  for (int ct = 0; ct < CT; ct++)
    for (int kt = 0; kt < KT; kt++) {
      float *tile = Wp + ((size_t)ct * KT + kt) * 1024;
      for (int j = 0; j < 32; j++)
        for (int kl = 0; kl < 32; kl++) {
          const int q = kl >> 3, s = (kl & 7) >> 1, kh = kl & 1;
          const int k = kt * 32 + kl, c = ct * 32 + j;
          tile[j * 32 + q * 8 + kh * 4 + s] = (k < K && c < ncols) ? W[(size_t)k * ncols + c] : 0.f;
        }
    }
}
int pn_dense_nt(int N) { return (N % 128 == 0) ? 4 : 2; }

// ---- launchers -----------------------------------------------------------------------------------
void pn_launch_dense(hipStream_t st, int strict, const PnSegs &A, const float *W, const float *Wp, const float *bias,
                     int N, int act, const float *tansig, float *out, int ldo, int n_rows) {
  if (strict) {
    dim3 grid((N + 63) / 64, n_rows);
    hipLaunchKernelGGL(pn_dense_strict_kernel, grid, dim3(64), 0, st, A, W, bias, N, act, tansig, out, ldo);
    return;
  }
  const int tps = (A.width[0] + 31) / 32, KT = tps * A.n;   // equal-width panels
  const int NT = pn_dense_nt(N);
  const int n_mtiles = (n_rows + BM - 1) / BM;
  const int n_cblocks = pn_ct_padded(N, NT) / NT;
  const int grid = 8 * ((n_mtiles + 7) / 8) * n_cblocks;
  if (NT == 4)
    hipLaunchKernelGGL(pn_dense_mfma_kernel<4>, dim3(grid), dim3(NN_THREADS), 0, st, A, Wp, bias, N, KT, tps, act,
                       tansig, out, ldo, n_rows, n_mtiles, n_cblocks);
  else
    hipLaunchKernelGGL(pn_dense_mfma_kernel<2>, dim3(grid), dim3(NN_THREADS), 0, st, A, Wp, bias, N, KT, tps, act,
                       tansig, out, ldo, n_rows, n_mtiles, n_cblocks);
}

void pn_launch_gru(hipStream_t st, int strict, const PnSegs &X, const float *h_old, const float *W, const float *U,
                   const float *Wp, const float *Up, const float *b, int N, int act, const float *tansig,
                   float *h_new, int n_rows) {
  if (strict) {
    dim3 grid((N + 63) / 64, n_rows);
    hipLaunchKernelGGL(pn_gru_strict_kernel, grid, dim3(64), 0, st, X, h_old, W, U, b, N, act, tansig, h_new);
    return;
  }
  const int tps = (X.width[0] + 31) / 32, KTx = tps * X.n;   // equal-width panels
  const int n_mtiles = (n_rows + BM - 1) / BM, NTn = N / 32;
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
  hipLaunchKernelGGL(pn_gru_mfma_kernel, dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, Wp, Up, b, N, KTx, tps, act,
                     tansig, h_new, n_rows, n_mtiles);
}
