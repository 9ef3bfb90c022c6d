// NOT BUILT (round 3): the first-generation fp16-operand kernels, replaced by the hi-plane-only instantiation of
// pn_nn_x3.hip (A fragments straight from a fragment-order shadow, 64-row wave tiles).  Kept for the measurements DESIGN.md 4.2b cites.
// fp16-input variant of the gain-network kernels (BASELINE.json configs[4]: "fp16 weights/activations
// variant, tolerance re-stated vs CPU fp32 reference") for gfx950.
//
// Same layer graph, tiling, software pipeline and fused epilogues as pn_nn.hip; the only change is
// the GEMM operands: activations are rounded to fp16 (RNE), weights are pre-packed as fp16, and the
// products are accumulated in fp32 by v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate).  Every
// layer's epilogue stores its output twice: fp32 (recurrent state, conv rings as seen by taps, the
// g/r result) and an fp16 shadow copy with the same indexing, which is what the next layer's GEMM
// reads (AH = true): half the operand bytes per K-tile and no conversion while staging.  The first
// layer reads the fp32 features (AH = false) and converts while staging.  Same values either way.  Bias preload, table tanh/sigmoid,
// GRU gating, state blend and every stored activation stay fp32, and the DSP front/back end is
// untouched, so the deviation from the CPU reference comes only from the 11-bit operand mantissas
// (and the hardware's summation order inside a 16-wide MFMA dot).  Tolerance: see
// tests/test_gpu_parity.py::test_fp16_variant_tolerance and DESIGN.md.
//
// Tile: 128 streams x (NT x 32) columns per 256-thread block, K-tile 64 (four MFMA k-steps of 16);
// LDS rows padded to 72 halfs (144 B) -> conflict-free ds_read_b128 / ds_write_b64.
#include "../pn_nn_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

#define HK 64             // K-tile
#define HLD 72            // padded LDS row stride (halfs)

struct NnSharedH {
  _Float16 A[2][BM][HLD];        // 2 x 18432 B
  _Float16 B[2][4 * 32][HLD];    // 2 x 18432 B
  float tansig[208];
};

typedef float fvec4 __attribute__((ext_vector_type(4)));      // plain LLVM vector: stays in registers whatever it is cast from
template <bool AH> struct HTileRegsT { fvec4 a[8]; uint4 b[4]; };   // AH uses a[0..3]

// A tile of 128 rows x 64 k.  AH = false: row-major fp32 panel, 8 float4 per thread (16 per row), converted to
// fp16 when stored to LDS.  AH = true: tile-major fp16 shadow panel (the pointer is a _Float16* carried as float*,
// ld = logical row width), 4 x 16 B per thread (8 per row), copied as is.
// (the register arrays are passed by reference to an array of the exact size: through a decayed pointer the compiler put
// them in scratch memory and the prefetch turned into load -> wait -> scratch store)
template <bool AH>
__device__ __forceinline__ void h_load_A(fvec4 (&ra)[8], const float *__restrict__ p, int ld, int k0, int m0) {
  const int tid = threadIdx.x;
  if constexpr (AH) {
    // shadow buffers are tile-major, [M tile][column tile][128 rows][32 cols]: the 128 x 32 tile a block writes is
    // one contiguous 8 KB run (row-major would leave every 128-byte line half-written by two different blocks)
    const _Float16 *ph = reinterpret_cast<const _Float16 *>(p) + ((size_t)(m0 / BM) * (ld >> 5) + (k0 >> 5)) * (BM * 32);
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int idx = tid + NN_THREADS * it;
      const int row = idx >> 3, c = idx & 7;
      ra[it] = *reinterpret_cast<const fvec4 *>(ph + (size_t)(c >> 2) * (BM * 32) + row * 32 + 8 * (c & 3));
    }
  } else {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int idx = tid + NN_THREADS * it;
      const int row = idx >> 4, c = idx & 15;
      ra[it] = *reinterpret_cast<const fvec4 *>(p + (size_t)(m0 + row) * ld + k0 + 4 * c);
    }
  }
}
template <bool AH>
__device__ __forceinline__ void h_store_A(_Float16 (*As)[HLD], const fvec4 (&ra)[8]) {
  const int tid = threadIdx.x;
  if constexpr (AH) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int idx = tid + NN_THREADS * it;
      *reinterpret_cast<fvec4 *>(&As[idx >> 3][8 * (idx & 7)]) = ra[it];
    }
  } else {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int idx = tid + NN_THREADS * it;
      const int row = idx >> 4, c = idx & 15;
      half4 h;
      h[0] = (_Float16)ra[it][0]; h[1] = (_Float16)ra[it][1]; h[2] = (_Float16)ra[it][2]; h[3] = (_Float16)ra[it][3];
      *reinterpret_cast<half4 *>(&As[row][4 * c]) = h;
    }
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
}


// ---- epilogue through LDS --------------------------------------------------------------------------
// A lane holds 16 values of one output column in 16 different rows, so storing them directly costs 16 dword (and 16
// short) store instructions per 32-column tile; with K loops this short (12 MFMAs of 32 cycles per K-tile) the stores
// would dominate — a VMEM instruction costs the issuing wave ~80-100 cycles whatever its width.  The tile goes
// through LDS instead and leaves as 16-byte stores: 4 per thread for the fp32 rows, 2 for the tile-major fp16 shadow.
struct alignas(16) HStage { float f[BM][32]; _Float16 h[BM][32]; };          // 16 KB + 8 KB, aliases the A/B buffers
static_assert(sizeof(HStage) <= sizeof(NnSharedH) - sizeof(float) * 208, "stage");

// called by all 256 threads; v[i] = value of (row m0 + 32*wave + (i&3) + 8*(i>>2) + 4*(lane>>5), column col0 + (lane&31))
__device__ __forceinline__ void h_store_tile(HStage &T, const float (&v)[16], float *__restrict__ out, int ldo,
                                             _Float16 *__restrict__ outH_tile, int m0, int col0, int n_rows, int n_cols) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __syncthreads();                                     // every wave is done with the LDS contents being replaced
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int r = 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
    T.f[r][lane & 31] = v[i];
    T.h[r][lane & 31] = (_Float16)v[i];
  }
  __syncthreads();
  if (col0 + 32 <= n_cols && (ldo & 3) == 0) {          // whole tile inside the output: 16-byte row segments
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int idx = tid + NN_THREADS * j, r = idx >> 3, c = idx & 7;
      if (m0 + r < n_rows)
        *reinterpret_cast<float4 *>(out + (size_t)(m0 + r) * ldo + col0 + 4 * c) = *reinterpret_cast<const float4 *>(&T.f[r][4 * c]);
    }
  } else {                                              // ragged last column tile (the 34-wide outputs)
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int idx = tid + NN_THREADS * j, r = idx >> 5, c = idx & 31;
      if (m0 + r < n_rows && col0 + c < n_cols) out[(size_t)(m0 + r) * ldo + col0 + c] = T.f[r][c];
    }
  }
  if (outH_tile) {                                      // 8 KB contiguous; rows past n_rows are padding rows of the buffer
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int idx = tid + NN_THREADS * j;
      reinterpret_cast<float4 *>(outH_tile)[idx] = reinterpret_cast<const float4 *>(&T.h[0][0])[idx];
    }
  }
}

template <int NT, bool AH>
__global__ __launch_bounds__(NN_THREADS) void pn_dense_f16_kernel(
    PnSegs A, const _Float16 *__restrict__ Wp, const float *__restrict__ bias, int N, int KT, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ out, int ldo, _Float16 *__restrict__ outH, int ldoH,
    int n_rows, int n_mtiles, int n_cblocks) {
  __shared__ NnSharedH S;
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
  const _Float16 *wbase = Wp + (size_t)(cb * NT) * KT * 2048;
  PN_PANEL_LOCALS(A);
  HTileRegsT<AH> R0, R1;
#define HD_FETCH(R, gg) do {                                                                   \
    int g_ = (gg); g_ = g_ < KT ? g_ : KT - 1;                                                 \
    const int sg_ = g_ / tps, k0_ = (g_ - sg_ * tps) * HK;                                     \
    h_load_A<AH>((R).a, pn_seg_ptr(PN_PANEL_PASS, sg_), pld, k0_, m0);                         \
    _Pragma("unroll") for (int t = 0; t < NT; t++) (R).b[t] = h_load_B(wbase + ((size_t)t * KT + g_) * 2048); \
  } while (0)
#define HD_STASH(R, buf) do {                                                                  \
    h_store_A<AH>(S.A[buf], (R).a);                                                            \
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
  HStage &T = *reinterpret_cast<HStage *>(&S.A[0][0][0]);
  float tab_local = 0; (void)tab_local;
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int col0 = (cb * NT + t) * 32;
    if (col0 >= N) break;                               // padding column tiles of the 34-wide layers (uniform)
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = pn_act(acc[t][i], act, S.tansig);
    _Float16 *tileH = outH ? outH + ((size_t)mt * (ldoH >> 5) + (col0 >> 5)) * (BM * 32) : nullptr;
    h_store_tile(T, v, out, ldo, tileH, m0, col0, n_rows, N);
  }
}

// GRU step, acc[0..3] = z, r, hx, tmp; schedule as in pn_gru_mfma_kernel (x tiles then h tiles)
template <bool AH>
__global__ __launch_bounds__(NN_THREADS) void pn_gru_f16_kernel(
    PnSegs X, const float *__restrict__ h_old, const _Float16 *__restrict__ h_oldH, const _Float16 *__restrict__ Wp,
    const _Float16 *__restrict__ Up, const float *__restrict__ b, int N, int KTx, int tps, int act,
    const float *__restrict__ tansig, float *__restrict__ h_new, _Float16 *__restrict__ h_newH, int n_rows, int n_mtiles) {
  __shared__ NnSharedH S;
  const int NTn = N >> 5;
  int mt, nt;
  if (!pn_tile_of_block(n_mtiles, NTn, mt, nt)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = mt * BM, KTh = N / HK;
  const int T1 = KTx, TT = KTx + KTh;
  const int col = nt * 32 + (lane & 31);
  for (int i = tid; i < 201; i += (int)blockDim.x) S.tansig[i] = tansig[i];   // 201 entries whatever the block size (a 192-thread block once left 192..200 unstaged)
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
  HTileRegsT<AH> R0, R1;
  const float *hA = AH ? reinterpret_cast<const float *>(h_oldH) : h_old;      // recurrent GEMM operand
#define HG_FETCH(R, gg) do {                                                                               \
    int g_ = (gg); g_ = g_ < TT ? g_ : TT - 1;                                                             \
    const bool p1_ = g_ < T1;                                                                              \
    const int kx_ = p1_ ? g_ : 0, kh_ = p1_ ? 0 : g_ - T1;                                                 \
    const int sg_ = kx_ / tps, k0_ = (kx_ - sg_ * tps) * HK;                                               \
    h_load_A<AH>((R).a, p1_ ? pn_seg_ptr(PN_PANEL_PASS, sg_) : hA, p1_ ? pld : N, p1_ ? k0_ : kh_ * HK, m0); \
    (R).b[0] = h_load_B(p1_ ? Wz + (size_t)kx_ * 2048 : Uz + (size_t)kh_ * 2048);                          \
    (R).b[1] = h_load_B(p1_ ? Wr + (size_t)kx_ * 2048 : Ur + (size_t)kh_ * 2048);                          \
    (R).b[2] = h_load_B(p1_ ? Wh + (size_t)kx_ * 2048 : Uh + (size_t)kh_ * 2048);                          \
  } while (0)
#define HG_STASH(R, buf) do {                                                                              \
    h_store_A<AH>(S.A[buf], (R).a);                                                                        \
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
    float ho[16], v[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
      ho[i] = h_old[(size_t)(m0 + 32 * wave + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * N + col];
#pragma unroll
    for (int i = 0; i < 16; i++) {                       // nnet.cpp:144,156,161-179
      const float z = pn_sigmoid(acc[0][i], S.tansig);
      const float r = pn_sigmoid(acc[1][i], S.tansig);
      float h = bh;
      h += acc[3][i] * r;
      h = h + acc[2][i];
      const float hv = pn_act(h, act, S.tansig);
      v[i] = z * ho[i] + (1 - z) * hv;
    }
    HStage &T = *reinterpret_cast<HStage *>(&S.A[0][0][0]);
    _Float16 *tileH = h_newH ? h_newH + ((size_t)mt * (N >> 5) + nt) * (BM * 32) : nullptr;
    h_store_tile(T, v, h_new, N, tileH, m0, nt * 32, n_rows, N);
  }
}

// ---- fp32 rows -> tile-major fp16 shadow [M tile][column tile][128][32] (an RNN state loaded from the host) ---------
// one thread per (row, 8 columns): reads 32 bytes, writes 16
__global__ __launch_bounds__(256) void pn_shadow_f16_kernel(const float *__restrict__ src, int ld, int width,
                                                            _Float16 *__restrict__ dst, int n_rows_padded) {
  const int groups = width >> 3;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = idx / groups;
  const int c8 = (int)(idx - row * groups) * 8;
  if (row >= (size_t)n_rows_padded) return;
  const float4 a = *reinterpret_cast<const float4 *>(src + row * ld + c8), b = *reinterpret_cast<const float4 *>(src + row * ld + c8 + 4);
  half8 h;
  h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
  h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
  _Float16 *tile = dst + ((row / BM) * (width >> 5) + (c8 >> 5)) * (size_t)(BM * 32);
  *reinterpret_cast<half8 *>(tile + (row % BM) * 32 + (c8 & 31)) = h;
}
void pn_launch_shadow_f16(hipStream_t st, const float *src, int ld, int width, void *dst, int n_rows_padded) {
  const size_t n = (size_t)n_rows_padded * (width >> 3);
  hipLaunchKernelGGL(pn_shadow_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ld, width, (_Float16 *)dst,
                     n_rows_padded);
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

// a_half: the panels of A are fp16 shadow buffers (pointers carried as float*, ld in halfs); outH (optional): fp16
// shadow of the output with row stride ldoH
void pn_launch_dense_f16(hipStream_t st, const PnSegs &A, int a_half, const void *Wp, const float *bias, int N, int act,
                         const float *tansig, float *out, int ldo, void *outH, int ldoH, int n_rows) {
  const int tps = (A.width[0] + HK - 1) / HK, KT = tps * A.n;   // equal-width panels, multiples of 64
  const int NT = pn_dense_nt(N);
  const int n_mtiles = (n_rows + BM - 1) / BM;
  const int n_cblocks = h_ct_padded(N, NT) / NT;
  const int grid = 8 * ((n_mtiles + 7) / 8) * n_cblocks;
#define HD_LAUNCH(NT_, AH_)                                                                                       \
  hipLaunchKernelGGL((pn_dense_f16_kernel<NT_, AH_>), dim3(grid), dim3(NN_THREADS), 0, st, A, (const _Float16 *)Wp,  \
                     bias, N, KT, tps, act, tansig, out, ldo, (_Float16 *)outH, ldoH, n_rows, n_mtiles, n_cblocks)
  if (NT == 4) { if (a_half) HD_LAUNCH(4, true); else HD_LAUNCH(4, false); }
  else { if (a_half) HD_LAUNCH(2, true); else HD_LAUNCH(2, false); }
#undef HD_LAUNCH
}

void pn_launch_gru_f16(hipStream_t st, const PnSegs &X, int a_half, const float *h_old, const void *h_oldH,
                       const void *Wp, const void *Up, const float *b, int N, int act, const float *tansig,
                       float *h_new, void *h_newH, int n_rows) {
  const int tps = (X.width[0] + HK - 1) / HK, KTx = tps * X.n;
  const int n_mtiles = (n_rows + BM - 1) / BM, NTn = N / 32;
  const int grid = 8 * ((n_mtiles + 7) / 8) * NTn;
  if (a_half)
    hipLaunchKernelGGL(pn_gru_f16_kernel<true>, dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, (const _Float16 *)h_oldH,
                       (const _Float16 *)Wp, (const _Float16 *)Up, b, N, KTx, tps, act, tansig, h_new,
                       (_Float16 *)h_newH, n_rows, n_mtiles);
  else
    hipLaunchKernelGGL(pn_gru_f16_kernel<false>, dim3(grid), dim3(NN_THREADS), 0, st, X, h_old, (const _Float16 *)h_oldH,
                       (const _Float16 *)Wp, (const _Float16 *)Up, b, N, KTx, tps, act, tansig, h_new,
                       (_Float16 *)h_newH, n_rows, n_mtiles);
}
