// Shared device helpers of the network kernels (pn_nn.hip: fp32 MFMA + STRICT, pn_nn_small.hip, pn_nn_x3.hip: fp16 matrix cores).
#pragma once
#include "pn_launch_check.h"
#include "pn_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define BM 128            // streams per block tile
#define NN_THREADS 256

struct PnSegs {           // A operand = concatenation along K of up to 5 row-major panels
  const float *p[5];
  int ld[5];              // row stride (floats)
  int width[5];           // valid columns; the MFMA path requires every panel to be readable (and
                          // zero) up to the next multiple of 32 and all panels to be equally wide
  int n;
};

enum { ACT_LINEAR = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_RELU = 3 };

// panel pointer by (uniform) index without dynamically indexing the by-value kernel argument
// (a runtime index would spill the whole struct to scratch)
#define PN_PANEL_ARGS const float *pp0, const float *pp1, const float *pp2, const float *pp3, const float *pp4, int pld
#define PN_PANEL_PASS pp0, pp1, pp2, pp3, pp4, pld
#define PN_PANEL_LOCALS(A) const float *pp0 = (A).p[0], *pp1 = (A).p[1], *pp2 = (A).p[2], *pp3 = (A).p[3], \
                           *pp4 = (A).p[4]; const int pld = (A).ld[0]
__device__ __forceinline__ const float *pn_seg_ptr(PN_PANEL_ARGS, int sg) {
  (void)pld;
  const float *p = pp0;
  p = sg == 1 ? pp1 : p;
  p = sg == 2 ? pp2 : p;
  p = sg == 3 ? pp3 : p;
  p = sg == 4 ? pp4 : p;
  return p;
}

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
// the same function in two halves, so that callers can issue many table reads before using any of them
struct PnTsArg { float sign, x; int i; };
__device__ __forceinline__ PnTsArg pn_tansig_arg(float x) {
  PnTsArg a;
  a.sign = 1;
  if (x < 0) { x = -x; a.sign = -1; }
  const float v = floorf(.5f + 25 * x);
  int i = (v < 2147483648.f) ? (int)v : (int)0x80000000;
  i = i > 200 ? 200 : i;
  i = i < 0 ? 0 : i;
  a.x = x - .04f * i;
  a.i = i;
  return a;
}
__device__ __forceinline__ float pn_tansig_fin(const PnTsArg &a, float y) {
  const float dy = 1 - y * y;
  y = y + a.x * dy * (1 - y * a.x);
  return a.sign * y;
}
__device__ __forceinline__ float pn_sigmoid(float x, const float *tab) { return .5f + .5f * pn_tansig(.5f * x, tab); }
__device__ __forceinline__ float pn_act(float v, int act, const float *tab) {
  if (act == ACT_SIGMOID) return pn_sigmoid(v, tab);
  if (act == ACT_TANH) return pn_tansig(v, tab);
  if (act == ACT_RELU) return v < 0 ? 0 : v;
  return v;
}

// GRU gates, candidate and blend for the 16 outputs of a lane (nnet.cpp:144,156,161-179; activations vec.h:53-75).
// Staged so that the table reads of all outputs are in flight together: evaluated one output at a time each of the
// 48 dependent LDS reads costs its full latency.  row0 = first row of the lane (rows row0 + (i&3) + 8(i>>2)).
// v[i] = new state of the lane's i-th output from the four accumulators z, r, hx (W_h x), tmp (U_h h + b)
__device__ __forceinline__ void pn_gru_gate16(const floatx16 &az_, const floatx16 &ar_, const floatx16 &ahx, const floatx16 &atmp,
                                              const float (&ho)[16], float bh, int act, const float *tab, float (&v)[16]) {
  PnTsArg az[16], ar[16];
  float tz[16], tr[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { az[i] = pn_tansig_arg(.5f * az_[i]); ar[i] = pn_tansig_arg(.5f * ar_[i]); }
#pragma unroll
  for (int i = 0; i < 16; i++) { tz[i] = tab[az[i].i]; tr[i] = tab[ar[i].i]; }
  float z[16], hp[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    z[i] = .5f + .5f * pn_tansig_fin(az[i], tz[i]);                 // sigmoid_approx
    const float r = .5f + .5f * pn_tansig_fin(ar[i], tr[i]);
    float h = bh;
    h += atmp[i] * r;
    hp[i] = h + ahx[i];
  }
  if (act == ACT_TANH || act == ACT_SIGMOID) {
    PnTsArg ah[16];
#pragma unroll
    for (int i = 0; i < 16; i++) ah[i] = pn_tansig_arg(act == ACT_SIGMOID ? .5f * hp[i] : hp[i]);
    float th[16];
#pragma unroll
    for (int i = 0; i < 16; i++) th[i] = tab[ah[i].i];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float t = pn_tansig_fin(ah[i], th[i]);
      hp[i] = act == ACT_SIGMOID ? .5f + .5f * t : t;
    }
  } else if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 16; i++) hp[i] = hp[i] < 0 ? 0 : hp[i];
  }
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = z[i] * ho[i] + (1 - z[i]) * hp[i];
}
__device__ __forceinline__ void pn_gru_epilogue(const floatx16 *acc, const float (&ho)[16], float bh, int act,
                                                const float *tab, float *__restrict__ h_new,
                                                _Float16 *__restrict__ h_newH, int N, int col, int row0, int n_rows) {
  float v[16];
  pn_gru_gate16(acc[0], acc[1], acc[2], acc[3], ho, bh, act, tab, v);
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int row = row0 + (i & 3) + 8 * (i >> 2);
    if (row < n_rows) {
      h_new[(size_t)row * N + col] = v[i];
      // fp16 shadow in the first-generation fp16 kernels' layout, tile-major [M tile][column tile][128][32] (experimental/pn_nn_f16_v1.hip)
      if (h_newH) h_newH[(((size_t)(row / BM) * (N >> 5) + (col >> 5)) * BM + (row % BM)) * 32 + (col & 31)] = (_Float16)v[i];
    }
  }
}

// ---- fragment-order fp32 shadows (direct-operand GRU kernels, pn_nn_d.hip) --------------------------------------------------
//   shadow[M tile of 128][column tile of 32][q 0..3][kh 0..1][row 0..127][4 floats: k = 32 ct + 8 q + 2 s + kh, s = 0..3]
// = 8 slabs (j = 2q + kh) of 128 sixteen-byte entries per (M tile, column tile): lane (row r, k-half kh) of a consumer wave
// loads entry (j, r) = its operand of four consecutive v_mfma_f32_32x32x2_f32 k-steps.
typedef float fvec4 __attribute__((ext_vector_type(4)));
#define PN_SHADOW_CHUNK 1024           // uint4 per (M tile of 128, column tile)
#define PN_TLD 36                      // epilogue stage: 32 rows x 32 columns per wave, rows padded to 36 floats
// Stores one 32 x 32 output tile of a wave (v[i] = value of row (i&3) + 8(i>>2) + 4(lane>>5), column lane&31) through
// the wave's private LDS stage, kept in FRAGMENT column order (column c = 8q + 2s + kh at 8q + 4kh + s): one
// ds_read_b128 is a slab entry.  fp32 rows go out as 32-byte runs (4 lanes = one 128-byte row segment; out may be
// null), the shadow as 16-byte entries, 32 consecutive rows of a slab = 512 contiguous bytes (S may be null; srow0 = first
// row within the M tile of 128).  Rows at or past n_rows are not stored anywhere.
__device__ __forceinline__ void pn_store_tile_frag(float *T, const float (&v)[16], float *__restrict__ out, int ldo, int col0,
                                             int n_cols, int grow0, int n_rows, uint4 *__restrict__ S, int srow0, int lane) {
  {
    const int c = lane & 31, pc = (c & 24) + 4 * (c & 1) + ((c >> 1) & 3);
#pragma unroll
    for (int i = 0; i < 16; i++) T[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * PN_TLD + pc] = v[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (out) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const int idx = lane + 64 * p, row = idx >> 2, q = idx & 3;
      const fvec4 e = *reinterpret_cast<const fvec4 *>(&T[row * PN_TLD + 8 * q]);        // kh = 0: columns 8q + 0, 2, 4, 6
      const fvec4 o = *reinterpret_cast<const fvec4 *>(&T[row * PN_TLD + 8 * q + 4]);    // kh = 1: columns 8q + 1, 3, 5, 7
      if (grow0 + row < n_rows) {
        float *dst = out + (size_t)(grow0 + row) * ldo + col0 + 8 * q;
        if (col0 + 32 <= n_cols && (ldo & 3) == 0) {
          *reinterpret_cast<fvec4 *>(dst) = fvec4{e.x, o.x, e.y, o.y};
          *reinterpret_cast<fvec4 *>(dst + 4) = fvec4{e.z, o.z, e.w, o.w};
        } else {
          const float f[8] = {e.x, o.x, e.y, o.y, e.z, o.z, e.w, o.w};
#pragma unroll
          for (int j = 0; j < 8; j++) if (col0 + 8 * q + j < n_cols) dst[j] = f[j];
        }
      }
    }
  }
  if (S) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int j = 2 * p + (lane >> 5), row = lane & 31;                                // slab j = 2q + kh
      // (row-guarded like the fp32 rows: with row-range chains the rows past n_rows belong to another chain's launches)
      if (grow0 + row < n_rows) S[j * 128 + srow0 + row] = *reinterpret_cast<const uint4 *>(&T[row * PN_TLD + 4 * j]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();                      // the stage is reused by the next tile of this wave
}


// MFMA result hazard: a VALU / VMEM read of a register written by v_mfma_f32_32x32x2_f32 needs 18 wait
// states after the MFMA on gfx950 — measured (tools/probes/mfma_waitstate_probe.hip: 17 states still return the old
// value in every lane, 18 never do) — and that is exactly what hipcc's hazard recogniser inserts, also across a loop
// exit (tools/probes/mfma_exit_hazard.hip, profiles/r02c_*).  Nothing in these kernels reads an accumulator before the
// epilogue, so no manual padding is needed; round 1's pn_mfma_drain() (32 wait states per K-tile) is gone.  The
// margin is zero by construction, which is why pn_ctx_create runs a known-answer self-test of these kernels.

// XCD-aware block numbering: hardware places block b on XCD b % 8; give each XCD whole
// activation panels (all column tiles of an M tile run on the same XCD's L2).
__device__ __forceinline__ bool pn_tile_of_block(int n_mtiles, int n_ctiles, int &mt, int &ct) {
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  mt = (idx / n_ctiles) * 8 + xcd;
  ct = idx % n_ctiles;
  return mt < n_mtiles;
}

