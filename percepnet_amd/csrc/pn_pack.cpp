// Host-side re-packing of the nnet_data.h weight matrices into the tile orders the fp32 MFMA kernels read (pn_nn.hip,
// pn_nn_small.hip).  No HIP in this file: it is also built alone, with -fsanitize=address,undefined, by the CPU test suite
// (tests/c/host_sanitize.cpp).
#include "pn_common.h"

// ---- host: weight packing ---------------------------------------------------------------------
// W[K][ncols] (reference layout) -> Wp[CT][ceil(K/32)][tile of 1024 floats], zero padded,
// CT = ceil(ncols/32) rounded up to a multiple of ct_round (the kernel's column tiles per block).
// A tile is stored in MFMA FRAGMENT order: element (column j, k_local = 8q + 2s + kh) at ((q*2 + kh)*32 + j)*4 + s, i.e.
// eight 512-byte chunks (q, kh), each holding for the 32 columns the four k values one lane feeds to four consecutive
// MFMA k-steps.  A wavefront whose lane = kh*32 + j reads chunk pair q with ONE fully coalesced 1 KB load (the
// small-batch kernels take their B operand straight from global memory like that); the batch kernels stage a tile into
// LDS with 256 linear float4 loads and un-permute while storing (pn_store_B).
int pn_ct_padded(int ncols, int ct_round) {
  const int CT = (ncols + 31) / 32;
  return ((CT + ct_round - 1) / ct_round) * ct_round;
}
// k_alloc >= K: number of K rows the kernel will sweep (the zero-padded panel width)
size_t pn_packed_floats(int k_alloc, int ncols, int ct_round) {
  return (size_t)pn_ct_padded(ncols, ct_round) * ((k_alloc + 31) / 32) * 1024;
}
void pn_pack_weights(const float *W, int K, int k_alloc, int ncols, int ct_round, float *Wp) {
  const int CT = pn_ct_padded(ncols, ct_round), KT = (k_alloc + 31) / 32;
  for (int ct = 0; ct < CT; ct++)
    for (int kt = 0; kt < KT; kt++) {
      float *tile = Wp + ((size_t)ct * KT + kt) * 1024;
      for (int j = 0; j < 32; j++)
        for (int kl = 0; kl < 32; kl++) {
          const int q = kl >> 3, s = (kl & 7) >> 1, kh = kl & 1;
          const int k = kt * 32 + kl, c = ct * 32 + j;
          tile[((q * 2 + kh) * 32 + j) * 4 + s] = (k < K && c < ncols) ? W[(size_t)k * ncols + c] : 0.f;
        }
    }
}
int pn_dense_nt(int N) { return (N % 128 == 0) ? 4 : 2; }


// weights of a narrow layer for pn_dense_n16_kernel: Wq[ct][t][lane][e] = W[k = 16t + 4e + (lane >> 4)][col = 16 ct + (lane & 15)]
size_t pn_packed_floats_n16(int K, int ncols) { return (size_t)((ncols + 15) / 16) * ((K + 15) / 16) * 256; }
void pn_pack_weights_n16(const float *W, int K, int ncols, float *Wq) {
  const int CT = (ncols + 15) / 16, KG = (K + 15) / 16;
  for (int ct = 0; ct < CT; ct++)
    for (int t = 0; t < KG; t++)
      for (int lane = 0; lane < 64; lane++)
        for (int e = 0; e < 4; e++) {
          const int k = 16 * t + 4 * e + (lane >> 4), c = 16 * ct + (lane & 15);
          Wq[(((size_t)ct * KG + t) * 64 + lane) * 4 + e] = (k < K && c < ncols) ? W[(size_t)k * ncols + c] : 0.f;
        }
}

// ---- launch-geometry predicates without a GPU (include/percepnet_hip.h: pn_debug_check_launch) ----------------------------
#include "pn_launch_check.h"
// kind: 0 dense on the fp32 MFMA kernels (batch and small-batch), 1 dense on the shadow-operand kernels, 2 GRU on the
// shadow-operand kernels (n_out neurons), 3 narrow dense on 16x16x4 tiles.  n_panels panels of `width` columns each.
extern "C" PN_EXPORT int pn_debug_check_launch(int kind, int n_panels, int width, int n_out) {
  int w[5] = {width, width, width, width, width};
  switch (kind) {
    case 0: return pn_check_dense_geometry("pn_launch_dense", n_panels, w, 0);
    case 1: return pn_check_dense_geometry("pn_launch_dense_x3", n_panels, w, 1);
    case 2: return pn_check_gru_geometry("pn_launch_gru_x3", n_panels, w, n_out);
    case 3: return pn_check_n16_geometry("pn_launch_dense_n16", n_panels, w, 8);
    default: pn_set_error("pn_debug_check_launch: unknown kind %d", kind); return -1;
  }
}
