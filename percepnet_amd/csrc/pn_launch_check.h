// Geometry predicates of the network launchers (pn_nn.hip, pn_nn_small.hip, pn_nn_x3.hip) — HIP-free, so that the
// refusals can be exercised without a GPU (pn_debug_check_launch, tests/test_abi.py).  A launcher that refuses returns -1
// with pn_set_error and launches nothing; launch_rnn / pn_process_* fail the frame (round-4 verdict item 8: a refused
// launch used to return silently and the frame completed with stale layer outputs).
#pragma once
void pn_set_error(const char *fmt, ...);

// dense / conv1d layers: the K range is n_panels panels of EQUAL width, swept as 32-column tiles that the software
// pipelines consume in PAIRS (the prefetch clamps to the last tile: an odd count would accumulate it twice).
// whole_tiles: the shadow-operand kernels read whole 32-column tiles only (fp32 kernels zero-fill a ragged last tile).
static inline int pn_check_dense_geometry(const char *who, int n_panels, const int *width, int whole_tiles) {
  if (n_panels < 1 || n_panels > 5) { pn_set_error("%s: %d panels (1..5)", who, n_panels); return -1; }
  for (int j = 1; j < n_panels; j++)
    if (width[j] != width[0]) { pn_set_error("%s: unequal panel widths (%d, %d)", who, width[0], width[j]); return -1; }
  if (width[0] < 1 || (whole_tiles && (width[0] & 31))) { pn_set_error("%s: panel width %d (need whole 32-column tiles)", who, width[0]); return -1; }
  const int kt = (width[0] + 31) / 32 * n_panels;
  if (kt < 2 || (kt & 1)) { pn_set_error("%s: %d K-tiles of panel width %d (must be an even count)", who, kt, width[0]); return -1; }
  return 0;
}
// GRU layers of the shadow-operand kernels: input and recurrent k-tiles both consumed in pairs, N whole column tiles
static inline int pn_check_gru_geometry(const char *who, int n_panels, const int *width, int N) {
  if (pn_check_dense_geometry(who, n_panels, width, 1)) return -1;
  if (N < 32 || (N & 31) || ((N / 32) & 1)) { pn_set_error("%s: %d neurons (need an even number of 32-column tiles)", who, N); return -1; }
  return 0;
}
// narrow layers on 16x16x4 tiles: k-groups of 16 in bursts of `depth`, power-of-two groups per panel
static inline int pn_check_n16_geometry(const char *who, int n_panels, const int *width, int depth) {
  if (n_panels < 1 || n_panels > 5) { pn_set_error("%s: %d panels (1..5)", who, n_panels); return -1; }
  for (int j = 1; j < n_panels; j++)
    if (width[j] != width[0]) { pn_set_error("%s: unequal panel widths (%d, %d)", who, width[0], width[j]); return -1; }
  const int gps = width[0] / 16;
  if (gps < 1 || (width[0] & 15) || (gps & (gps - 1))) { pn_set_error("%s: panel width %d is not a power-of-two number of 16-column groups", who, width[0]); return -1; }
  if ((gps * n_panels) % depth) { pn_set_error("%s: %d k-groups (must be a multiple of %d)", who, gps * n_panels, depth); return -1; }
  return 0;
}
