// Kernel launchers shared by the host files (pn_context.cpp, pn_featgen.cpp).
#pragma once
#include "pn_common.h"

// grid_cap (last argument of every DSP launcher): test hook of the create-time DSP self-test (pn_context.cpp: dsp_selftest).
// When > 0 the launcher caps its grid at this many blocks, so that a 40-stream batch walks several grid-stride rounds of ONE
// block (the regime in which a mis-scheduled persistent loop once corrupted later rounds).  It is state of
// the temporary self-test context only (pn_ctx::dsp_grid_cap): no other context, thread or device ever sees it.  0 = off.

// ---- kernels / helpers implemented in pn_dsp.hip and pn_nn.hip -----------------------------------
struct PnSegs { const float *p[5]; int ld[5]; int width[5]; int n; };
// in: stream s's 480 samples at in + s*in_stride; i16_scale: 1/32768 (CLI, main.cpp:34) or 1 (training binary,
// denoise.cpp:41,697); aux: optional [n_streams][PN_AUX_STRIDE] side outputs for the training-feature path
void pn_launch_frontend(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in,
                        int in_is_i16, long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring,
                        float2 *Ps, float *feat, int *silence, int *last_period, float *last_gain, float *aux, int grid_cap);
// the same kernel instantiated with two streams per wavefront (pn_dsp_fe_g2.hip): lower latency per stream, lower
// throughput — used by small-batch contexts
void pn_launch_frontend_g2(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in,
                           int in_is_i16, long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring,
                           float2 *Ps, float *feat, int *silence, int *last_period, float *last_gain, float *aux, int grid_cap);
// the phase-split front end (pn_dsp_fe_split_s.hip, pn_dsp_fe_split_p.hip): three launches with their own lane mapping
// and register / LDS budget; spec_in and pitch are independent of each other, spec_out needs both.  Same results, bit
// for bit, as pn_launch_frontend.
void pn_launch_fe_spec_in(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in, int in_is_i16,
                          long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring, int grid_cap);
void pn_launch_fe_pitch(hipStream_t st, int n_streams, int64_t frame, const float *hist, float *feat, int *last_period,
                        float *last_gain, float *aux, int grid_cap);
void pn_launch_fe_spec_out(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const float *hist,
                           const float2 *yring, const float *eyring, const int *last_period, float2 *Ps, float *feat,
                           int *silence, float *aux, int grid_cap);
void pn_launch_frontend_split(hipStream_t st, const PnTables *T, int n_streams, int64_t frame, const void *in, int in_is_i16,
                              long long in_stride, float i16_scale, float *hist, float2 *yring, float *eyring, float2 *Ps,
                              float *feat, int *silence, int *last_period, float *last_gain, float *aux, int grid_cap);
// per-stream re-initialisation (pn_state.hip): rows ids[] of a [n_slots][rows][row_floats] array / of a fragment-order shadow
void pn_launch_zero_rows(hipStream_t st, void *base, int row_floats, long long row_stride, int n_slots, long long slot_stride,
                         const int *d_ids, int n);
void pn_launch_zero_shadow_rows(hipStream_t st, void *S, int width, int np, int n_slots, long long slot_stride_halfs,
                                const int *d_ids, int n);
// per-call active set (pn_active.hip): the rows ids[0..n) are the streams a call does NOT advance
struct PnActiveArgs {
  const int *ids;                                    // inactive stream ids (device)
  float *synth; int *last_period; float *last_gain;  // in-place state
  void *out; int out_row_words; float *d_gr;         // the caller's output rows (480 int16 = 240 words, or 480 floats); d_gr may be NULL
  float *save_synth; uint32_t *save_out; float *save_gr; int *save_period; float *save_gain;   // save area, row i = ids[i]
  float *hist; float2 *yring; float *eyring, *c1ring, *c2ring, *gru[4], *rb;
  uint4 *c1ringH, *c2ringH, *gruH[4], *rbH; int np;  // operand shadows (np = 0: none)
  long long B, Bp, t, tn;                            // t / tn: the counters of the tick the fix-up follows
  int restore_only;                                  // the frame FAILED (a refused launch): put the in-place state and the caller's rows of the
                                                     // skipped streams back, shift nothing (the context's counters did not advance)
};
void pn_launch_inactive_save(hipStream_t st, const PnActiveArgs &a, int n);
int pn_launch_spin(hipStream_t st, long long ticks);     // one wave asleep for `ticks` of the 100 MHz wall clock (queue probe)
void pn_launch_inactive_fixup(hipStream_t st, const PnActiveArgs &a, int n);
// training-feature path (pn_targets.hip)
void pn_launch_targets(hipStream_t st, const PnTables *T, int n_pairs, const float *ex_clean, const float *ex_noisy,
                       const float *ey_look_noisy, const float *aux_clean, const float *aux_noisy,
                       const int *period_noisy, float *records, long long rec_stride, float *gr);
void pn_launch_saturate_i16(hipStream_t st, int n_pairs, const float *in, int16_t *out, long long out_stride);
void pn_launch_backend(hipStream_t st, const PnTables *T, int n_streams, const float2 *Xs, const float2 *Ps,
                       const float *gr, const float *ex_postfilter /* NULL = off */, const int *silence, float *synth_mem,
                       void *out, int out_is_i16, int grid_cap);
size_t pn_packed_floats(int k_alloc, int ncols, int ct_round);
void pn_pack_weights(const float *W, int K, int k_alloc, int ncols, int ct_round, float *Wp);
int pn_dense_nt(int N);
// split-precision variant (pn_nn_x3.hip): operands as fp16 hi/lo planes in fragment order; panels of A / h_oldS / outS /
// h_newS are the uint4* shadows (carried as float* in PnSegs), width = logical columns (multiple of 32)
size_t pn_packed_halfs_x3(int k_alloc, int ncols, int ct_round, int np /* planes: 2 = hi+lo (split precision), 1 = fp16 operands */);
int pn_pack_weights_x3(const float *W, int K, int k_alloc, int ncols, int ct_round, int np, void *Wp);   // -1: weight outside fp16 range
int pn_dense_x3_nt(int N);
int pn_launch_dense_x3(hipStream_t st, const PnSegs &A, const void *Wp, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, void *outS, int nts_out, int n_rows, int rg /* row groups of 32 per wave: 1 | 2 */, int np);
int pn_launch_gru_x3(hipStream_t st, const PnSegs &X, const float *h_old, const void *h_oldS, const void *Wp,
                      const void *Up, const float *b, int N, int act, const float *tansig, float *h_new, void *h_newS,
                      int n_rows, int rg, int np);
int pn_x3_rg_for(int n_rows);
int pn_x3_sat_set(int enable);          // debug counter of operand values clamped to +-65504 (current device): reset + switch
long long pn_x3_sat_read();             // ... and its value, or -1
int pn_launch_split_x3(hipStream_t st, const float *src, int ld, int width, void *S, int n_rows_padded, int np);
// direct-operand fp32 GRU kernels (pn_nn_d.hip): the large-batch GRU steps of nn_mode PN_NN_MFMA.  X panels / h_oldS / h_newS are the
// uint4* fragment-order fp32 shadows (carried as float* in PnSegs), Wp / Up the fp32 packed tiles of pn_pack_weights; bit-identical
// to pn_launch_gru.  rg: row groups of 32 per wave (1 | 2)
int pn_launch_gru_d(hipStream_t st, const PnSegs &X, const float *h_old, const void *h_oldS, const float *Wp,
                    const float *Up, const float *b, int N, int act, const float *tansig, float *h_new, void *h_newS,
                    int n_rows, int rg);
int pn_launch_split_d(hipStream_t st, const float *src, int ld, int width, void *S, int n_rows);
int pn_direct_for(int n_rows);          // 1: a PN_NN_MFMA context of this many streams runs the direct-operand family (PERCEPNET_NN_DIRECT overrides)
int pn_direct_rg_for(int n_rows);       // its rows per wave / 32 (PERCEPNET_NN_DIRECT_RG overrides)
// narrow layers (N <= 48) of small-batch contexts: 16x16x4 MFMA tiles, one wave per (16 rows, 16 columns) (pn_nn_small.hip)
size_t pn_packed_floats_n16(int K, int ncols);
void pn_pack_weights_n16(const float *W, int K, int ncols, float *Wq);
int pn_launch_dense_n16(hipStream_t st, const PnSegs &A, const float *Wq, const float *bias, int N, int act,
                         const float *tansig, float *out, int ldo, int n_rows);
// the batch form for large batches (pn_nn_n48.hip): 128-row blocks x 48 columns of 16x16x4 tiles, the same packed weights Wq
int pn_launch_dense_n48(hipStream_t st, const PnSegs &A, const float *Wq, const float *bias, int N, int act,
                        const float *tansig, float *out, int ldo, int n_rows);
// The network launchers return 0, or -1 (pn_set_error) WITHOUT launching when they refuse a geometry: the caller fails
// the frame (launch_rnn -> pn_process_*), it must never report a frame whose layer outputs are stale.
// pn_check_dense_geometry / pn_check_gru_geometry (pn_launch_check.h) are the HIP-free predicates behind the refusals.
// small: 1 = the small-batch kernel family (pn_nn_small.hip), 0 = the batch-GEMM kernels; ignored when strict.
// pn_small_rows(): the batch size up to which a context picks the small family (PERCEPNET_SMALL_ROWS, default 4096).
int pn_small_rows();
int pn_small_gru_rows();
// outS (optional, batch kernels only): fragment-order fp32 shadow of `out`, a buffer nts_out column tiles wide (pn_nn_common.h)
int pn_launch_dense(hipStream_t st, int strict, const PnSegs &A, const float *W, const float *Wp, const float *bias,
                     int N, int act, const float *tansig, float *out, int ldo, int n_rows, int small, void *outS = nullptr, int nts_out = 0);
int pn_launch_gru(hipStream_t st, int strict, const PnSegs &X, const float *h_old, const float *W, const float *U,
                   const float *Wp, const float *Up, const float *b, int N, int act, const float *tansig,
                   float *h_new, int n_rows, int small);

