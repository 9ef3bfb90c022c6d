/* libpercepnet_hip.so — C-ABI of the MI355X-native batched PercepNet inference path.
 *
 * Two layers of entry points:
 *
 *  (1) The reference's own frame-engine interface (reference src/rnnoise.h:49-68), same names,
 *      argument meaning and return values, so existing callers (reference src/main.cpp:30-39)
 *      re-link unchanged.  The reference compiles its "C" API as C++ (no extern "C" in
 *      rnnoise.h), so its exported symbols are Itanium-mangled; this library exports BOTH the
 *      mangled names (csrc/rnnoise_compat.cpp) and the extern "C" ones declared below with a
 *      pn_ prefix-free alias set (`rnnoise_*_c`).
 *
 *  (2) The batched interface the GPU needs: one context = B independent 48 kHz streams advanced
 *      in lock-step, one 10 ms frame (480 samples) per stream per call.  Plain pointers and
 *      sizes only; device pointers are raw HIP device addresses (e.g. torch's data_ptr()).
 *
 * All functions are thread-compatible per context; one context belongs to one HIP device.
 * Errors: functions returning int give 0 on success, <0 on failure; pn_last_error() returns a
 * thread-local description.  The library never falls back to a CPU path: if no HIP device is
 * usable, context creation fails.
 */
#ifndef PERCEPNET_HIP_H
#define PERCEPNET_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include "percepnet_nnet_data.h"

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared between this push and its pop (plus the
   reference's nine C++-mangled names, csrc/rnnoise_compat.cpp) are exported.  Harmless for callers. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define PN_FRAME_SIZE 480      /* reference denoise.cpp:19 */
#define PN_NB_BANDS 34         /* reference denoise.cpp:35 */
#define PN_NB_FEATURES 70      /* reference denoise.cpp:40 */

typedef struct pn_ctx pn_ctx;
typedef struct pn_model pn_model;

/* Network evaluation mode. */
enum {
  PN_NN_MFMA = 0,    /* fp32 MFMA GEMM over the stream batch (v_mfma_f32_32x32x2_f32): each output
                        is a k-ascending fmaf chain from the bias — the reference's summation
                        order (nnet.cpp:59-72) with fused instead of separate rounding */
  PN_NN_STRICT = 1,  /* one lane per (stream, neuron), separate mul and add in the reference's
                        order: bit-identical to the CPU reference; slow, for parity tests */
  PN_NN_MFMA_F16 = 2,/* BASELINE configs[4]: the GEMM operands (weights and activations) of conv1, conv2, the five
                        GRUs and fc_gb rounded to fp16, fp32 accumulation (v_mfma_f32_32x32x16_f16); fc (70 inputs)
                        and fc_rb (K = 128) run on the fp32 kernels; bias, activations, gating, state, DSP fp32.
                        Operands beyond +-65504 saturate (PERCEPNET_X3_SATCOUNT=1 counts them).
                        Tolerance re-stated: DESIGN.md 4.2b (6 LSB bound, 4 measured) */
  PN_NN_MFMA_X3 = 3  /* split precision: every fp32 GEMM operand is carried as an fp16 (hi, lo) pair and every
                        product formed as lo*hi + hi*lo + hi*hi by three v_mfma_f32_32x32x16_f16 into an fp32
                        accumulator (operand error ~2^-22, below the reference's own accumulation rounding);
                        state, gating and activations fp32.  Same parity bounds as PN_NN_MFMA */
};

/* ---- models ------------------------------------------------------------------------------ */
/* Borrow an in-memory RNNModel laid out as nnet_data.h lays it out (replaces the reference's
   link-time `percepnet_model_orig`, denoise.cpp:49-51,267).  The arrays are copied. */
pn_model *pn_model_from_rnnmodel(const RNNModel *m);
/* Load the PNW1 container (percepnet_amd/weights.py) — this library's implementation of the
   declared-but-undefined rnnoise_model_from_file (rnnoise.h:62). */
pn_model *pn_model_from_blob(const void *blob, size_t nbytes);
pn_model *pn_model_from_file(FILE *f);
void pn_model_free(pn_model *m);

/* ---- batched contexts ---------------------------------------------------------------------- */
/* device: HIP device ordinal; n_streams >= 1; stream: a hipStream_t to launch on (NULL = the
   context creates its own non-blocking stream).  State starts all-zero (rnnoise_init,
   denoise.cpp:259-280). */
pn_ctx *pn_ctx_create(const pn_model *model, int device, int n_streams, int nn_mode, void *hip_stream);
void pn_ctx_destroy(pn_ctx *ctx);
int pn_ctx_reset(pn_ctx *ctx);                       /* zero all stream state, frame counter = 0 */
/* Per-stream lifecycle: put the n streams ids[0..n) (host array, each in [0, n_streams), duplicates allowed) back into
   the state rnnoise_init leaves ONE DenoiseState in (denoise.cpp:259-280: all-zero DSP and network state) while every
   other stream of the context keeps its state and the context keeps its frame counter — the batched counterpart of
   rnnoise_destroy + rnnoise_create for a slot whose call has ended and whose next call begins.  Asynchronous on the
   context's stream: it takes effect between the frames submitted before and after it (also on the pipelined host path).
   The first frame processed after it is that stream's frame 0 (its first output frame is the one main.cpp:37 skips). */
int pn_ctx_reset_streams(pn_ctx *ctx, const int32_t *ids, int n);
int pn_ctx_n_streams(const pn_ctx *ctx);
int64_t pn_ctx_frames_done(const pn_ctx *ctx);
size_t pn_ctx_device_bytes(const pn_ctx *ctx);       /* HBM footprint of state + weights (weights only if this context created their device copy) */
/* Bytes of the packed weight copy this context reads — its own or one shared with other contexts of the same model content,
   device and network mode (pn_ctx_describe: weights=own|shared).  A shared copy outlives its creator while any user lives,
   and is then reported by no context's pn_ctx_device_bytes: add it once per distinct copy when summing a process. */
size_t pn_ctx_weight_bytes(const pn_ctx *ctx);
/* Which kernel families this context launches (chosen at creation from its batch size and nn_mode), as a
   NUL-terminated "key=value ..." string, e.g. "nn=mfma_f32 dense=batch gru=batch gru_rb=batch narrow=n16 frontend=split".
   gru / gru_rb: small | batch | direct_rows32 | direct_rows64 (fp32 mode from 24 576 streams: the GRU steps read fragment-order
   fp32 shadows of their inputs, +19 KB of device memory per stream, results bit-identical) | x3_* / f16_* in the shadow-operand
   modes; narrow: n16 | batch | small | fc_gb:n48+fc_rb:batch.  Returns the length written (excluding the NUL) or -1. */
int pn_ctx_describe(const pn_ctx *ctx, char *buf, size_t buf_bytes);

/* Advance every stream by one frame.  Device-resident buffers, asynchronous on the context's
   stream.  in: [n_streams][480]; out: [n_streams][480]; gr (optional, may be NULL):
   [n_streams][68] = g[34] | r[34], the reference's feature_test.raw tap (denoise.cpp:533-534).
   f32 = the rnnoise_process_frame sample convention (nominal [-1,1));
   i16 = the CLI convention (main.cpp:34,36): in/32768.f, out = trunc(x*32768) wrapped to 16 bit.
   in and out may alias.
   Ordering is the caller's: the launches only see what is complete on the context's stream.  A
   context created with hip_stream = NULL runs on its own NON-BLOCKING stream, which does not
   synchronise with the null stream or with any other stream: buffers filled by another stream
   (or by hipMemcpyAsync) must be ordered first (hipStreamWaitEvent / a synchronise), and the
   outputs consumed after pn_ctx_synchronize or an event on that stream. */
int pn_process_f32(pn_ctx *ctx, const float *d_in, float *d_out, float *d_gr);
int pn_process_i16(pn_ctx *ctx, const int16_t *d_in, int16_t *d_out, float *d_gr);
/* Per-call ACTIVE SET.  In the reference a stream's state advances only when ITS rnnoise_process_frame is called
   (src/denoise.cpp:508-547, src/rnnoise.h:60); these calls advance only the n streams listed in ids[] (host array,
   distinct ids, any order).  Every other stream keeps all of its state — history, look-ahead, pitch memory, synthesis
   memory, conv FIFOs, GRU states — bit for bit as if the call had not happened for it: its row of d_in is ignored, its rows
   of d_out and d_gr are left untouched, and when it is listed again it continues exactly like a reference stream that was
   only fed the frames it received.  (pn_ctx_read_features rows of a skipped stream are undefined for that tick.)
   n == n_streams is pn_process_*; the cost is paid per SKIPPED stream (two small launches over those rows: ~4 KB saved
   and ~52 KB of ring entries shifted per skipped stream-tick), nothing is added to an all-active call. */
int pn_process_f32_active(pn_ctx *ctx, const float *d_in, float *d_out, float *d_gr, const int32_t *ids, int n);
int pn_process_i16_active(pn_ctx *ctx, const int16_t *d_in, int16_t *d_out, float *d_gr, const int32_t *ids, int n);
/* The same on the pipelined host path (see pn_submit_host_* below): rows of h_in of skipped streams are ignored; their rows of
   h_out / h_gr are UNSPECIFIED for that frame (the caller knows which streams it listed).  A refused id list consumes no slot. */
int pn_submit_host_f32_active(pn_ctx *ctx, const float *h_in, float *h_out, float *h_gr, const int32_t *ids, int n);
int pn_submit_host_i16_active(pn_ctx *ctx, const int16_t *h_in, int16_t *h_out, float *h_gr, const int32_t *ids, int n);
/* Optional output stage: the reference's envelope post-filter
   (post_filtering, denoise.cpp:216-250), which it only runs on train()'s TEST synthesis (743),
   applied to the gains inside the back-end kernel between the g/r tap and pitch_filter — the same
   place.  Off by default (= rnnoise_process_frame exactly); the tap keeps the network's raw g.
   Takes effect from the next frame. */
int pn_ctx_set_postfilter(pn_ctx *ctx, int enable);
/* n_frames consecutive frames per call: in/out are [n_frames][n_streams][480] (frame-major). */
int pn_process_i16_multi(pn_ctx *ctx, const int16_t *d_in, int16_t *d_out, float *d_gr, int n_frames);
/* Host-buffer convenience wrappers (H2D, process, D2H, synchronise). */
int pn_process_host_f32(pn_ctx *ctx, const float *h_in, float *h_out, float *h_gr);
int pn_process_host_i16(pn_ctx *ctx, const int16_t *h_in, int16_t *h_out, float *h_gr);
/* Pipelined host-buffer entry points: the same work as pn_process_host_*, but the call returns
   once the frame is queued.  Copy-in, the launches and copy-out of consecutive frames overlap on
   three streams with double-buffered device staging, so a caller feeding frames back to back gets
   the device rate instead of the serial copy+compute+copy rate.  At most two frames are in
   flight: the call blocks until the frame submitted two calls earlier has been delivered.
   Lifetime: h_in must stay unmodified, and h_out / h_gr are undefined, until THAT frame is
   delivered — after pn_host_wait(ctx), or once the second next pn_submit_host_* call has
   returned.  Use pinned host memory (pn_host_alloc / hipHostMalloc / hipHostRegister): with
   pageable memory the runtime stages the copies and nothing overlaps.  Results are identical to
   pn_process_host_*; the two families may be mixed (the synchronous one drains the pipeline). */
int pn_submit_host_f32(pn_ctx *ctx, const float *h_in, float *h_out, float *h_gr);
int pn_submit_host_i16(pn_ctx *ctx, const int16_t *h_in, int16_t *h_out, float *h_gr);
int pn_host_wait(pn_ctx *ctx);               /* every submitted frame delivered */
/* Builds the three-stream pipeline NOW instead of inside the first pn_submit_host_* call: the copy streams are probed against
   the context's stream (a 1 ms sleeper kernel on it, up to 6 attempts x 3 pairings: tens of milliseconds; not legal while that
   stream is being captured).  A caller on a real-time clock calls this once before its first frame arrives. */
int pn_host_pipeline_prepare(pn_ctx *ctx);
/* Non-blocking: how many submitted frames have been DELIVERED (output copy complete) so far; -1 on error.  For callers on a
   real-time clock that timestamp each frame's delivery between arrivals (reference contract: src/main.cpp:30-39). */
int64_t pn_host_frames_delivered(pn_ctx *ctx);
/* How the two copy streams (host-to-device, device-to-host) of the pipelined path were obtained, one letter each: "n" a
   default-priority stream probed to share its hardware queue with neither the compute stream nor the other copy stream, "h" /
   "l" a high- / low-priority stream (the fallback); "" before the first pn_submit_host_* call.  Diagnostics (bench.py). */
const char *pn_ctx_pipe_streams(pn_ctx *ctx);
void *pn_host_alloc(size_t bytes);           /* pinned host memory (hipHostMalloc); NULL on failure */
/* One host feeding several GPUs: bind the CALLING THREAD to the CPUs of the NUMA node `device` hangs off (sysfs
   /sys/bus/pci/devices/<bdf>/numa_node) — call it in the thread that will own the device BEFORE pn_ctx_create / pn_host_alloc,
   so that first touch places its pinned buffers next to that GPU.  Returns the node (>= 0) when bound, -1 when the affinity
   was left alone; msg (optional) receives one line saying what was done or why not.  Never fatal.  (The reference's
   fan-out, utils/run.sh:49,65,99, places nothing.) */
int pn_bind_thread_to_device_numa(int device, char *msg, size_t msg_bytes);
void pn_host_free(void *p);
int pn_ctx_synchronize(pn_ctx *ctx);

/* Mid-pipeline taps for per-stage parity tests (device -> host copies, synchronising).
   features: [n_streams][70] of the last frame; silence: [n_streams] int32. */
int pn_ctx_read_features(pn_ctx *ctx, float *h_feat, int32_t *h_silence);
/* The same tap into caller-owned DEVICE buffers, asynchronous on the context's stream (either may be NULL). */
int pn_ctx_read_features_dev(pn_ctx *ctx, float *d_feat, int32_t *d_silence);
/* Run only the network on host-supplied features [n_streams][70] -> g,r [n_streams][68]: compute_rnn (rnn.cpp:42-81)
   on the context's RNN state.  Advances only the network's state (conv FIFOs, GRUs), like calling the reference's
   compute_rnn on an RNNState directly; the DSP state and frame counter of pn_process_* are untouched. */
int pn_ctx_compute_rnn_host(pn_ctx *ctx, const float *h_feat, float *h_gr);
/* Load / store the network state of every stream from / to host arrays in the reference's RNNState layout
   (nnet_data.h:28-38): conv1 [n_streams][4*128] and conv2 [n_streams][2*512] = the live part of the FIFOs, oldest
   frame first (nnet.cpp:191-199); gru1, gru2, gru3, gru_gb [n_streams][512]; gru_rb [n_streams][128].  NULL
   arrays are skipped.  Synchronous.  (Checkpoint/resume of the recurrent state, and what the exported
   compute_rnn(RNNState*, ...) uses.)  Available in every network mode: the fp16-operand and split-precision modes keep
   the fp32 values next to their operand shadows and re-derive the shadows on a load. */
int pn_ctx_set_rnn_state_host(pn_ctx *ctx, const float *conv1, const float *conv2, const float *gru1, const float *gru2,
                              const float *gru3, const float *gru_gb, const float *gru_rb);
int pn_ctx_get_rnn_state_host(pn_ctx *ctx, float *conv1, float *conv2, float *gru1, float *gru2, float *gru3,
                              float *gru_gb, float *gru_rb);

/* ---- per-kernel timing (HIP events on the context's stream) ------------------------------- */
/* When enabled, every launch of the named kernel families is bracketed by events. */
int pn_ctx_set_profiling(pn_ctx *ctx, int enable);
/* name: one of pn_kernel_name(i), i in [0, pn_kernel_count()).  Returns total milliseconds and
   launch count since the last pn_ctx_reset_profile (synchronises the stream). */
int pn_kernel_count(void);
const char *pn_kernel_name(int i);
int pn_ctx_kernel_time(pn_ctx *ctx, const char *name, double *total_ms, int64_t *launches);
int pn_ctx_reset_profile(pn_ctx *ctx);

/* Debug tap (tests/tools): copy an internal device buffer to the host; which = 0 feat, 1 c1ring,
   2 c2ring, 3 c2out, 4..7 gru1..gb (ping-pong pair), 8 rb, 9 g|r, 10 look-ahead spectra ring, 11 comb-filtered
   spectrum, 12 history ring, 13 the pitch period the last frame's comb filter used (int32 per stream).  Returns bytes copied
   or -1. */
long long pn_ctx_debug_copy(pn_ctx *ctx, int which, void *dst, long long max_bytes);
/* Launch-refusal hooks (tests).  A network launcher that is asked for a geometry its software pipeline cannot run returns
   an error WITHOUT launching and the frame fails: pn_process_* / pn_submit_host_* / pn_ctx_compute_rnn_host return -1 with
   pn_last_error() naming the launcher — never 0 with stale layer outputs.  (The context's stream state is undefined after
   a failed frame: pn_ctx_reset before reuse.)
   pn_debug_check_launch runs the launchers' geometry predicates without a GPU: kind 0 dense on the fp32 MFMA kernels,
   1 dense / 2 GRU (n_out neurons) on the shadow-operand kernels, 3 narrow dense on 16x16x4 tiles; n_panels panels of
   `width` columns.  0 = accepted, -1 = refused (pn_last_error()).
   pn_ctx_debug_inject_launch_failure(ctx, 1) makes every following frame of a non-STRICT context ask the fc layer's
   launcher for a refused geometry. */
int pn_debug_check_launch(int kind, int n_panels, int width, int n_out);
int pn_ctx_debug_inject_launch_failure(pn_ctx *ctx, int enable);
/* The digest function behind the shared-weights cache key (SHA-256, FIPS 180-4), exposed so that the CPU tests can check it
   against known answers: a model's packed device copy is shared by every context whose model has the same digest. */
void pn_debug_sha256(const void *data, size_t len, unsigned char out[32]);
/* SHA-256 of a model's content (arrays in storage order, then each layer's activation and reset_after as two int32). */
void pn_model_digest(const pn_model *model, unsigned char out[32]);

/* ---- batched training-feature generator ----------------------------------- */
/* The reference's `percepNet <speech> <noisy> <count> <output>` binary (train(), denoise.cpp:603-787,
   declared rnnoise.h:66) for n_pairs (speech, noisy) pairs in lock-step.  Samples are int16 at
   NORM_RATIO 1 (denoise.cpp:41,697: the float sample IS the int16 value).  One record = 138 float32:
   Ey_lookahead[34] | Ephaty[34] | T | pitch_corr | g[34] | r[34] (denoise.cpp:764-773), g being
   envelope-post-filtered as in the reference's default (TEST) build (45-47, 743).  The optional PCM is
   that build's test_output.pcm.  No model is involved. */
typedef struct pn_featgen pn_featgen;
pn_featgen *pn_featgen_create(int device, int n_pairs, void *hip_stream);
void pn_featgen_destroy(pn_featgen *fg);
int pn_featgen_reset(pn_featgen *fg);
int pn_featgen_n_pairs(const pn_featgen *fg);
int64_t pn_featgen_frames_done(const pn_featgen *fg);
size_t pn_featgen_device_bytes(const pn_featgen *fg);
int pn_featgen_synchronize(pn_featgen *fg);
/* One frame, device buffers, asynchronous: speech/noisy [n_pairs][480]; records [n_pairs][138];
   test_pcm (may be NULL) [n_pairs][480]. */
int pn_featgen_process_i16(pn_featgen *fg, const int16_t *d_speech, const int16_t *d_noisy, float *d_records,
                           int16_t *d_test_pcm);
/* n_frames frames, pair-major "file images": speech/noisy [n_pairs][n_frames][480], records
   [n_pairs][n_frames][138] (= each pair's output file), test_pcm [n_pairs][n_frames][480] or NULL. */
int pn_featgen_process_i16_files(pn_featgen *fg, const int16_t *d_speech, const int16_t *d_noisy, int n_frames,
                                 float *d_records, int16_t *d_test_pcm);
int pn_featgen_process_host_i16_files(pn_featgen *fg, const int16_t *h_speech, const int16_t *h_noisy, int n_frames,
                                      float *h_records, int16_t *h_test_pcm);
/* File-level driver with train()'s file semantics (whole frames cycled at EOF, 693-715) for n_jobs
   jobs at once; test_out_paths / test_in_paths may be NULL (or hold NULL entries). */
int pn_featgen_run_files(int device, int n_jobs, const char *const *speech_paths, const char *const *noisy_paths,
                         const int *counts, const char *const *out_paths, const char *const *test_out_paths,
                         const char *const *test_in_paths);

const char *pn_last_error(void);
const char *pn_version(void);
int pn_device_count(void);                   /* usable HIP devices (0 when there is none) */

/* ---- reference frame-engine interface, extern "C" spelling -------------------------------- */
/* (the C++-mangled rnnoise_* symbols with the reference's exact prototypes are exported too) */
typedef struct DenoiseState DenoiseState;
int rnnoise_get_size_c(void);
int rnnoise_init_c(DenoiseState *st, RNNModel *model);
DenoiseState *rnnoise_create_c(RNNModel *model);
void rnnoise_destroy_c(DenoiseState *st);
float rnnoise_process_frame_c(DenoiseState *st, float *out, const float *in, FILE *f_feature);
int rnnoise_train_c(int argc, char **argv);            /* train(), rnnoise.h:66 */
RNNModel *rnnoise_model_from_file_c(FILE *f);
void rnnoise_model_free_c(RNNModel *model);
/* compute_rnn (rnnoise.h:68, rnn.cpp:42-81), also exported under its C++-mangled name
   _Z11compute_rnnP8RNNStatePfS1_PKf: one network step on a caller-owned RNNState (host arrays); gains[34],
   strengths[34], input[70].  The state is uploaded to a cached batch-of-one context, advanced on the GPU (network mode
   PERCEPNET_STRICT=1|0, device PERCEPNET_DEVICE) and written back, so the caller sees the reference's semantics. */
void rnnoise_compute_rnn_c(RNNState *rnn, float *gains, float *strengths, const float *input);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
