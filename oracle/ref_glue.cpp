/* TEST INFRASTRUCTURE — never linked into the product.
 *
 * Glue that turns the UNTOUCHED reference sources (compiled where they lie under
 * /root/reference/src by oracle/Makefile) into a ctypes-callable shared object,
 * oracle/_ref/libpercepnet_ref.so.  Nothing from the reference is copied: this file only
 *   (1) supplies the one symbol the reference needs and does not ship,
 *       `percepnet_model_orig` (declared denoise.cpp:49-51, normally generated into the absent
 *       src/nnet_data.cpp by dump_percepnet.py:128-155), populated at run time from a PNW1 blob
 *       (percepnet_amd/weights.py) instead of 180 MB of C text, and
 *   (2) exports extern "C" trampolines around the reference's C++-mangled entry points
 *       (rnnoise.h:49-68) and a few non-static stage functions for per-stage parity tests.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "rnnoise.h"   /* from -I/root/reference/src */
#include "kiss_fft.h"
#include "pitch.h"
#include "celt_lpc.h"

static DenseLayer  l_fc, l_fc_gb, l_fc_rb;
static Conv1DLayer l_conv1, l_conv2;
static GRULayer    l_gru1, l_gru2, l_gru3, l_gru_gb, l_gru_rb;
static float *g_blob_copy = NULL;

extern const RNNModel percepnet_model_orig = {
  &l_fc, &l_conv1, &l_conv2, &l_gru1, &l_gru2, &l_gru3, &l_gru_gb, &l_gru_rb, &l_fc_gb, &l_fc_rb
};

/* stage functions that are external in the reference but not declared in its headers */
void compute_band_energy(float *bandE, const kiss_fft_cpx *X);
void compute_band_corr(float *bandE, const kiss_fft_cpx *X, const kiss_fft_cpx *P);
void interp_band_gain(float *g, const float *bandE);
void pitch_filter(kiss_fft_cpx *X, const kiss_fft_cpx *P, const float *Ex, const float *Ep,
                  const float *Exp, const float *g, const float *r);

extern "C" {

/* Load a PNW1 blob (see percepnet_amd/weights.py:pack_blob). Returns 0 on success. */
int ref_load_weights(const void *blob, size_t nbytes) {
  const unsigned char *p = (const unsigned char*)blob;
  if (nbytes < 8 || memcmp(p, "PNW1", 4) != 0) return -1;
  uint32_t n; memcpy(&n, p+4, 4);
  if (n != 10) return -2;
  free(g_blob_copy);
  g_blob_copy = (float*)malloc(nbytes);
  memcpy(g_blob_copy, blob, nbytes);
  const unsigned char *q = (const unsigned char*)g_blob_copy;
  size_t off = 8;
  DenseLayer *dense[3] = {&l_fc, &l_fc_gb, &l_fc_rb};
  Conv1DLayer *conv[2] = {&l_conv1, &l_conv2};
  GRULayer *gru[5] = {&l_gru1, &l_gru2, &l_gru3, &l_gru_gb, &l_gru_rb};
  int nd = 0, nc = 0, ng = 0;
  for (uint32_t li = 0; li < n; li++) {
    uint32_t h[6]; memcpy(h, q+off, 24); off += 24;
    uint32_t kind=h[0], nin=h[1], nn=h[2], ks=h[3], act=h[4], ra=h[5];
    const float *bias = (const float*)(q+off);
    if (kind == 0) {
      off += 4*(size_t)nn; const float *w = (const float*)(q+off); off += 4*(size_t)nin*nn;
      DenseLayer *L = dense[nd++]; L->bias=bias; L->input_weights=w; L->nb_inputs=nin; L->nb_neurons=nn; L->activation=act;
    } else if (kind == 1) {
      off += 4*(size_t)nn; const float *w = (const float*)(q+off); off += 4*(size_t)nin*ks*nn;
      Conv1DLayer *L = conv[nc++]; L->bias=bias; L->input_weights=w; L->nb_inputs=nin; L->kernel_size=ks; L->nb_neurons=nn; L->activation=act;
    } else if (kind == 2) {
      off += 4*(size_t)6*nn; const float *w = (const float*)(q+off); off += 4*(size_t)nin*3*nn;
      const float *rw = (const float*)(q+off); off += 4*(size_t)nn*3*nn;
      GRULayer *L = gru[ng++]; L->bias=bias; L->input_weights=w; L->recurrent_weights=rw; L->nb_inputs=nin; L->nb_neurons=nn; L->activation=act; L->reset_after=ra;
    } else return -3;
  }
  return off == nbytes ? 0 : -4;
}

void *ref_create(void) { return rnnoise_create(NULL); }

void ref_destroy(void *st) {
  /* the reference leaks the calloc'd NN state (denoise.cpp:326-331); fine for tests */
  rnnoise_destroy((DenoiseState*)st);
}

int ref_get_size(void) { return rnnoise_get_size(); }

/* One frame through rnnoise_process_frame (denoise.cpp:508). The reference fwrite()s g,r
 * (34+34 floats) to f_feature unconditionally (533-534); capture them through fmemopen. */
void ref_process_frame(void *st, float *out, const float *in, float *gr68) {
  static FILE *f = NULL; static float buf[68+4]; /* slack: fmemopen NUL-terminates on flush */
  if (!f) { f = fmemopen(buf, sizeof(buf), "wb"); setvbuf(f, NULL, _IONBF, 0); }
  rewind(f);
  rnnoise_process_frame((DenoiseState*)st, out, in, f);
  fflush(f);
  if (gr68) memcpy(gr68, buf, 68*sizeof(float));
}

/* The percepNet_run loop (main.cpp:30-39) on in-memory PCM: n_frames*480 int16 in,
 * (n_frames-1)*480 int16 out (first output frame dropped, main.cpp:37), optional g/r tap
 * n_frames*68 floats (== ./feature_test.raw). */
void ref_run_pcm(const short *pcm_in, int n_frames, short *pcm_out, float *gr) {
  DenoiseState *st = rnnoise_create(NULL);
  float x[480]; float tap[68];
  for (int t = 0; t < n_frames; t++) {
    for (int i = 0; i < 480; i++) x[i] = ((float)pcm_in[t*480+i])/32768.f;
    ref_process_frame(st, x, x, tap);
    if (gr) memcpy(gr + (size_t)t*68, tap, sizeof(tap));
    if (t > 0) for (int i = 0; i < 480; i++) { short s = x[i]*32768; pcm_out[(size_t)(t-1)*480+i] = s; }
  }
  rnnoise_destroy(st);
}

/* float-in/float-out variant (no int16 quantisation) for ULP-level comparisons */
void ref_run_float(const float *in, int n_frames, float *out, float *gr) {
  DenoiseState *st = rnnoise_create(NULL);
  float tap[68];
  for (int t = 0; t < n_frames; t++) {
    ref_process_frame(st, out + (size_t)t*480, in + (size_t)t*480, tap);
    if (gr) memcpy(gr + (size_t)t*68, tap, sizeof(tap));
  }
  rnnoise_destroy(st);
}

/* ---- per-stage taps (all are external symbols of the reference objects) ---- */
void ref_fft960(const float *in_ri, float *out_ri) {          /* kiss_fft.cpp:566 */
  static kiss_fft_state *k = NULL;
  if (!k) k = opus_fft_alloc_twiddles(960, NULL, NULL, NULL, 0);
  opus_fft_c(k, (const kiss_fft_cpx*)in_ri, (kiss_fft_cpx*)out_ri);
}
void ref_fft_tables(float *tw_ri, short *bitrev, short *factors) {
  kiss_fft_state *k = opus_fft_alloc_twiddles(960, NULL, NULL, NULL, 0);
  memcpy(tw_ri, k->twiddles, 960*2*sizeof(float));
  memcpy(bitrev, k->bitrev, 960*sizeof(short));
  memcpy(factors, k->factors, 16*sizeof(short));
}
void ref_band_energy(float *bandE, const float *X_ri) { compute_band_energy(bandE, (const kiss_fft_cpx*)X_ri); }
void ref_band_corr(float *bandE, const float *X_ri, const float *P_ri) { compute_band_corr(bandE, (const kiss_fft_cpx*)X_ri, (const kiss_fft_cpx*)P_ri); }
void ref_interp_band_gain(float *g481, const float *bandE) { interp_band_gain(g481, bandE); }
void ref_pitch_downsample(const float *x1728, float *x_lp864) {  /* pitch.cpp:148 as called denoise.cpp:405 */
  float *pre[1]; pre[0] = (float*)x1728;
  pitch_downsample(pre, x_lp864, 1728, 1);
}
void ref_pitch_search(float *buf864, int *pitch, float *corr) {  /* denoise.cpp:406 */
  pitch_search(buf864 + 384, buf864, 960, 588, pitch, corr);
}
float ref_remove_doubling(float *buf864, int *T0, int prev_period, float prev_gain) { /* denoise.cpp:410 */
  return remove_doubling(buf864, 768, 60, 960, T0, prev_period, prev_gain);
}
void ref_compute_rnn_new(void **pst) { *pst = rnnoise_create(NULL); }
/* compute_rnn (rnn.cpp:42) on the NN state embedded in a DenoiseState created by ref_create:
 * the RNNState is the last member of DenoiseState (denoise.cpp:71-85). */
void ref_compute_rnn(void *st, float *g, float *r, const float *feat70) {
  RNNState *rnn = (RNNState*)((char*)st + rnnoise_get_size() - sizeof(RNNState));
  compute_rnn(rnn, g, r, feat70);
}
/* toy-shape layer kernels for the nnet_data_test.h known answers (tests/testnnet.cpp:19-66) */
void ref_dense(const float *bias, const float *w, int nin, int nn, int act, float *out, const float *in) {
  DenseLayer L = {bias, w, nin, nn, act}; compute_dense(&L, out, in);
}
void ref_conv1d(const float *bias, const float *w, int nin, int ks, int nn, int act, float *out, float *mem, const float *in) {
  Conv1DLayer L = {bias, w, nin, ks, nn, act}; compute_conv1d(&L, out, mem, in);
}
void ref_gru(const float *bias, const float *w, const float *rw, int nin, int nn, int act, float *state, const float *in) {
  GRULayer L = {bias, w, rw, nin, nn, act, 1}; compute_gru(&L, state, in);
}

/* The `percepNet` training binary's body (denoise.cpp:603-787) with its own argv contract
 * (<speech> <noisy> <count> <output>); it also drops test_input.pcm/test_output.pcm into the cwd. */
int ref_train(const char *speech, const char *noisy, const char *count, const char *output) {
  char *argv[5] = {(char*)"percepNet", (char*)speech, (char*)noisy, (char*)count, (char*)output};
  return train(5, argv);
}

} /* extern "C" */
