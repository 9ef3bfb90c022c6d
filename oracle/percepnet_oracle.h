/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's per-frame hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * See percepnet_oracle.c for the reference file:line each function follows. */
#ifndef PERCEPNET_ORACLE_H
#define PERCEPNET_ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PNO_FRAME 480
#define PNO_WINDOW 960
#define PNO_FREQ 481
#define PNO_NB_BANDS 34
#define PNO_NB_FEATURES 70
#define PNO_COMB_BUF 5760
#define PNO_PITCH_BUF 1728

typedef struct pno_model pno_model;   /* parsed PNW1 blob (borrowed pointers into it) */
typedef struct pno_state pno_state;   /* one stream's DenoiseState equivalent */

/* tables (CommonState + erb_band): returns pointers to static storage */
void pno_tables(const float **twiddles_ri /*960*2*/, const short **bitrev /*960*/,
                const float **half_window /*480*/, const float **comb_hann /*7*/,
                const int **nfftborder /*34*/);
const float *pno_tansig_table(void);  /* 201 entries */

pno_model *pno_model_from_blob(const void *blob, size_t nbytes); /* blob must outlive the model */
void pno_model_free(pno_model *m);

pno_state *pno_create(const pno_model *m);
void pno_destroy(pno_state *st);
/* rnnoise_process_frame: 480 floats in -> 480 floats out (may alias), optional g/r tap [68] */
void pno_process_frame(pno_state *st, float *out, const float *in, float *gr68);
/* the percepNet_run loop on in-memory PCM (first output frame dropped) */
void pno_run_pcm(const pno_model *m, const short *pcm_in, int n_frames, short *pcm_out, float *gr);
/* optional envelope post-filter (denoise.cpp:216-250) between the g/r tap and pitch_filter */
void pno_set_postfilter(pno_state *st, int on);
void pno_run_pcm_pf(const pno_model *m, const short *pcm_in, int n_frames, short *pcm_out, float *gr, int postfilter);
void pno_run_float(const pno_model *m, const float *in, int n_frames, float *out, float *gr);
/* pno_run_pcm for n_streams streams on n_threads host threads, `group` streams sharing each sweep over the
   weights (same per-stream arithmetic, bit-identical results; pinned by tests/test_oracle.py).
   pcm_in [S][T*480]; pcm_out [S][(T-1)*480]; gr [S][T][68]; feat [S][T][70]; sil [S][T]; outputs may be NULL */
void pno_run_pcm_batch(const pno_model *m, const short *pcm_in, int n_streams, int n_frames, short *pcm_out,
                       float *gr, float *feat, int *sil, int group, int n_threads);

/* ---- stage functions (exported for per-stage parity tests) ---- */
void pno_fft960(const float *in_ri, float *out_ri);
void pno_band_energy(float *bandE, const float *X_ri);
void pno_band_corr(float *bandE, const float *X_ri, const float *P_ri);
void pno_interp_band_gain(float *g481, const float *bandE); /* bins >=400 left untouched */
void pno_pitch_downsample(const float *x1728, float *x_lp864);
void pno_pitch_search(const float *buf864, int *pitch, float *corr);
float pno_remove_doubling(const float *buf864, int *T0, int prev_period, float prev_gain);
void pno_compute_rnn(pno_state *st, float *g, float *r, const float *feat70);
/* features of one frame without running the NN: returns silence flag; feat70 written */
int pno_frame_features(pno_state *st, const float *in, float *feat70);
/* per-stage taps of one frame, DSP only (no NN): X, P, Y [481][2]; Ex, Ep, Exp, Ey [34]; any pointer may be NULL */
int pno_frame_stages(pno_state *st, const float *in, float *X_ri, float *P_ri, float *Y_ri, float *Ex, float *Ep,
                     float *Exp, float *Ey, int *period, float *feat70);
void pno_state_comb_buf(const pno_state *st, float *dst5760);
float pno_tansig(float x);
float pno_sigmoid(float x);
void pno_dense(const float *bias, const float *w, int nin, int nn, int act, float *out, const float *in);
void pno_conv1d(const float *bias, const float *w, int nin, int ks, int nn, int act, float *out, float *mem, const float *in);
void pno_gru(const float *bias, const float *w, const float *rw, int nin, int nn, int act, float *state, const float *in);


/* ---- the `percepNet` training-feature binary (train(), denoise.cpp:603-787) ---- */
typedef struct pno_train pno_train;
pno_train *pno_train_create(void);
void pno_train_destroy(pno_train *tr);
void pno_train_frame(pno_train *tr, const float *x480, const float *n480, float *out138, float *test_out480);
void pno_train_run(const short *speech, const short *noisy, int count, float *out138xcount, short *test_out);

#ifdef __cplusplus
}
#endif
#endif
