/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of jzi040941/PercepNet's inference hot path
 * (rnnoise_process_frame as driven by percepNet_run).  This is the checker, never the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning: the reference's own tests pin only the three toy NN kernels
 * (tests/testnnet.cpp:19-66 + tests/nnet_data_test.h); everything else is pinned by executing
 * the compiled reference (oracle/_ref, built from the untouched sources by oracle/Makefile):
 * tests/test_oracle.py requires this file to be BIT-EXACT against it, per stage and
 * end to end (PCM and g/r tap), and tests/golden/ holds vectors produced by that build.
 *
 * Arithmetic contract: every float operation below is a separately rounded IEEE binary32 op in
 * the order the reference's source evaluates it (README build: g++ -O3, x86-64 SSE2, no FMA,
 * no reassociation); the "double islands"  are evaluated in double.  Build with
 * -ffp-contract=off (oracle/Makefile).
 *
 * State is de-duplicated relative to DenoiseState (denoise.cpp:71-85): pitch_buf ==
 * comb_buf[1632,3360) and analysis_mem == comb_buf[2400,2880) after the per-frame shift
 * (SURVEY A.2), so only comb_buf + synthesis_mem + the NN state are kept — the same layout the
 * HIP path uses, which makes this file the stage-by-stage template for the kernels.
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "percepnet_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define FRAME 480
#define WINDOW 960
#define FREQ 481
#define NB 34
#define COMB_BUF 5760
#define PITCH_MAX 768
#define PITCH_MIN 60
#define COMB_M 3
#define NFFT 960

typedef struct { float r, i; } cpx;

/* ------------------------------------------------------------------ tables */
static int g_init;
static cpx g_tw[NFFT];            /* compute_twiddles kiss_fft.cpp:406-421 */
static short g_bitrev[NFFT];      /* compute_bitrev_table kiss_fft.cpp:315-345, factors 5,3,4,4,4 */
static float g_half_window[FRAME];/* check_init denoise.cpp:191-192 */
static float g_comb_hann[2*COMB_M+1]; /* denoise.cpp:200-206 */
static int g_border[NB];          /* ERBBand::nfftborder erbband.h:63-75 */

/* erbband.h:56-61 — float in, double math, float out */
static float freq2erb(float f) { return 9.265 * log(1 + f / (24.7 * 9.265)); }
static float erb2freq(float e) { return 24.7 * 9.265 * (exp(e / 9.265) - 1); }

static void init_tables(void) {
  int i, k;
  if (g_init) return;
  for (i = 0; i < NFFT; i++) {
    const double pi = 3.14159265358979323846264338327;
    double phase = (-2 * pi / NFFT) * i;
    g_tw[i].r = (float)cos(phase);
    g_tw[i].i = (float)sin(phase);
  }
  /* digit reversal: input index n = n0 + 5*(n1 + 3*(n2 + 4*(n3 + 4*n4))) lands at
     n0*192 + n1*64 + n2*16 + n3*4 + n4 (what compute_bitrev_table's recursion produces for
     factors {5,192, 3,64, 4,16, 4,4, 4,1}) */
  for (i = 0; i < NFFT; i++) {
    int n = i, n0 = n % 5; n /= 5;
    int n1 = n % 3; n /= 3;
    int n2 = n % 4; n /= 4;
    int n3 = n % 4; n /= 4;
    g_bitrev[i] = (short)(n0 * 192 + n1 * 64 + n2 * 16 + n3 * 4 + n);
  }
  for (i = 0; i < FRAME; i++)
    g_half_window[i] = sin(.5 * M_PI * sin(.5 * M_PI * (i + .5) / FRAME) * sin(.5 * M_PI * (i + .5) / FRAME));
  {
    float temp_sum = 0;
    for (i = 1; i < COMB_M * 2 + 2; i++) {
      g_comb_hann[i - 1] = 0.5 - 0.5 * cos(2.0 * M_PI * i / (COMB_M * 2 + 2));
      temp_sum += g_comb_hann[i - 1];
    }
    for (i = 1; i < COMB_M * 2 + 2; i++) g_comb_hann[i - 1] /= temp_sum;
  }
  /* ERBBand(960, 32, 0, 20000): linspace in float (erbband.h:6-32), cutoffs, borders */
  {
    const int N = NB - 2;
    float erb_low = freq2erb(0.f), erb_high = freq2erb(20000.f);
    float lims[NB], cut[NB];
    float num = (float)(N + 2);
    float delta = (erb_high - erb_low) / (num - 1);
    for (i = 0; i < N + 1; i++) lims[i] = erb_low + delta * i;
    lims[N + 1] = erb_high;
    for (i = 0; i < N + 2; i++) cut[i] = erb2freq(lims[i]);
    for (k = 0; k < N + 2; k++) g_border[k] = (int)((cut[k] + 25) / 50.f);
    for (k = 0; k < N; k++)
      if (g_border[k + 1] - g_border[k] < 2) g_border[k + 1] += (2 - (g_border[k + 1] - g_border[k]));
  }
  g_init = 1;
}

void pno_tables(const float **tw, const short **bitrev, const float **hw, const float **ch, const int **border) {
  init_tables();
  if (tw) *tw = (const float *)g_tw;
  if (bitrev) *bitrev = g_bitrev;
  if (hw) *hw = g_half_window;
  if (ch) *ch = g_comb_hann;
  if (border) *border = g_border;
}

/* ------------------------------------------------------------------ FFT */
/* opus_fft_c (kiss_fft.cpp:566-586) for nfft=960: scale by 1/960 + digit-reverse scatter, then
   opus_fft_impl (518-564) runs radix-4 (m=1), radix-4 (m=4), radix-4 (m=16), radix-3 (m=64),
   radix-5 (m=192).  Butterflies of one stage are independent, so they are written here as
   flat loops over the butterfly index (the form a wavefront executes). */
#define CMUL(m, a, b) do { (m).r = (a).r*(b).r - (a).i*(b).i; (m).i = (a).r*(b).i + (a).i*(b).r; } while (0)

static void bfly4_m1(cpx *F) { /* kf_bfly4, degenerate m==1 branch kiss_fft.cpp:112-131 */
  cpx s0, s1;
  s0.r = F[0].r - F[2].r; s0.i = F[0].i - F[2].i;
  F[0].r += F[2].r; F[0].i += F[2].i;
  s1.r = F[1].r + F[3].r; s1.i = F[1].i + F[3].i;
  F[2].r = F[0].r - s1.r; F[2].i = F[0].i - s1.i;
  F[0].r += s1.r; F[0].i += s1.i;
  s1.r = F[1].r - F[3].r; s1.i = F[1].i - F[3].i;
  F[1].r = s0.r + s1.i; F[1].i = s0.i - s1.r;
  F[3].r = s0.r - s1.i; F[3].i = s0.i + s1.r;
}

static void bfly4(cpx *F, int m, const cpx *t1, const cpx *t2, const cpx *t3) { /* kiss_fft.cpp:144-165 */
  cpx s0, s1, s2, s3, s4, s5;
  CMUL(s0, F[m], *t1); CMUL(s1, F[2*m], *t2); CMUL(s2, F[3*m], *t3);
  s5.r = F[0].r - s1.r; s5.i = F[0].i - s1.i;
  F[0].r += s1.r; F[0].i += s1.i;
  s3.r = s0.r + s2.r; s3.i = s0.i + s2.i;
  s4.r = s0.r - s2.r; s4.i = s0.i - s2.i;
  F[2*m].r = F[0].r - s3.r; F[2*m].i = F[0].i - s3.i;
  F[0].r += s3.r; F[0].i += s3.i;
  F[m].r = s5.r + s4.i; F[m].i = s5.i - s4.r;
  F[3*m].r = s5.r - s4.i; F[3*m].i = s5.i + s4.r;
}

static void bfly3(cpx *F, int m, const cpx *t1, const cpx *t2, float epi3_i) { /* kiss_fft.cpp:202-226 */
  cpx s0, s1, s2, s3;
  CMUL(s1, F[m], *t1); CMUL(s2, F[2*m], *t2);
  s3.r = s1.r + s2.r; s3.i = s1.i + s2.i;
  s0.r = s1.r - s2.r; s0.i = s1.i - s2.i;
  F[m].r = F[0].r - s3.r * .5f; F[m].i = F[0].i - s3.i * .5f;
  s0.r *= epi3_i; s0.i *= epi3_i;
  F[0].r += s3.r; F[0].i += s3.i;
  F[2*m].r = F[m].r + s0.i; F[2*m].i = F[m].i - s0.r;
  F[m].r = F[m].r - s0.i; F[m].i = F[m].i + s0.r;
}

static void bfly5(cpx *F0, int m, const cpx *tw, int u, cpx ya, cpx yb) { /* kiss_fft.cpp:269-303, fstride=1 */
  cpx *F1 = F0 + m, *F2 = F0 + 2*m, *F3 = F0 + 3*m, *F4 = F0 + 4*m;
  cpx s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12;
  s0 = *F0;
  CMUL(s1, *F1, tw[u]); CMUL(s2, *F2, tw[2*u]); CMUL(s3, *F3, tw[3*u]); CMUL(s4, *F4, tw[4*u]);
  s7.r = s1.r + s4.r; s7.i = s1.i + s4.i;
  s10.r = s1.r - s4.r; s10.i = s1.i - s4.i;
  s8.r = s2.r + s3.r; s8.i = s2.i + s3.i;
  s9.r = s2.r - s3.r; s9.i = s2.i - s3.i;
  F0->r = F0->r + (s7.r + s8.r);
  F0->i = F0->i + (s7.i + s8.i);
  s5.r = s0.r + (s7.r*ya.r + s8.r*yb.r);
  s5.i = s0.i + (s7.i*ya.r + s8.i*yb.r);
  s6.r = s10.i*ya.i + s9.i*yb.i;
  s6.i = -(s10.r*ya.i + s9.r*yb.i);
  F1->r = s5.r - s6.r; F1->i = s5.i - s6.i;
  F4->r = s5.r + s6.r; F4->i = s5.i + s6.i;
  s11.r = s0.r + (s7.r*yb.r + s8.r*ya.r);
  s11.i = s0.i + (s7.i*yb.r + s8.i*ya.r);
  s12.r = s9.i*ya.i - s10.i*yb.i;
  s12.i = s10.r*yb.i - s9.r*ya.i;
  F2->r = s11.r + s12.r; F2->i = s11.i + s12.i;
  F3->r = s11.r - s12.r; F3->i = s11.i - s12.i;
}

static void fft960(const cpx *in, cpx *out) {
  int b;
  const float scale = 1.f / NFFT; /* kiss_fft.cpp:459 */
  init_tables();
  for (b = 0; b < NFFT; b++) { out[g_bitrev[b]].r = scale * in[b].r; out[g_bitrev[b]].i = scale * in[b].i; }
  for (b = 0; b < 240; b++) bfly4_m1(out + 4*b);                       /* m=1,  N=240 */
  for (b = 0; b < 240; b++) { int i = b / 4,  j = b % 4;               /* m=4,  N=60, mm=16, fstride=60 */
    bfly4(out + i*16 + j, 4,  &g_tw[j*60],  &g_tw[2*j*60],  &g_tw[3*j*60]); }
  for (b = 0; b < 240; b++) { int i = b / 16, j = b % 16;              /* m=16, N=15, mm=64, fstride=15 */
    bfly4(out + i*64 + j, 16, &g_tw[j*15],  &g_tw[2*j*15],  &g_tw[3*j*15]); }
  for (b = 0; b < 320; b++) { int i = b / 64, j = b % 64;              /* radix-3 m=64, N=5, mm=192, fstride=5 */
    bfly3(out + i*192 + j, 64, &g_tw[j*5], &g_tw[2*j*5], g_tw[5*64].i); }
  for (b = 0; b < 192; b++)                                            /* radix-5 m=192, N=1, fstride=1 */
    bfly5(out + b, 192, g_tw, b, g_tw[192], g_tw[384]);
}

void pno_fft960(const float *in_ri, float *out_ri) { fft960((const cpx *)in_ri, (cpx *)out_ri); }

/* ------------------------------------------------------------------ band ops */
/* compute_band_energy denoise.cpp:89-123 */
static void band_energy(float *bandE, const cpx *X) {
  int i, j; float sum[NB] = {0};
  for (i = 0; i < NB - 1; i++) {
    int band_size = g_border[i + 1] - g_border[i];
    for (j = 0; j < band_size; j++) {
      float tmp, frac = (float)j / band_size;
      tmp = X[g_border[i] + j].r * X[g_border[i] + j].r;
      tmp += X[g_border[i] + j].i * X[g_border[i] + j].i;
      sum[i] += (1 - frac) * tmp;
      sum[i + 1] += frac * tmp;
    }
  }
  sum[0] *= 2; sum[NB - 1] *= 2;
  for (i = 0; i < NB; i++) bandE[i] = sum[i];
}
/* compute_band_corr denoise.cpp:125-160 */
static void band_corr(float *bandE, const cpx *X, const cpx *P) {
  int i, j; float sum[NB] = {0};
  for (i = 0; i < NB - 1; i++) {
    int band_size = g_border[i + 1] - g_border[i];
    for (j = 0; j < band_size; j++) {
      float tmp, frac = (float)j / band_size;
      tmp = X[g_border[i] + j].r * P[g_border[i] + j].r;
      tmp += X[g_border[i] + j].i * P[g_border[i] + j].i;
      sum[i] += (1 - frac) * tmp;
      sum[i + 1] += frac * tmp;
    }
  }
  sum[0] *= 2; sum[NB - 1] *= 2;
  for (i = 0; i < NB; i++) bandE[i] = sum[i];
}
/* interp_band_gain denoise.cpp:162-182; its memset clears 481 BYTES only, every bin < 400 is
   then overwritten, bins >= 400 keep the caller's initialiser (0 at both call sites) */
static void interp_band_gain(float *g, const float *bandE) {
  int i, j;
  for (i = 0; i < NB - 1; i++) {
    int band_size = g_border[i + 1] - g_border[i];
    for (j = 0; j < band_size; j++) {
      float frac = (float)j / band_size;
      g[g_border[i] + j] = (1 - frac) * bandE[i] + frac * bandE[i + 1];
    }
  }
}
void pno_band_energy(float *e, const float *X) { init_tables(); band_energy(e, (const cpx *)X); }
void pno_band_corr(float *e, const float *X, const float *P) { init_tables(); band_corr(e, (const cpx *)X, (const cpx *)P); }
void pno_interp_band_gain(float *g, const float *e) { init_tables(); interp_band_gain(g, e); }

/* apply_window denoise.cpp:282-289 + forward_transform 291-304 (keep bins 0..480) */
static void window_fft(cpx *X /*481*/, const float *x960) {
  cpx a[WINDOW], b[WINDOW]; int i;
  init_tables();
  for (i = 0; i < FRAME; i++) {
    a[i].r = x960[i] * g_half_window[i]; a[i].i = 0;
    a[WINDOW - 1 - i].r = x960[WINDOW - 1 - i] * g_half_window[i]; a[WINDOW - 1 - i].i = 0;
  }
  fft960(a, b);
  for (i = 0; i < FREQ; i++) X[i] = b[i];
}

/* ------------------------------------------------------------------ pitch */
/* celt_inner_prod pitch.h:136-144 (xcorr_kernel pitch.h:53-117 produces the same j-ascending
   chain per lag, so both collapse to this) */
static float inner_prod(const float *x, const float *y, int N) {
  float xy = 0; int i;
  for (i = 0; i < N; i++) xy = xy + x[i] * y[i];
  return xy;
}

/* pitch_downsample pitch.cpp:148-216 with len=1728, C=1, including _celt_autocorr
   (celt_lpc.cpp:198-279, lag=4, n=864, no window) and _celt_lpc (celt_lpc.cpp:37-88, p=4) */
void pno_pitch_downsample(const float *x, float *x_lp) {
  int i, j, k; const int n = 864;
  float ac[5], lpc[4] = {0, 0, 0, 0}, lpc2[5], tmp = 1.0f;
  for (i = 1; i < n; i++) x_lp[i] = .5f * (.5f * (x[2*i - 1] + x[2*i + 1]) + x[2*i]);
  x_lp[0] = .5f * (.5f * (x[1]) + x[0]);
  /* autocorr: fastN = 860 via celt_pitch_xcorr(x,x,ac,860,5), then the 4-sample tail */
  for (k = 0; k <= 4; k++) ac[k] = inner_prod(x_lp, x_lp + k, 860);
  for (k = 0; k <= 4; k++) {
    float d = 0;
    for (i = k + 860; i < n; i++) d = d + x_lp[i] * x_lp[i - k];
    ac[k] += d;
  }
  ac[0] *= 1.0001f;
  for (i = 1; i <= 4; i++) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
  /* Levinson-Durbin */
  {
    float error = ac[0], r;
    if (ac[0] != 0) {
      for (i = 0; i < 4; i++) {
        float rr = 0;
        for (j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
        rr += ac[i + 1];
        r = -rr / (error + 0.00001);          /* double add + double divide, celt_lpc.cpp:61 */
        lpc[i] = r;
        for (j = 0; j < (i + 1) >> 1; j++) {
          float t1 = lpc[j], t2 = lpc[i - 1 - j];
          lpc[j] = t1 + r * t2;
          lpc[i - 1 - j] = t2 + r * t1;
        }
        error = error - (r * r) * error;
        if (error < .001f * ac[0]) break;
      }
    }
  }
  for (i = 0; i < 4; i++) { tmp = .9f * tmp; lpc[i] = lpc[i] * tmp; }
  lpc2[0] = lpc[0] + .8f;
  lpc2[1] = lpc[1] + .8f * lpc[0];
  lpc2[2] = lpc[2] + .8f * lpc[1];
  lpc2[3] = lpc[3] + .8f * lpc[2];
  lpc2[4] = .8f * lpc[3];
  /* celt_fir5 pitch.cpp:106-145, in place, zero initial memory */
  {
    float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (i = 0; i < n; i++) {
      float sum = x_lp[i];
      sum = sum + lpc2[0] * m0; sum = sum + lpc2[1] * m1; sum = sum + lpc2[2] * m2;
      sum = sum + lpc2[3] * m3; sum = sum + lpc2[4] * m4;
      m4 = m3; m3 = m2; m2 = m1; m1 = m0; m0 = x_lp[i];
      x_lp[i] = sum;
    }
  }
}

/* find_best_pitch pitch.cpp:46-104 (float instantiation) */
static void find_best_pitch(const float *xcorr, const float *y, int len, int max_pitch, int *best_pitch) {
  int i, j; float Syy = 1, best_num[2] = {-1, -1}, best_den[2] = {0, 0};
  best_pitch[0] = 0; best_pitch[1] = 1;
  for (j = 0; j < len; j++) Syy = Syy + y[j] * y[j];
  for (i = 0; i < max_pitch; i++) {
    if (xcorr[i] > 0) {
      float num, xcorr16 = xcorr[i];
      xcorr16 *= 1e-12f;
      num = xcorr16 * xcorr16;
      if (num * best_den[1] > best_num[1] * Syy) {
        if (num * best_den[0] > best_num[0] * Syy) {
          best_num[1] = best_num[0]; best_den[1] = best_den[0]; best_pitch[1] = best_pitch[0];
          best_num[0] = num; best_den[0] = Syy; best_pitch[0] = i;
        } else {
          best_num[1] = num; best_den[1] = Syy; best_pitch[1] = i;
        }
      }
    }
    Syy += y[i + len] * y[i + len] - y[i] * y[i];
    Syy = (1 > Syy) ? 1 : Syy;
  }
}

/* pitch_search pitch.cpp:283-386 as called at denoise.cpp:406: x_lp=buf+384, y=buf, len=960,
   max_pitch=588.  Also returns the fork-specific raw xcorr[best] (pitch.cpp:385). */
void pno_pitch_search(const float *buf, int *pitch, float *pitch_corr) {
  const float *x_lp = buf + 384, *y = buf;
  float x_lp4[240], y_lp4[387], xcorr[294];
  int i, j, best_pitch[2] = {0, 0}, offset;
  for (j = 0; j < 240; j++) x_lp4[j] = x_lp[2*j];
  for (j = 0; j < 387; j++) y_lp4[j] = y[2*j];
  for (i = 0; i < 147; i++) xcorr[i] = inner_prod(x_lp4, y_lp4 + i, 240);
  find_best_pitch(xcorr, y_lp4, 240, 147, best_pitch);
  for (i = 0; i < 294; i++) {
    float sum;
    xcorr[i] = 0;
    if (abs(i - 2*best_pitch[0]) > 2 && abs(i - 2*best_pitch[1]) > 2) continue;
    sum = inner_prod(x_lp, y + i, 480);
    xcorr[i] = (-1 > sum) ? -1 : sum;
  }
  find_best_pitch(xcorr, y, 480, 294, best_pitch);
  if (best_pitch[0] > 0 && best_pitch[0] < 294 - 1) {
    float a = xcorr[best_pitch[0] - 1], b = xcorr[best_pitch[0]], c = xcorr[best_pitch[0] + 1];
    if ((c - a) > .7f * (b - a)) offset = 1;
    else if ((a - c) > .7f * (b - c)) offset = -1;
    else offset = 0;
  } else offset = 0;
  *pitch = 2*best_pitch[0] - offset;
  *pitch_corr = xcorr[best_pitch[0]];
}

static float pitch_gain(float xy, float xx, float yy) { return xy / sqrtf(1 + xx * yy); } /* pitch.cpp:417-420 */

/* remove_doubling pitch.cpp:424-527 as called at denoise.cpp:410: maxperiod=768, minperiod=60, N=960 */
float pno_remove_doubling(const float *buf, int *T0_, int prev_period, float prev_gain) {
  static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};
  const int maxperiod = 384, minperiod = 30, minperiod0 = 60, N = 480;
  const float *x = buf + maxperiod;
  int k, i, T, T0, offset;
  float g, g0, pg, xy, xx, yy, xy2, xcorr[3], best_xy, best_yy, yy_lookup[385];
  *T0_ /= 2; prev_period /= 2;
  if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
  T = T0 = *T0_;
  xx = 0; xy = 0;
  for (i = 0; i < N; i++) { xx = xx + x[i] * x[i]; xy = xy + x[i] * x[i - T0]; } /* dual_inner_prod pitch.h:119 */
  yy_lookup[0] = xx; yy = xx;
  for (i = 1; i <= maxperiod; i++) {
    yy = yy + x[-i] * x[-i] - x[N - i] * x[N - i];
    yy_lookup[i] = (0 > yy) ? 0 : yy;
  }
  yy = yy_lookup[T0];
  best_xy = xy; best_yy = yy;
  g = g0 = pitch_gain(xy, xx, yy);
  for (k = 2; k <= 15; k++) {
    int T1, T1b; float g1, cont, thresh;
    T1 = (2*T0 + k) / (2*k);
    if (T1 < minperiod) break;
    if (k == 2) { if (T1 + T0 > maxperiod) T1b = T0; else T1b = T0 + T1; }
    else T1b = (2*second_check[k]*T0 + k) / (2*k);
    xy = 0; xy2 = 0;
    for (i = 0; i < N; i++) { xy = xy + x[i] * x[i - T1]; xy2 = xy2 + x[i] * x[i - T1b]; }
    xy = .5f * (xy + xy2);
    yy = .5f * (yy_lookup[T1] + yy_lookup[T1b]);
    g1 = pitch_gain(xy, xx, yy);
    if (abs(T1 - prev_period) <= 1) cont = prev_gain;
    else if (abs(T1 - prev_period) <= 2 && 5*k*k < T0) cont = .5f * prev_gain;
    else cont = 0;
    thresh = (.3f > .7f * g0 - cont) ? .3f : .7f * g0 - cont;
    if (T1 < 3*minperiod) thresh = (.4f > .85f * g0 - cont) ? .4f : .85f * g0 - cont;
    /* the reference's `else if (T1<2*minperiod)` arm is unreachable (pitch.cpp:496) */
    if (g1 > thresh) { best_xy = xy; best_yy = yy; T = T1; g = g1; }
  }
  best_xy = (0 > best_xy) ? 0 : best_xy;
  if (best_yy <= best_xy) pg = 1.0f; else pg = best_xy / (best_yy + 1);
  for (k = 0; k < 3; k++) xcorr[k] = inner_prod(x, x - (T + k - 1), N);
  if ((xcorr[2] - xcorr[0]) > .7f * (xcorr[1] - xcorr[0])) offset = 1;
  else if ((xcorr[0] - xcorr[2]) > .7f * (xcorr[1] - xcorr[2])) offset = -1;
  else offset = 0;
  if (pg > g) pg = g;
  *T0_ = 2*T + offset;
  if (*T0_ < minperiod0) *T0_ = minperiod0;
  return pg;
}

/* ------------------------------------------------------------------ network */
static float g_tansig[201];
static int g_tansig_init;
/* tansig_table.h:5-46 ("auto-generated by gen_tables") is (float)tanh(0.04*i) printed with
   "%f" (6 decimals) and read back as a float literal — reproduced literally here; the test
   checks all 201 entries against the reference header. */
const float *pno_tansig_table(void) {
  if (!g_tansig_init) {
    int i; char buf[32];
    for (i = 0; i <= 200; i++) {
      float t = (float)tanh(0.04 * i);
      snprintf(buf, sizeof(buf), "%f", (double)t);
      g_tansig[i] = (float)strtod(buf, NULL);
    }
    g_tansig_init = 1;
  }
  return g_tansig;
}
/* tansig_approx vec.h:53-70 */
float pno_tansig(float x) {
  const float *tab = pno_tansig_table();
  int i; float y, dy, sign = 1;
  if (x < 0) { x = -x; sign = -1; }
  i = (int)floorf(.5f + 25 * x);
  i = (0 > ((200 < i) ? 200 : i)) ? 0 : ((200 < i) ? 200 : i);
  x -= .04f * i;
  y = tab[i];
  dy = 1 - y * y;
  y = y + x * dy * (1 - y * x);
  return sign * y;
}
float pno_sigmoid(float x) { return .5f + .5f * pno_tansig(.5f * x); } /* vec.h:72-75 */

enum { ACT_LINEAR = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_RELU = 3 };
static void activation(float *o, int N, int act) { /* compute_activation nnet.cpp:74-103 */
  int i;
  if (act == ACT_SIGMOID) for (i = 0; i < N; i++) o[i] = pno_sigmoid(o[i]);
  else if (act == ACT_TANH) for (i = 0; i < N; i++) o[i] = pno_tansig(o[i]);
  else if (act == ACT_RELU) for (i = 0; i < N; i++) o[i] = o[i] < 0 ? 0 : o[i];
}
/* sgemv_accum nnet.cpp:59-72 / sgemv_accum16 vec.h:102-135: per output, j ascending, mul then add */
static void sgemv_accum(float *out, const float *w, int rows, int cols, int stride, const float *x) {
  int i, j;
  for (i = 0; i < rows; i++) {
    float acc = out[i];
    for (j = 0; j < cols; j++) acc = acc + w[j * stride + i] * x[j];
    out[i] = acc;
  }
}
void pno_dense(const float *bias, const float *w, int nin, int nn, int act, float *out, const float *in) {
  int i; /* compute_dense nnet.cpp:105-118 */
  for (i = 0; i < nn; i++) out[i] = bias[i];
  sgemv_accum(out, w, nn, nin, nn, in);
  activation(out, nn, act);
}
void pno_conv1d(const float *bias, const float *w, int nin, int ks, int nn, int act, float *out, float *mem, const float *in) {
  float tmp[1536]; int i; /* compute_conv1d nnet.cpp:182-200 */
  memcpy(tmp, mem, sizeof(float) * nin * (ks - 1));
  memcpy(tmp + nin * (ks - 1), in, sizeof(float) * nin);
  for (i = 0; i < nn; i++) out[i] = bias[i];
  sgemv_accum(out, w, nn, nin * ks, nn, tmp);
  activation(out, nn, act);
  memcpy(mem, tmp + nin, sizeof(float) * nin * (ks - 1));
}
void pno_gru(const float *b, const float *w, const float *rw, int M, int N, int act, float *state, const float *in) {
  float tmp[512], z[512], r[512], h[512]; int i; const int stride = 3 * N; /* compute_gru nnet.cpp:120-180, reset_after */
  for (i = 0; i < N; i++) z[i] = b[i];
  for (i = 0; i < N; i++) z[i] += b[3*N + i];
  sgemv_accum(z, w, N, M, stride, in);
  sgemv_accum(z, rw, N, N, stride, state);
  activation(z, N, ACT_SIGMOID);
  for (i = 0; i < N; i++) r[i] = b[N + i];
  for (i = 0; i < N; i++) r[i] += b[4*N + i];
  sgemv_accum(r, w + N, N, M, stride, in);
  sgemv_accum(r, rw + N, N, N, stride, state);
  activation(r, N, ACT_SIGMOID);
  for (i = 0; i < N; i++) h[i] = b[2*N + i];
  for (i = 0; i < N; i++) tmp[i] = b[5*N + i];
  sgemv_accum(tmp, rw + 2*N, N, N, stride, state);
  for (i = 0; i < N; i++) h[i] += tmp[i] * r[i];
  sgemv_accum(h, w + 2*N, N, M, stride, in);
  activation(h, N, act);
  for (i = 0; i < N; i++) h[i] = z[i] * state[i] + (1 - z[i]) * h[i];
  for (i = 0; i < N; i++) state[i] = h[i];
}

typedef struct { int kind, nin, nn, ks, act, reset_after; const float *bias, *w, *rw; } layer_t;
struct pno_model { layer_t L[10]; };
enum { L_FC, L_CONV1, L_CONV2, L_GRU1, L_GRU2, L_GRU3, L_GRU_GB, L_GRU_RB, L_FC_GB, L_FC_RB };

pno_model *pno_model_from_blob(const void *blob, size_t nbytes) {
  const unsigned char *p = (const unsigned char *)blob; size_t off = 8; uint32_t n, li;
  pno_model *m;
  if (nbytes < 8 || memcmp(p, "PNW1", 4) != 0) return NULL;
  memcpy(&n, p + 4, 4);
  if (n != 10) return NULL;
  m = (pno_model *)calloc(1, sizeof(*m));
  for (li = 0; li < n; li++) {
    uint32_t h[6]; layer_t *L = &m->L[li];
    memcpy(h, p + off, 24); off += 24;
    L->kind = h[0]; L->nin = h[1]; L->nn = h[2]; L->ks = h[3]; L->act = h[4]; L->reset_after = h[5];
    L->bias = (const float *)(p + off); off += 4 * (size_t)(L->kind == 2 ? 6 * L->nn : L->nn);
    L->w = (const float *)(p + off); off += 4 * (size_t)L->nin * L->ks * L->nn * (L->kind == 2 ? 3 : 1);
    if (L->kind == 2) { L->rw = (const float *)(p + off); off += 4 * (size_t)L->nn * 3 * L->nn; }
  }
  if (off != nbytes) { free(m); return NULL; }
  return m;
}
void pno_model_free(pno_model *m) { free(m); }

struct pno_state {
  const pno_model *m;
  float comb_buf[COMB_BUF];
  float synthesis_mem[FRAME];
  float last_gain; int last_period; float pitch_corr;
  float conv1_mem[4 * 128], conv2_mem[2 * 512];
  float gru1[512], gru2[512], gru3[512], gru_gb[512], gru_rb[128];
  int postfilter;                 /* optional output stage, see pno_set_postfilter */
};

pno_state *pno_create(const pno_model *m) {
  pno_state *st = (pno_state *)calloc(1, sizeof(*st)); /* rnnoise_init: all-zero state denoise.cpp:259-280 */
  st->m = m; init_tables();
  return st;
}
void pno_destroy(pno_state *st) { free(st); }

/* compute_rnn rnn.cpp:42-81 */
void pno_compute_rnn(pno_state *st, float *gains, float *strengths, const float *input) {
  const layer_t *L = st->m->L;
  float dense_out[128], c1[512], c2[512], gb_in[2560], rb_in[1024];
  pno_dense(L[L_FC].bias, L[L_FC].w, 70, 128, L[L_FC].act, dense_out, input);
  pno_conv1d(L[L_CONV1].bias, L[L_CONV1].w, 128, 5, 512, L[L_CONV1].act, c1, st->conv1_mem, dense_out);
  pno_conv1d(L[L_CONV2].bias, L[L_CONV2].w, 512, 3, 512, L[L_CONV2].act, c2, st->conv2_mem, c1);
  pno_gru(L[L_GRU1].bias, L[L_GRU1].w, L[L_GRU1].rw, 512, 512, L[L_GRU1].act, st->gru1, c2);
  pno_gru(L[L_GRU2].bias, L[L_GRU2].w, L[L_GRU2].rw, 512, 512, L[L_GRU2].act, st->gru2, st->gru1);
  pno_gru(L[L_GRU3].bias, L[L_GRU3].w, L[L_GRU3].rw, 512, 512, L[L_GRU3].act, st->gru3, st->gru2);
  pno_gru(L[L_GRU_GB].bias, L[L_GRU_GB].w, L[L_GRU_GB].rw, 512, 512, L[L_GRU_GB].act, st->gru_gb, st->gru3);
  memcpy(rb_in, st->gru3, 512 * 4); memcpy(rb_in + 512, c2, 512 * 4);
  pno_gru(L[L_GRU_RB].bias, L[L_GRU_RB].w, L[L_GRU_RB].rw, 1024, 128, L[L_GRU_RB].act, st->gru_rb, rb_in);
  memcpy(gb_in, c2, 2048); memcpy(gb_in + 512, st->gru1, 2048); memcpy(gb_in + 1024, st->gru2, 2048);
  memcpy(gb_in + 1536, st->gru3, 2048); memcpy(gb_in + 2048, st->gru_gb, 2048);
  pno_dense(L[L_FC_GB].bias, L[L_FC_GB].w, 2560, 34, L[L_FC_GB].act, gains, gb_in);
  pno_dense(L[L_FC_RB].bias, L[L_FC_RB].w, 128, 34, L[L_FC_RB].act, strengths, st->gru_rb);
}

/* ------------------------------------------------------------------ frame engine */
typedef struct { cpx X[FREQ], P[FREQ]; float Ex[NB], Ep[NB], Exp[NB]; int silence; } frame_ana;

/* compute_frame_features denoise.cpp:372-434 on the de-duplicated state */
static void frame_features(pno_state *st, frame_ana *a, const float *in) {
  float p[WINDOW], pbuf[864], pitch_corr, gain, E = 0;
  int i, k, pitch_index;
  memmove(st->comb_buf, st->comb_buf + FRAME, (COMB_BUF - FRAME) * sizeof(float));
  memcpy(st->comb_buf + COMB_BUF - FRAME, in, FRAME * sizeof(float));
  /* frame_analysis (333-346): window [analysis_mem | delayed frame] == comb_buf[2400,3360) */
  window_fft(a->X, st->comb_buf + 2400);
  band_energy(a->Ex, a->X);
  pno_pitch_downsample(st->comb_buf + 1632, pbuf);          /* pitch_buf == comb_buf[1632,3360) */
  pno_pitch_search(pbuf, &pitch_index, &pitch_corr);
  pitch_index = PITCH_MAX - pitch_index;
  gain = pno_remove_doubling(pbuf, &pitch_index, st->last_period, st->last_gain);
  st->last_period = pitch_index; st->last_gain = gain; st->pitch_corr = pitch_corr;
  for (i = 0; i < WINDOW; i++) p[i] = 0;
  for (k = -COMB_M; k < COMB_M + 1; k++)                    /* comb filter 416-422 */
    for (i = 0; i < WINDOW; i++)
      p[i] += st->comb_buf[2400 - pitch_index * k + i] * g_comb_hann[k + COMB_M];
  window_fft(a->P, p);
  band_energy(a->Ep, a->P);
  band_corr(a->Exp, a->X, a->P);
  for (i = 0; i < NB; i++) a->Exp[i] = fmin(1, fmax(0, a->Exp[i] / sqrt(1e-15 + a->Ex[i] * a->Ep[i])));
  for (i = 0; i < NB; i++) E += a->Ex[i];
  a->silence = E < 0.1;
}

/* compute_lookahead_band_energy 498-506 + create_features 487-496 */
static void make_features(pno_state *st, const frame_ana *a, float *features) {
  cpx Y[FREQ]; float Ey[NB]; int i;
  window_fft(Y, st->comb_buf + COMB_BUF - WINDOW);
  band_energy(Ey, Y);
  for (i = 0; i < NB; i++) features[i] = Ey[i];
  for (i = 0; i < NB; i++) features[NB + i] = a->Exp[i];
  for (i = 0; i < 68; i++) features[i] = features[i] * 30;
  features[68] = (float)st->last_period / (PITCH_MAX - 3 * PITCH_MIN);
  features[69] = st->pitch_corr;
}

int pno_frame_features(pno_state *st, const float *in, float *feat70) {
  frame_ana a;
  frame_features(st, &a, in);
  make_features(st, &a, feat70);
  return a.silence;
}

/* Per-stage taps for the GPU per-stage parity tests: one frame of compute_frame_features
   (denoise.cpp:372-434) + compute_lookahead_band_energy (498-506) + create_features (487-496), DSP only — the DSP state
   (comb_buf, last_period, last_gain) never depends on the network, so a state driven through this function evolves like
   one driven through pno_process_frame.  X = spectrum of the frame being enhanced (frame_analysis 333-346 through
   kiss_fft.cpp:566-586), P = spectrum of the comb-filtered signal (416-427), Y = look-ahead spectrum; all [481] re,im.
   Any output pointer may be NULL.  Returns the silence flag. */
int pno_frame_stages(pno_state *st, const float *in, float *X_ri, float *P_ri, float *Y_ri, float *Ex, float *Ep,
                     float *Exp, float *Ey, int *period, float *feat70) {
  frame_ana a; cpx Y[FREQ]; float ey[NB], f[70];
  frame_features(st, &a, in);
  make_features(st, &a, f);
  window_fft(Y, st->comb_buf + COMB_BUF - WINDOW);
  band_energy(ey, Y);
  if (X_ri) memcpy(X_ri, a.X, sizeof(a.X));
  if (P_ri) memcpy(P_ri, a.P, sizeof(a.P));
  if (Y_ri) memcpy(Y_ri, Y, sizeof(Y));
  if (Ex) memcpy(Ex, a.Ex, sizeof(a.Ex));
  if (Ep) memcpy(Ep, a.Ep, sizeof(a.Ep));
  if (Exp) memcpy(Exp, a.Exp, sizeof(a.Exp));
  if (Ey) memcpy(Ey, ey, sizeof(ey));
  if (period) *period = st->last_period;
  if (feat70) memcpy(feat70, f, sizeof(f));
  return a.silence;
}
/* the stream's comb_buf (denoise.cpp:32: 5760 samples, oldest first) */
void pno_state_comb_buf(const pno_state *st, float *dst5760) { memcpy(dst5760, st->comb_buf, sizeof(st->comb_buf)); }

/* rnnoise_process_frame denoise.cpp:508-547 */
static void post_filtering(float *g, const float *Ey);

/* the reference's envelope post-filter (denoise.cpp:216-250) exists only on train()'s TEST
   synthesis (call site 743: post_filtering(g, Ey) between the gains and pitch_filter, Ey = band energy of the
   spectrum being enhanced).  As an optional inference stage it sits at the same place in rnnoise_process_frame:
   after compute_rnn and the g/r tap (531-534), before pitch_filter (536); the tap keeps the network's raw g. */
void pno_set_postfilter(pno_state *st, int on) { st->postfilter = on != 0; }

/* everything of rnnoise_process_frame after compute_rnn (denoise.cpp:531-546): the tap, pitch_filter, the gain,
   inverse transform and overlap-add.  Shared by the single-stream and the batched drivers. */
static void frame_finish(pno_state *st, frame_ana *a, float *g, const float *r, float *out) {
  float gf[FREQ], rf[FREQ], inv_r[NB];
  cpx x[WINDOW], y[WINDOW]; float t[WINDOW]; int i;
  if (st->postfilter) post_filtering(g, a->Ex);
  if (!a->silence) {                                         /* pitch_filter 436-485 */
    for (i = 0; i < FREQ; i++) rf[i] = 0;
    for (i = 0; i < NB; i++) inv_r[i] = 1 - r[i];
    interp_band_gain(rf, inv_r);
    for (i = 0; i < FREQ; i++) { a->X[i].r = rf[i] * a->X[i].r; a->X[i].i = rf[i] * a->X[i].i; }
    interp_band_gain(rf, r);
    for (i = 0; i < FREQ; i++) { a->X[i].r += rf[i] * a->P[i].r; a->X[i].i += rf[i] * a->P[i].i; }
  }
  for (i = 0; i < FREQ; i++) gf[i] = 0;                      /* gf[]={1} then bins<400 rewritten 517,539 */
  interp_band_gain(gf, g);
  for (i = 0; i < FREQ; i++) { a->X[i].r *= gf[i]; a->X[i].i *= gf[i]; }
  /* frame_synthesis 352-359 / inverse_transform 306-324: Hermitian extension, FORWARD fft,
     reversed read-out, x960 */
  for (i = 0; i < FREQ; i++) x[i] = a->X[i];
  for (; i < WINDOW; i++) { x[i].r = x[WINDOW - i].r; x[i].i = -x[WINDOW - i].i; }
  fft960(x, y);
  t[0] = WINDOW * y[0].r;
  for (i = 1; i < WINDOW; i++) t[i] = WINDOW * y[WINDOW - i].r;
  for (i = 0; i < FRAME; i++) { t[i] *= g_half_window[i]; t[WINDOW - 1 - i] *= g_half_window[i]; }
  for (i = 0; i < FRAME; i++) out[i] = t[i] + st->synthesis_mem[i];
  memcpy(st->synthesis_mem, t + FRAME, FRAME * sizeof(float));
}

void pno_process_frame(pno_state *st, float *out, const float *in, float *gr68) {
  frame_ana a; float features[70], g[NB], r[NB];
  frame_features(st, &a, in);
  make_features(st, &a, features);
  pno_compute_rnn(st, g, r, features);
  if (gr68) { memcpy(gr68, g, sizeof(g)); memcpy(gr68 + NB, r, sizeof(r)); }
  frame_finish(st, &a, g, r, out);
}

/* float -> short as the CLI's x86-64 build does it (main.cpp:36: cvttss2si then the low 16
   bits; out-of-int32-range and NaN give 0x80000000 -> 0) */
static short f2s(float v) {
  int32_t t = (fabsf(v) < 2147483648.f) ? (int32_t)v : INT32_MIN;
  return (short)(uint16_t)((uint32_t)t & 0xffffu);
}

/* main.cpp:30-39 */
void pno_run_pcm(const pno_model *m, const short *pcm_in, int n_frames, short *pcm_out, float *gr) {
  pno_run_pcm_pf(m, pcm_in, n_frames, pcm_out, gr, 0);
}
void pno_run_pcm_pf(const pno_model *m, const short *pcm_in, int n_frames, short *pcm_out, float *gr, int postfilter) {
  pno_state *st = pno_create(m); float x[FRAME]; int t, i;
  pno_set_postfilter(st, postfilter);
  for (t = 0; t < n_frames; t++) {
    for (i = 0; i < FRAME; i++) x[i] = ((float)pcm_in[(size_t)t * FRAME + i]) / 32768.f;
    pno_process_frame(st, x, x, gr ? gr + (size_t)t * 68 : NULL);
    if (t > 0) for (i = 0; i < FRAME; i++) pcm_out[(size_t)(t - 1) * FRAME + i] = f2s(x[i] * 32768);
  }
  pno_destroy(st);
}
void pno_run_float(const pno_model *m, const float *in, int n_frames, float *out, float *gr) {
  pno_state *st = pno_create(m); int t;
  for (t = 0; t < n_frames; t++)
    pno_process_frame(st, out + (size_t)t * FRAME, in + (size_t)t * FRAME, gr ? gr + (size_t)t * 68 : NULL);
  pno_destroy(st);
}


/* ================================================================== batched driver (test speed only)
 * pno_run_pcm for many streams at once.  The per-stream arithmetic is IDENTICAL to the single-stream
 * functions above — every output of every stream is produced by the same sequence of separately
 * rounded IEEE binary32 operations (bias, then j-ascending `acc = acc + w*x`, nnet.cpp:59-72) — only
 * the loop nest differs: a group of streams shares one sweep over the 32 MB of weights (the
 * single-stream loop is bound by re-streaming them every frame) and groups are spread over host
 * threads.  tests/test_oracle.py::test_batched_oracle_is_bit_identical pins this driver to
 * pno_run_pcm / pno_frame_features bit for bit; it exists so that the GPU parity tests can afford
 * BASELINE's sizes (1024 streams x 1000 frames). */
#include <pthread.h>

#define BG_MAX 32            /* streams per group (share a weight sweep) */
typedef float v8f __attribute__((vector_size(32), aligned(4), may_alias));

/* out[g][i] += sum_j w[j*stride+i]*x[g][j], j ascending, mul then add, for 4 streams x 16 outputs */
__attribute__((target_clones("avx2", "default")))
static void sgemv_accum_4x16(float *out, int ldo, const float *w, int cols, int stride, const float *x, int ldx) {
  v8f a00 = *(v8f *)(out), a01 = *(v8f *)(out + 8);
  v8f a10 = *(v8f *)(out + ldo), a11 = *(v8f *)(out + ldo + 8);
  v8f a20 = *(v8f *)(out + 2 * ldo), a21 = *(v8f *)(out + 2 * ldo + 8);
  v8f a30 = *(v8f *)(out + 3 * ldo), a31 = *(v8f *)(out + 3 * ldo + 8);
  int j;
  for (j = 0; j < cols; j++) {
    const v8f w0 = *(const v8f *)(w + (size_t)j * stride), w1 = *(const v8f *)(w + (size_t)j * stride + 8);
    const float x0 = x[j], x1 = x[ldx + j], x2 = x[2 * ldx + j], x3 = x[3 * ldx + j];
    a00 = a00 + w0 * x0; a01 = a01 + w1 * x0;
    a10 = a10 + w0 * x1; a11 = a11 + w1 * x1;
    a20 = a20 + w0 * x2; a21 = a21 + w1 * x2;
    a30 = a30 + w0 * x3; a31 = a31 + w1 * x3;
  }
  *(v8f *)(out) = a00; *(v8f *)(out + 8) = a01;
  *(v8f *)(out + ldo) = a10; *(v8f *)(out + ldo + 8) = a11;
  *(v8f *)(out + 2 * ldo) = a20; *(v8f *)(out + 2 * ldo + 8) = a21;
  *(v8f *)(out + 3 * ldo) = a30; *(v8f *)(out + 3 * ldo + 8) = a31;
}

static void sgemv_accum_b(float *out, int ldo, const float *w, int rows, int cols, int stride, const float *x, int ldx, int G) {
  int i0, g0, g, i, j;
  for (i0 = 0; i0 < rows; i0 += 16) {
    const int nb = rows - i0 < 16 ? rows - i0 : 16;
    for (g0 = 0; g0 < G; g0 += 4) {
      const int gb = G - g0 < 4 ? G - g0 : 4;
      if (nb == 16 && gb == 4) { sgemv_accum_4x16(out + (size_t)g0 * ldo + i0, ldo, w + i0, cols, stride, x + (size_t)g0 * ldx, ldx); continue; }
      for (g = g0; g < g0 + gb; g++)
        for (i = i0; i < i0 + nb; i++) {
          float acc = out[(size_t)g * ldo + i];
          for (j = 0; j < cols; j++) acc = acc + w[(size_t)j * stride + i] * x[(size_t)g * ldx + j];
          out[(size_t)g * ldo + i] = acc;
        }
    }
  }
}

static void dense_b(const layer_t *L, float *out, int ldo, const float *in, int ldi, int G) {
  int g, i;
  for (g = 0; g < G; g++) for (i = 0; i < L->nn; i++) out[(size_t)g * ldo + i] = L->bias[i];
  sgemv_accum_b(out, ldo, L->w, L->nn, L->nin, L->nn, in, ldi, G);
  for (g = 0; g < G; g++) activation(out + (size_t)g * ldo, L->nn, L->act);
}
/* compute_conv1d nnet.cpp:182-200; mem[g] points at stream g's FIFO */
static void conv1d_b(const layer_t *L, float *out, float *const *mem, const float *in, float *tmp /*[G][1536]*/, int G) {
  const int nin = L->nin, ks = L->ks, nn = L->nn; int g, i;
  for (g = 0; g < G; g++) {
    memcpy(tmp + (size_t)g * 1536, mem[g], sizeof(float) * nin * (ks - 1));
    memcpy(tmp + (size_t)g * 1536 + nin * (ks - 1), in + (size_t)g * nin, sizeof(float) * nin);
    for (i = 0; i < nn; i++) out[(size_t)g * nn + i] = L->bias[i];
  }
  sgemv_accum_b(out, nn, L->w, nn, nin * ks, nn, tmp, 1536, G);
  for (g = 0; g < G; g++) {
    activation(out + (size_t)g * nn, nn, L->act);
    memcpy(mem[g], tmp + (size_t)g * 1536 + nin, sizeof(float) * nin * (ks - 1));
  }
}
/* compute_gru nnet.cpp:120-180 (reset_after), statement for statement as pno_gru; state[g] = stream g's state */
static void gru_b(const layer_t *L, float *const *state, const float *in, int ldi, float *ws /*[5][G][512]*/, int G) {
  const int M = L->nin, N = L->nn, stride = 3 * N; const float *b = L->bias; int g, i;
  float *z = ws, *r = ws + (size_t)G * 512, *h = ws + (size_t)2 * G * 512, *tmp = ws + (size_t)3 * G * 512, *sc = ws + (size_t)4 * G * 512;
  for (g = 0; g < G; g++) {
    float *zg = z + (size_t)g * 512, *rg = r + (size_t)g * 512, *hg = h + (size_t)g * 512, *tg = tmp + (size_t)g * 512;
    memcpy(sc + (size_t)g * 512, state[g], sizeof(float) * N);
    for (i = 0; i < N; i++) zg[i] = b[i];
    for (i = 0; i < N; i++) zg[i] += b[3 * N + i];
    for (i = 0; i < N; i++) rg[i] = b[N + i];
    for (i = 0; i < N; i++) rg[i] += b[4 * N + i];
    for (i = 0; i < N; i++) hg[i] = b[2 * N + i];
    for (i = 0; i < N; i++) tg[i] = b[5 * N + i];
  }
  sgemv_accum_b(z, 512, L->w, N, M, stride, in, ldi, G);
  sgemv_accum_b(z, 512, L->rw, N, N, stride, sc, 512, G);
  sgemv_accum_b(r, 512, L->w + N, N, M, stride, in, ldi, G);
  sgemv_accum_b(r, 512, L->rw + N, N, N, stride, sc, 512, G);
  sgemv_accum_b(tmp, 512, L->rw + 2 * N, N, N, stride, sc, 512, G);
  for (g = 0; g < G; g++) {
    float *zg = z + (size_t)g * 512, *rg = r + (size_t)g * 512, *hg = h + (size_t)g * 512, *tg = tmp + (size_t)g * 512;
    activation(zg, N, ACT_SIGMOID);
    activation(rg, N, ACT_SIGMOID);
    for (i = 0; i < N; i++) hg[i] += tg[i] * rg[i];
  }
  sgemv_accum_b(h, 512, L->w + 2 * N, N, M, stride, in, ldi, G);
  for (g = 0; g < G; g++) {
    float *zg = z + (size_t)g * 512, *hg = h + (size_t)g * 512, *st = state[g];
    activation(hg, N, L->act);
    for (i = 0; i < N; i++) hg[i] = zg[i] * st[i] + (1 - zg[i]) * hg[i];
    for (i = 0; i < N; i++) st[i] = hg[i];
  }
}

typedef struct {
  float feat[BG_MAX][70], fc[BG_MAX][128], c1[BG_MAX][512], c2[BG_MAX][512], ctmp[BG_MAX][1536];
  float gru_in[BG_MAX][512], rb_in[BG_MAX][1024], gb_in[BG_MAX][2560], gws[5][BG_MAX][512], g[BG_MAX][NB], r[BG_MAX][NB];
  frame_ana ana[BG_MAX];
} batch_ws;

/* compute_rnn rnn.cpp:42-81 for G streams */
static void compute_rnn_b(pno_state *const *st, batch_ws *W, int G) {
  const layer_t *L = st[0]->m->L; float *ptr[BG_MAX]; int g;
  dense_b(&L[L_FC], &W->fc[0][0], 128, &W->feat[0][0], 70, G);
  for (g = 0; g < G; g++) ptr[g] = st[g]->conv1_mem;
  conv1d_b(&L[L_CONV1], &W->c1[0][0], ptr, &W->fc[0][0], &W->ctmp[0][0], G);
  for (g = 0; g < G; g++) ptr[g] = st[g]->conv2_mem;
  conv1d_b(&L[L_CONV2], &W->c2[0][0], ptr, &W->c1[0][0], &W->ctmp[0][0], G);
  for (g = 0; g < G; g++) ptr[g] = st[g]->gru1;
  gru_b(&L[L_GRU1], ptr, &W->c2[0][0], 512, &W->gws[0][0][0], G);
  for (g = 0; g < G; g++) { memcpy(W->gru_in[g], st[g]->gru1, 2048); ptr[g] = st[g]->gru2; }
  gru_b(&L[L_GRU2], ptr, &W->gru_in[0][0], 512, &W->gws[0][0][0], G);
  for (g = 0; g < G; g++) { memcpy(W->gru_in[g], st[g]->gru2, 2048); ptr[g] = st[g]->gru3; }
  gru_b(&L[L_GRU3], ptr, &W->gru_in[0][0], 512, &W->gws[0][0][0], G);
  for (g = 0; g < G; g++) { memcpy(W->gru_in[g], st[g]->gru3, 2048); ptr[g] = st[g]->gru_gb; }
  gru_b(&L[L_GRU_GB], ptr, &W->gru_in[0][0], 512, &W->gws[0][0][0], G);
  for (g = 0; g < G; g++) {
    memcpy(W->rb_in[g], st[g]->gru3, 2048); memcpy(W->rb_in[g] + 512, W->c2[g], 2048);
    ptr[g] = st[g]->gru_rb;
  }
  gru_b(&L[L_GRU_RB], ptr, &W->rb_in[0][0], 1024, &W->gws[0][0][0], G);
  for (g = 0; g < G; g++) {
    memcpy(W->gb_in[g], W->c2[g], 2048); memcpy(W->gb_in[g] + 512, st[g]->gru1, 2048); memcpy(W->gb_in[g] + 1024, st[g]->gru2, 2048);
    memcpy(W->gb_in[g] + 1536, st[g]->gru3, 2048); memcpy(W->gb_in[g] + 2048, st[g]->gru_gb, 2048);
    memcpy(W->gru_in[g], st[g]->gru_rb, 512);
  }
  dense_b(&L[L_FC_GB], &W->g[0][0], NB, &W->gb_in[0][0], 2560, G);
  dense_b(&L[L_FC_RB], &W->r[0][0], NB, &W->gru_in[0][0], 512, G);
}

typedef struct {
  const pno_model *m; const short *pcm_in; int S, T, G; short *pcm_out; float *gr, *feat; int *sil;
  int next_group;
} batch_job;

static void *batch_worker(void *arg) {
  batch_job *J = (batch_job *)arg;
  const int n_groups = (J->S + J->G - 1) / J->G, T = J->T;
  batch_ws *W = (batch_ws *)malloc(sizeof(batch_ws));
  pno_state *st[BG_MAX]; float x[FRAME]; int g, t, i;
  for (;;) {
    const int grp = __atomic_fetch_add(&J->next_group, 1, __ATOMIC_RELAXED);
    int s0, G;
    if (grp >= n_groups) break;
    s0 = grp * J->G; G = J->S - s0 < J->G ? J->S - s0 : J->G;
    for (g = 0; g < G; g++) st[g] = pno_create(J->m);
    for (t = 0; t < T; t++) {
      for (g = 0; g < G; g++) {
        const size_t s = (size_t)(s0 + g);
        const short *in = J->pcm_in + (s * T + t) * FRAME;
        for (i = 0; i < FRAME; i++) x[i] = ((float)in[i]) / 32768.f;          /* main.cpp:34 */
        frame_features(st[g], &W->ana[g], x);
        make_features(st[g], &W->ana[g], W->feat[g]);
        if (J->feat) memcpy(J->feat + (s * T + t) * 70, W->feat[g], 70 * sizeof(float));
        if (J->sil) J->sil[s * T + t] = W->ana[g].silence;
      }
      compute_rnn_b(st, W, G);
      for (g = 0; g < G; g++) {
        const size_t s = (size_t)(s0 + g);
        if (J->gr) { memcpy(J->gr + (s * T + t) * 68, W->g[g], NB * 4); memcpy(J->gr + (s * T + t) * 68 + NB, W->r[g], NB * 4); }
        frame_finish(st[g], &W->ana[g], W->g[g], W->r[g], x);
        if (t > 0 && J->pcm_out) {                                             /* main.cpp:36-38 */
          short *o = J->pcm_out + (s * (size_t)(T - 1) + (size_t)(t - 1)) * FRAME;
          for (i = 0; i < FRAME; i++) o[i] = f2s(x[i] * 32768);
        }
      }
    }
    for (g = 0; g < G; g++) pno_destroy(st[g]);
  }
  free(W);
  return NULL;
}

/* pcm_in [S][T*480]; pcm_out [S][(T-1)*480]; gr [S][T][68]; feat [S][T][70]; sil [S][T] (any output may be NULL) */
void pno_run_pcm_batch(const pno_model *m, const short *pcm_in, int n_streams, int n_frames, short *pcm_out,
                       float *gr, float *feat, int *sil, int group, int n_threads) {
  batch_job J; pthread_t th[512]; int i, n_groups;
  init_tables(); pno_tansig_table();        /* lazy tables built before the threads start */
  if (group < 1) group = 16;
  if (group > BG_MAX) group = BG_MAX;
  J.m = m; J.pcm_in = pcm_in; J.S = n_streams; J.T = n_frames; J.G = group; J.pcm_out = pcm_out; J.gr = gr; J.feat = feat; J.sil = sil;
  J.next_group = 0;
  n_groups = (n_streams + group - 1) / group;
  if (n_threads > n_groups) n_threads = n_groups;
  if (n_threads > 512) n_threads = 512;
  if (n_threads <= 1) { batch_worker(&J); return; }
  for (i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, batch_worker, &J);
  for (i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
}

/* ================================================================== training-feature generator
 * one iteration of the `percepNet` binary's train() loop, denoise.cpp:655-778,
 * as the reference's default build runs it (TEST defined at denoise.cpp:45-47, so the ideal gains
 * ARE envelope-post-filtered before they are written; the random gain/response block 673-691 and
 * the biquads 717-720 are commented out in the reference; NORM_RATIO 1, denoise.cpp:41). */
struct pno_train { pno_state *clean, *noisy; float pna, n0; };

pno_train *pno_train_create(void) {
  pno_train *tr = (pno_train *)calloc(1, sizeof(*tr)); int i;
  tr->clean = pno_create(NULL); tr->noisy = pno_create(NULL);
  tr->pna = 0;                                              /* denoise.cpp:207-210 */
  for (i = 1; i < COMB_M * 2 + 2; i++) tr->pna += g_comb_hann[i - 1] * g_comb_hann[i - 1];
  tr->n0 = 0.03;                                            /* denoise.cpp:211 */
  return tr;
}
void pno_train_destroy(pno_train *tr) { if (tr) { pno_destroy(tr->clean); pno_destroy(tr->noisy); free(tr); } }

/* post_filtering denoise.cpp:216-250 */
static void post_filtering(float *g, const float *Ey) {
  int i; float E0 = 0, E1 = 0, E_div, G, g_w[NB];
  for (i = 0; i < NB; i++) g_w[i] = g[i] * sinf(M_PI / 2 * g[i]);
  for (i = 0; i < NB; i++) E0 += g[i] * Ey[i];
  for (i = 0; i < NB; i++) E1 += g_w[i] * Ey[i];
  E_div = E0 / (E1 + 1e-6f);
  G = sqrtf(((1 + 0.02f) * E_div) / (1 + 0.02f * (E_div * E_div)));
  for (i = 0; i < NB; i++) g[i] = G * g_w[i];
}

/* x = speech frame, n = noisy frame (float(int16), NORM_RATIO 1).  out138 = the record train()
   appends to <output> (764-773).  test_out480 (may be NULL) = the TEST synthesis `out[]` (743-757)
   before the saturating short cast.  Float/double promotion follows the C++ overloads the
   reference resolves to (sqrt(float) -> float; pow(float,int) -> double). */
void pno_train_frame(pno_train *tr, const float *x, const float *n, float *out138, float *test_out480) {
  frame_ana Y, X; float g[NB], r[NB], Ephatp[NB], Ey_look[NB]; int i;
  const float *Ephaty = Y.Exp, *Exp = X.Exp;
  frame_features(tr->noisy, &Y, n);                         /* 730 */
  frame_features(tr->clean, &X, x);                         /* 731 */
  for (i = 0; i < NB; i++) {                                /* calc_ideal_gain 571-577 */
    g[i] = X.Ex[i] / (.0001 + Y.Ex[i]);
    if (g[i] > 1) g[i] = 1;
    if (g[i] < 0) g[i] = 0;
  }
  /* 733-734 compute Eyp = corr(Y, P) which nothing reads afterwards (filter_strength_calc is handed
     Ephaty in its place, 736): dead, skipped */
  for (i = 0; i < NB; i++)                                  /* estimate_phat_corr 549-553 */
    Ephatp[i] = Ephaty[i] / sqrt((1 - tr->pna) * pow(Ephaty[i], 2) + tr->pna);
  for (i = 0; i < NB; i++) {                                /* filter_strength_calc 555-569 */
    float a = Ephatp[i] * Ephatp[i] - Exp[i] * Exp[i], b, c, alpha;
    if (a < 0) a = 0;
    b = Ephatp[i] * Ephaty[i] * (1 - Exp[i] * Exp[i]);
    c = Exp[i] * Exp[i] - Ephaty[i] * Ephaty[i];
    if (c < 0) c = 0;
    alpha = (sqrtf(b * b + a * (c)) - b) / (a + 1e-8);
    r[i] = alpha / (1 + alpha);
  }
  for (i = 0; i < NB; i++)                                  /* adjust_gain_strength_by_condition 579-589 */
    if (Ephatp[i] < Exp[i]) {
      float g_att = sqrtf((1 + tr->n0 - Exp[i] * Exp[i]) / (1 + tr->n0 - Ephatp[i] * Ephatp[i]));
      r[i] = 0.99;
      g[i] *= g_att;
    }
  post_filtering(g, Y.Ex);                                  /* 743 (TEST) */
  {                                                         /* 744-757 (TEST): synthesis through st */
    float gf[FREQ], rf[FREQ], inv_r[NB], t[WINDOW]; cpx xx[WINDOW], yy[WINDOW];
    if (!Y.silence) {
      for (i = 0; i < FREQ; i++) rf[i] = 0;
      for (i = 0; i < NB; i++) inv_r[i] = 1 - r[i];
      interp_band_gain(rf, inv_r);
      for (i = 0; i < FREQ; i++) { Y.X[i].r = rf[i] * Y.X[i].r; Y.X[i].i = rf[i] * Y.X[i].i; }
      interp_band_gain(rf, r);
      for (i = 0; i < FREQ; i++) { Y.X[i].r += rf[i] * Y.P[i].r; Y.X[i].i += rf[i] * Y.P[i].i; }
    }
    for (i = 0; i < FREQ; i++) gf[i] = 0;
    interp_band_gain(gf, g);
    for (i = 0; i < FREQ; i++) { Y.X[i].r *= gf[i]; Y.X[i].i *= gf[i]; }
    for (i = 0; i < FREQ; i++) xx[i] = Y.X[i];
    for (; i < WINDOW; i++) { xx[i].r = xx[WINDOW - i].r; xx[i].i = -xx[WINDOW - i].i; }
    fft960(xx, yy);
    t[0] = WINDOW * yy[0].r;
    for (i = 1; i < WINDOW; i++) t[i] = WINDOW * yy[WINDOW - i].r;
    for (i = 0; i < FRAME; i++) { t[i] *= g_half_window[i]; t[WINDOW - 1 - i] *= g_half_window[i]; }
    if (test_out480) for (i = 0; i < FRAME; i++) test_out480[i] = t[i] + tr->clean->synthesis_mem[i];
    memcpy(tr->clean->synthesis_mem, t + FRAME, FRAME * sizeof(float));
  }
  {                                                         /* compute_lookahead_band_energy(noisy) 760 */
    cpx L[FREQ];
    window_fft(L, tr->noisy->comb_buf + COMB_BUF - WINDOW);
    band_energy(Ey_look, L);
  }
  memcpy(out138, Ey_look, sizeof(Ey_look));                 /* 764-773 */
  memcpy(out138 + NB, Ephaty, NB * sizeof(float));
  out138[68] = (float)tr->noisy->last_period / (PITCH_MAX - 3 * PITCH_MIN);
  out138[69] = tr->noisy->pitch_corr;
  memcpy(out138 + 70, g, sizeof(g));
  memcpy(out138 + 70 + NB, r, sizeof(r));
}

/* the whole binary on in-memory PCM (both inputs at least count frames long: no rewind) */
void pno_train_run(const short *speech, const short *noisy, int count, float *out, short *test_out) {
  pno_train *tr = pno_train_create(); float x[FRAME], n[FRAME], to[FRAME]; int t, i;
  for (t = 0; t < count; t++) {
    for (i = 0; i < FRAME; i++) { x[i] = (float)speech[(size_t)t * FRAME + i]; n[i] = (float)noisy[(size_t)t * FRAME + i]; }
    pno_train_frame(tr, x, n, out + (size_t)t * 138, to);
    if (test_out) for (i = 0; i < FRAME; i++) test_out[(size_t)t * FRAME + i] = (short)fmax(-32768, fmin(32767, to[i]));
  }
  pno_train_destroy(tr);
}
