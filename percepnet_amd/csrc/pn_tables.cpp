// Host-side construction of the read-only tables the kernels stage into LDS.
// Each table is computed with the same expression, in the same precision, as the reference
// computes it at start-up, so the uploaded float32 bits are identical to the CPU path's:
//   twiddles      kiss_fft.cpp:406-421  (cos/sin of a double phase, narrowed to float)
//   digit reversal kiss_fft.cpp:315-345 for the factorisation 960 = 5*3*4*4*4 (kf_factor 352-404)
//   half_window   denoise.cpp:191-192   (Vorbis power-complementary window, double -> float)
//   comb window   denoise.cpp:200-206   (normalised Hann, float accumulation of the sum)
//   band borders  erbband.h:43-75       (ERB scale, float/double mix as written there)
//   tanh table    tansig_table.h        ((float)tanh(.04*i) printed "%f", read back as float)
#include "pn_common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static float pn_freq2erb(float freq_hz) { return 9.265 * log(1 + freq_hz / (24.7 * 9.265)); }
static float pn_erb2freq(float n_erb) { return 24.7 * 9.265 * (exp(n_erb / 9.265) - 1); }

int pn_build_tables(PnTables *t) {
  memset(t, 0, sizeof(*t));
  for (int i = 0; i < PN_NFFT; i++) {
    const double pi = 3.14159265358979323846264338327;
    double phase = (-2 * pi / PN_NFFT) * i;
    t->tw[2 * i] = (float)cos(phase);
    t->tw[2 * i + 1] = (float)sin(phase);
  }
  // input index n = n0 + 5*(n1 + 3*(n2 + 4*(n3 + 4*n4)))  ->  n0*192 + n1*64 + n2*16 + n3*4 + n4
  for (int i = 0; i < PN_NFFT; i++) {
    int n = i;
    int n0 = n % 5; n /= 5;
    int n1 = n % 3; n /= 3;
    int n2 = n % 4; n /= 4;
    int n3 = n % 4; n /= 4;
    t->bitrev[i] = (int16_t)(n0 * 192 + n1 * 64 + n2 * 16 + n3 * 4 + n);
  }
  for (int i = 0; i < PN_FRAME; i++)
    t->half_window[i] = sin(.5 * M_PI * sin(.5 * M_PI * (i + .5) / PN_FRAME) * sin(.5 * M_PI * (i + .5) / PN_FRAME));
  {
    float temp_sum = 0;
    for (int i = 1; i < PN_COMB_M * 2 + 2; i++) {
      t->comb_hann[i - 1] = 0.5 - 0.5 * cos(2.0 * M_PI * i / (PN_COMB_M * 2 + 2));
      temp_sum += t->comb_hann[i - 1];
    }
    for (int i = 1; i < PN_COMB_M * 2 + 2; i++) t->comb_hann[i - 1] /= temp_sum;
    t->pna = 0;                                             // denoise.cpp:207-210 (float accumulation)
    for (int i = 1; i < PN_COMB_M * 2 + 2; i++) t->pna += t->comb_hann[i - 1] * t->comb_hann[i - 1];
    t->n0 = 0.03;                                           // denoise.cpp:211
  }
  {
    const int N = PN_NB - 2;
    float erb_low = pn_freq2erb(0.f), erb_high = pn_freq2erb(20000.f);
    float lims[PN_NB], cut[PN_NB];
    float num = (float)(N + 2);
    float delta = (erb_high - erb_low) / (num - 1);
    for (int i = 0; i < N + 1; i++) lims[i] = erb_low + delta * i;
    lims[N + 1] = erb_high;
    for (int i = 0; i < N + 2; i++) cut[i] = pn_erb2freq(lims[i]);
    int border[PN_NB];
    for (int k = 0; k < N + 2; k++) border[k] = (int)((cut[k] + 25) / 50.f);
    for (int k = 0; k < N; k++)
      if (border[k + 1] - border[k] < 2) border[k + 1] += (2 - (border[k + 1] - border[k]));
    for (int k = 0; k < PN_NB; k++) t->border[k] = (int16_t)border[k];
    // per-bin (band, frac) lookup for the three band loops (denoise.cpp:97-104, 133-140, 170-173)
    for (int i = 0; i < PN_NB - 1; i++) {
      int band_size = border[i + 1] - border[i];
      for (int j = 0; j < band_size; j++) {
        int bin = border[i] + j;
        if (bin < PN_SPEC_BINS) { t->bin_band[bin] = (uint8_t)i; t->bin_frac[bin] = (float)j / band_size; }
      }
    }
    // band-major operand layout (see PnTables): band b = [interval b-1, frac part][interval b, (1-frac) part]
    int pos = 0;
    for (int b = 0; b < PN_NB; b++) {
      t->band_start[b] = (uint16_t)pos;
      const int n1 = b >= 1 ? border[b] - border[b - 1] : 0, n2 = b <= PN_NB - 2 ? border[b + 1] - border[b] : 0;
      for (int j = 0; j < n1; j++) t->band_pos_a[border[b - 1] + j] = (uint16_t)(pos + j);
      pos += (n1 + 3) / 4 * 4;
      for (int j = 0; j < n2; j++) t->band_pos_b[border[b] + j] = (uint16_t)(pos + j);
      pos += (n2 + 3) / 4 * 4;
      t->band_nq[b] = (uint16_t)((pos - t->band_start[b]) / 4);
    }
    if (pos != PN_BAND_LAYOUT_FLOATS || border[PN_NB - 1] != PN_SPEC_BINS) {
      pn_set_error("band layout is %d floats / last border %d, the kernels are built for %d / %d (pn_tables.cpp and the "
                   "DSP kernels disagree: rebuild)", pos, border[PN_NB - 1], PN_BAND_LAYOUT_FLOATS, PN_SPEC_BINS);
      return -1;
    }
  }
  for (int i = 0; i <= 200; i++) {
    char buf[32];
    float v = (float)tanh(0.04 * i);
    snprintf(buf, sizeof(buf), "%f", (double)v);
    t->tansig[i] = (float)strtod(buf, NULL);
  }
  return 0;
}
