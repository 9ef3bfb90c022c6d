// Training-feature path, SURVEY §8(f) row 1: what the `percepNet` binary's train() loop does after its
// two compute_frame_features calls (reference denoise.cpp:732-773).  The heavy part — the two analyses
// per pair — is the same pn_frontend_kernel the inference path uses (one launch over the speech
// streams, one over the noisy streams); this file holds the per-band target arithmetic and the TEST
// build's saturating short cast.
//
// Arithmetic follows the overloads the reference's C++ resolves to: sqrt(float) is the float sqrt,
// pow(float,int) and every expression holding a double literal are double.  Compiled with
// -ffp-contract=off like the rest of the DSP.  One deliberate tolerance: post_filtering's sinf
// (denoise.cpp:227) is libm's on the CPU and OCML's here; both are within 1 ULP of sin, not of each
// other, so the 34 ideal gains (and nothing else in the record) are compared with a ULP bound.
#include "pn_common.h"

__global__ __launch_bounds__(64) void pn_targets_kernel(
    const PnTables *__restrict__ T, int n_pairs,
    const float *__restrict__ ex_clean,       // [n][36]  Ex of the speech streams  (X = analysis of frame t-5)
    const float *__restrict__ ex_noisy,       // [n][36]  Ey
    const float *__restrict__ ey_look_noisy,  // [n][36]  Ey_lookahead (compute_lookahead_band_energy, 760)
    const float *__restrict__ aux_clean,      // [n][PN_AUX_STRIDE]  Ep | Exp | pitch_corr
    const float *__restrict__ aux_noisy,      //                     Ephat | Ephaty | pitch_corr
    const int *__restrict__ period_noisy,     // [n] noisy->last_period
    float *__restrict__ records,              // pair p's 138 floats at records + p*rec_stride
    long long rec_stride,
    float *__restrict__ gr) {                 // optional [n][68]: g (post-filtered) | r for the TEST synthesis
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const float pna = T->pna, n0 = T->n0;
  const float *Ex = ex_clean + (size_t)p * 36, *Ey = ex_noisy + (size_t)p * 36;
  const float *Exp = aux_clean + (size_t)p * PN_AUX_STRIDE + PN_NB;
  const float *Ephaty = aux_noisy + (size_t)p * PN_AUX_STRIDE + PN_NB;
  float *rec = records + (size_t)p * rec_stride;
  float g[PN_NB], gw[PN_NB], ey[PN_NB];
  float E0 = 0, E1 = 0;
#pragma unroll
  for (int i = 0; i < PN_NB; i++) {
    const float exp_ = Exp[i], ephaty = Ephaty[i];
    ey[i] = Ey[i];
    // calc_ideal_gain (571-577): the .0001 literal makes the division double
    float gi = (float)((double)Ex[i] / (.0001 + (double)ey[i]));
    if (gi > 1) gi = 1;
    if (gi < 0) gi = 0;
    // estimate_phat_corr (549-553): pow(float,2) -> double, exactly the square
    const double e2 = (double)ephaty * (double)ephaty;
    const float ephatp = (float)((double)ephaty / sqrt((double)(1 - pna) * e2 + (double)pna));
    // filter_strength_calc (555-569), called with Ephaty in the Eyp slot (736)
    float a = ephatp * ephatp - exp_ * exp_;
    if (a < 0) a = 0;
    const float b = ephatp * ephaty * (1 - exp_ * exp_);
    float c = exp_ * exp_ - ephaty * ephaty;
    if (c < 0) c = 0;
    const float alpha = (float)((double)(sqrtf(b * b + a * (c)) - b) / ((double)a + 1e-8));
    float ri = alpha / (1 + alpha);
    // adjust_gain_strength_by_condition (579-589)
    if (ephatp < exp_) {
      const float g_att = sqrtf((1 + n0 - exp_ * exp_) / (1 + n0 - ephatp * ephatp));
      ri = 0.99;
      gi *= g_att;
    }
    g[i] = gi;
    rec[70 + PN_NB + i] = ri;
    if (gr) gr[(size_t)p * 68 + PN_NB + i] = ri;
    rec[i] = ey_look_noisy[(size_t)p * 36 + i];          // 764
    rec[PN_NB + i] = ephaty;                             // 765
  }
  // post_filtering (216-250), applied before the record is written because the reference builds
  // with TEST defined (45-47, 743): three sequential float sums in band order
#pragma unroll
  for (int i = 0; i < PN_NB; i++) gw[i] = g[i] * sinf((float)(M_PI / 2 * (double)g[i]));
#pragma unroll
  for (int i = 0; i < PN_NB; i++) E0 += g[i] * ey[i];
#pragma unroll
  for (int i = 0; i < PN_NB; i++) E1 += gw[i] * ey[i];
  const float E_div = E0 / (E1 + 1e-6f);
  const float G = sqrtf(((1 + 0.02f) * E_div) / (1 + 0.02f * (E_div * E_div)));
#pragma unroll
  for (int i = 0; i < PN_NB; i++) {
    const float v = G * gw[i];
    rec[70 + i] = v;                                     // 771
    if (gr) gr[(size_t)p * 68 + i] = v;
  }
  rec[68] = (float)period_noisy[p] / (PN_PITCH_MAX - 3 * PN_PITCH_MIN);   // 767
  rec[69] = aux_noisy[(size_t)p * PN_AUX_STRIDE + 2 * PN_NB];             // 768
}

// (short)fmax(-32768, fmin(32767, out[i]*NORM_RATIO)), denoise.cpp:754-756 (NORM_RATIO 1)
__global__ __launch_bounds__(256) void pn_saturate_i16_kernel(int n_pairs, const float *__restrict__ in,
                                                              int16_t *__restrict__ out, long long out_stride) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n_pairs * PN_FRAME) return;
  const size_t p = idx / PN_FRAME; const int i = (int)(idx % PN_FRAME);
  const double d = fmax(-32768.0, fmin(32767.0, (double)in[idx]));
  out[p * out_stride + i] = (int16_t)(int)d;
}

void pn_launch_targets(hipStream_t st, const PnTables *T, int n_pairs, const float *ex_clean, const float *ex_noisy,
                       const float *ey_look_noisy, const float *aux_clean, const float *aux_noisy,
                       const int *period_noisy, float *records, long long rec_stride, float *gr) {
  hipLaunchKernelGGL(pn_targets_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, st, T, n_pairs, ex_clean, ex_noisy,
                     ey_look_noisy, aux_clean, aux_noisy, period_noisy, records, rec_stride, gr);
}

void pn_launch_saturate_i16(hipStream_t st, int n_pairs, const float *in, int16_t *out, long long out_stride) {
  const size_t n = (size_t)n_pairs * PN_FRAME;
  hipLaunchKernelGGL(pn_saturate_i16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_pairs, in, out,
                     out_stride);
}
