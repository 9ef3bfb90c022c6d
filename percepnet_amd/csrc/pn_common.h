// Internal shared definitions for libpercepnet_hip (host + device).
#pragma once
#ifndef PN_NO_HIP                 // PN_NO_HIP: the HIP-free host pieces (pn_model.cpp, pn_pack.cpp, pn_tables.cpp) built alone, e.g.
#include <hip/hip_runtime.h>      // by the sanitizer harness tests/c/host_sanitize.cpp with plain g++
#endif
#include <stdint.h>
#include <stddef.h>

#define PN_EXPORT __attribute__((visibility("default")))   // the library is built with -fvisibility=hidden
#define PN_FRAME 480
#define PN_WINDOW 960
#define PN_FREQ 481
#define PN_NB 34
#define PN_NFEAT 70
#define PN_FEAT_STRIDE 128      // features padded with zeros to a multiple of the GEMM K-tiles (32 fp32 / 64 fp16)
#define PN_AUX_STRIDE 72        // front-end side outputs for the training-feature path: Ep[34] | Exp[34] | pitch_corr
#define PN_NFFT 960
#define PN_HIST_FRAMES 12       // comb_buf = 5760 samples = 12 frames (denoise.cpp:32), kept as a ring
#define PN_HIST (PN_HIST_FRAMES * PN_FRAME)
#define PN_HIST_STRIDE (PN_HIST + 8)   // per-stream ring + a mirror of its first 8 samples: any 4 consecutive logical samples are
                                       // 4 consecutive floats in memory, also across the ring's wrap (unaligned dwordx4 comb-tap loads)
#define PN_SPEC_BINS 400        // bins >= 400 never contribute (denoise.cpp:89-182)
#define PN_PITCH_MAX 768
#define PN_PITCH_MIN 60
#define PN_COMB_M 3

// Read-only tables shared by all streams (CommonState + erb_band, denoise.cpp:61-69,87,186-214).
// Computed on the host in double exactly as the reference computes them, uploaded once.
struct PnTables {
  float tw[PN_NFFT * 2];          // twiddles (r,i) kiss_fft.cpp:406-421
  float half_window[PN_FRAME];    // denoise.cpp:191-192
  float comb_hann[8];             // 7 used, denoise.cpp:200-206
  float tansig[208];              // 201 used, tansig_table.h
  float pna, n0;                  // CommonState.power_noise_attenuation, .n0 (denoise.cpp:207-211)
  float bin_frac[PN_SPEC_BINS];   // (float)j / band_size for bin = border[i] + j
  int16_t bitrev[PN_NFFT];        // digit-reversal scatter index kiss_fft.cpp:315-345
  int16_t border[PN_NB + 2];      // ERBBand::nfftborder erbband.h:63-75
  uint8_t bin_band[PN_SPEC_BINS]; // band i with border[i] <= bin < border[i+1]
  // Band-major operand layout of the phase-split front end's band reductions (pn_dsp_fe_split_s.hip): band b's chain
  // reads, in order, frac*tmp of interval b-1 then (1-frac)*tmp of interval b; its operands are stored contiguously
  // from float band_start[b] (a multiple of 4), each part padded with zeros to a multiple of 4, band_nq[b] quads in all.
  uint16_t band_pos_a[PN_SPEC_BINS]; // where frac*tmp of bin k goes (band bin_band[k]+1, part 1)
  uint16_t band_pos_b[PN_SPEC_BINS]; // where (1-frac)*tmp of bin k goes (band bin_band[k], part 2)
  uint16_t band_start[PN_NB + 2];
  uint16_t band_nq[PN_NB + 2];
};
#define PN_BAND_LAYOUT_FLOATS 920   // total size of that layout for the 34-band table of erbband.h (checked in pn_build_tables)

int pn_build_tables(PnTables *t);          // 0, or -1 with pn_set_error (a build whose table layout and kernels disagree)

// Network geometry (rnn_train.py:105-121 / rnn.cpp:42-81)
enum { PN_L_FC, PN_L_CONV1, PN_L_CONV2, PN_L_GRU1, PN_L_GRU2, PN_L_GRU3, PN_L_GRU_GB, PN_L_GRU_RB,
       PN_L_FC_GB, PN_L_FC_RB, PN_NLAYERS };
enum { PN_KIND_DENSE = 0, PN_KIND_CONV1D = 1, PN_KIND_GRU = 2 };

struct PnLayerHost {
  int kind, nin, nn, ks, act, reset_after;
  const float *bias, *w, *rw;   // host pointers into the model's own copy (nnet_data.h layouts)
};

struct PnLayerSrc { int kind, nin, nn, ks, act, reset_after; const float *bias, *w, *rw; };   // one layer's arrays, wherever they live

struct pn_model {
  PnLayerHost L[PN_NLAYERS];
  float *storage;               // one malloc holding every array
  size_t n_floats;
  unsigned char sha256[32];     // SHA-256 of the arrays and the layer descriptors: key of the per-device cache of packed weights
};

#ifndef PN_NO_HIP
#define PN_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) { pn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return -1; } \
  } while (0)

#endif
void pn_set_error(const char *fmt, ...);

// pn_model.cpp: the fixed topology and the size of a layer's arrays in the nnet_data.h layout
struct PnGeom { int kind, nin, nn, ks; };
extern const PnGeom pn_kGeom[];
size_t pn_layer_floats(int kind, int nin, int nn, int ks, size_t *nb, size_t *nw, size_t *nr);
// pn_pack.cpp: host-side re-packing of the weight matrices for the fp32 MFMA kernels
size_t pn_packed_floats(int k_alloc, int ncols, int ct_round);
void pn_pack_weights(const float *W, int K, int k_alloc, int ncols, int ct_round, float *Wp);
int pn_ct_padded(int ncols, int ct_round);
int pn_dense_nt(int N);
size_t pn_packed_floats_n16(int K, int ncols);
void pn_pack_weights_n16(const float *W, int K, int ncols, float *Wq);

#ifndef PN_NO_HIP
// Every entry point runs on the context's device and leaves the caller's current device as it found it (callers
// hand in torch data_ptr()s from a thread whose current device torch manages).
struct DeviceGuard {
  int prev = -1; bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    if (prev == dev) prev = -1;                       // nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};
#define PN_ON_DEVICE(c) DeviceGuard _dg((c)->device); if (!_dg.ok) { pn_set_error("hipSetDevice(%d) failed", (c)->device); return -1; }

#endif
