// Host side of libpercepnet_hip: models, batched contexts, the per-frame launch sequence and the
// C-ABI declared in include/percepnet_hip.h.  Mirrors the reference's frame engine
// (rnnoise_create/init/process_frame, denoise.cpp:252-280,508-547) for B streams in lock-step.
#include "pn_common.h"
#include "../../include/percepnet_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <array>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "pn_launch.h"
#include "pn_selftest_golden.h"

extern "C" int pn_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// ---- contexts -----------------------------------------------------------------------------------------
enum { KF_FRONTEND, KF_FC, KF_CONV1, KF_CONV2, KF_GRU512, KF_GRU_RB, KF_FC_GB, KF_FC_RB, KF_BACKEND, KF_FE_SPEC_IN, KF_FE_PITCH,
       KF_FE_SPEC_OUT, KF_COUNT };
static const char *kKernelNames[KF_COUNT] = {"frontend", "fc", "conv1", "conv2", "gru512", "gru_rb", "fc_gb", "fc_rb", "backend",
                                            "fe_spec_in", "fe_pitch", "fe_spec_out"};
enum { FE_MONO_G4 = 0, FE_MONO_G2 = 1, FE_SPLIT = 2 };

struct DevLayer { float *bias, *w, *rw, *wp, *rwp, *wq; };   // wq: narrow layers of small-batch fp32 contexts (pn_pack_weights_n16)

// Device copy of a model's biases and (re-packed) weights, shared by every context of one (model content, device, network
// mode, narrow-layer packing): the reference binds all its states to ONE static model (denoise.cpp:49-51,267: a borrowed
// pointer, zero copies); here N contexts — N legacy rnnoise_create handles, the shards of a CLI run, a service that opens
// and closes contexts — share one 32 MB upload and one re-pack instead of N.  Reference-counted, freed with its last user.
struct SharedWeights {
  int refs = 0;
  DevLayer L[PN_NLAYERS];
  std::vector<void *> allocs;
  size_t bytes = 0;
};
// SHA-256 of the model content (pn_model_from_sources: arrays + activations + reset_after), array length, device, nn_mode, narrow
// layers packed for the n16 kernel.  The digest IS the identity: no host copy of the model is kept and nothing is compared byte for
// byte on a hit (round 5 kept 32 MB per entry and memcmp'ed it under the build lock).
typedef std::tuple<std::array<unsigned char, 32>, size_t, int, int, int> WeightsKey;
static std::mutex g_weights_mu;                              // guards g_weights and every refs counter
static std::mutex g_weights_build_mu[16];                    // per device (mod 16): uploads and re-packs of DIFFERENT devices run
                                                             // side by side (percepnet_run --devices creates its contexts from one thread per device)
static std::map<WeightsKey, SharedWeights *> g_weights;

struct pn_ctx {
  int device, B, nn_mode;
  int x3_rg;                       // split-precision mode: row groups of 32 per wave (1: 128-row blocks, 2: 256-row blocks), fixed at creation from B
  int small, small_gru;            // network kernel family per layer kind: 1 = small-batch (pn_nn_small.hip), fixed at creation from B
  int n48;                         // fp32 MFMA mode, batches above the n16 limit: fc_gb on the batch form of the 16x16x4 kernel (pn_nn_n48.hip)
  int direct;                      // fp32 MFMA mode, large batches: 1 = the direct-operand family (pn_nn_d.hip: A fragments from fp32 shadows,
                                   // 64 rows per wave when x3_rg == 2); fixed at creation from B, never together with small / small_gru
  int fe_mode;                     // front end: FE_SPLIT = three phase kernels (pn_dsp_fe_split_*.hip); FE_MONO_G4 / FE_MONO_G2 = the
                                   // single-launch kernel with four / two streams per wavefront (pn_dsp_fe.hip, pn_dsp_fe_g2.hip)
  size_t Bp;                       // B rounded up to the largest GEMM M tile (256): row count of every network buffer
  hipStream_t stream; bool own_stream;
  hipStream_t chain_stream[4] = {nullptr, nullptr, nullptr, nullptr};      // launch_rnn: streams of the row-range chains 1..3 (chain 0 = stream)
  hipEvent_t chain_fork = nullptr, chain_join[4] = {nullptr, nullptr, nullptr, nullptr};
  char chain_kind[5] = {'-', '-', '-', '-', 0};   // how each chain stream was obtained (n: default priority, probed; h: priority stream)
  int64_t t;                       // frames done: indexes the DSP rings (hist slot t%12, yring/eyring t%6)
  int64_t tn;                      // network steps done: indexes the conv rings (tn%5, tn%3) and the GRU ping-pong (tn&1).
                                   // == t unless pn_ctx_compute_rnn_host advanced the network on its own (rnn.cpp:42 is
                                   // callable on an RNNState without a DenoiseState in the reference too)
  size_t bytes;
  PnLayerHost geom[PN_NLAYERS];
  DevLayer L[PN_NLAYERS];           // = weights->L (pointers into the shared copy)
  SharedWeights *weights = NULL; WeightsKey weights_key; bool weights_were_cached = false;
  // pn_ctx_reset_streams: stream ids on the device, and a ring of pinned host copies (the H2D copy runs when the stream gets
  // to it — frames may be in flight — so its source must outlive the call; slot k is reused once its copy has executed)
  int *d_ids = NULL; int ids_cap = 0;
  struct IdSlot { int *h = NULL; hipEvent_t ev = nullptr; } id_slot[4];
  unsigned id_calls = 0;
  // pn_process_*_active: inactive-row list on the device (+ its ring of pinned host copies) and the save area of the
  // in-place state of those rows, grown on demand
  struct Active {
    int *d_ids = NULL; int cap = 0;
    pn_ctx::IdSlot slot[4]; unsigned calls = 0;
    float *save_synth = NULL, *save_gr = NULL, *save_gain = NULL; uint32_t *save_out = NULL; int *save_period = NULL;
    std::vector<uint8_t> mark; std::vector<int32_t> inactive;
  } act;
  PnTables *tables; float *tansig;
  float *hist, *synth, *last_gain, *feat, *c1ring, *c2ring, *c2out, *gru[4], *rb, *gr, *io_in, *io_out;
  // fp16-operand variant only: shadow copies (2 bytes per element, same indexing) of the buffers the GEMMs read
  uint16_t *c1ringH, *c2ringH, *c2outH, *gruH[4], *rbH;
  float2 *yring, *Ps;              // yring: [6][B][400] look-ahead spectra (X of frame t = slot (t+1)%6)
  float *eyring;                   // [6][B][36] look-ahead band energies
  bool postfilter = false;         // optional envelope post-filter in the back end (pn_ctx_set_postfilter)
  bool x3_sat = false;             // PERCEPNET_X3_SATCOUNT=1 (shadow-operand modes): count operand values clamped to the fp16 range
  int dsp_grid_cap = 0;            // > 0 only in the DSP self-test's temporary context: its DSP launches use that many blocks
  bool inject_bad_launch = false;  // pn_ctx_debug_inject_launch_failure (tests): the next frames hand fc a geometry its launcher refuses
  int *last_period, *silence;
  std::vector<void *> allocs;
  bool profiling;
  struct Ev { int fam; hipEvent_t a, b; };
  std::vector<Ev> events;
  std::vector<hipEvent_t> event_pool;   // recycled timing events: no hipEventCreate/Destroy inside a timed region
  double fam_ms[KF_COUNT]; int64_t fam_n[KF_COUNT];
  // pipelined host-buffer path (pn_submit_host_*): created on first use
  struct Pipe {
    bool init = false;
    hipStream_t h2d = nullptr, d2h = nullptr;
    hipEvent_t in_ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, delivered[2] = {nullptr, nullptr};
    void *in[2] = {nullptr, nullptr}, *out[2] = {nullptr, nullptr};
    float *gr[2] = {nullptr, nullptr};
    int64_t submitted = 0;
    char kind[3] = {'?', '?', 0};             // how each copy stream was obtained: n (default priority, probed) / h / l (priority stream)
  } pipe;
};

static thread_local bool g_last_alloc_oom = false;     // the last dev_alloc failure on this thread was hipErrorOutOfMemory
static int dev_alloc_into(std::vector<void *> &allocs, size_t &total, hipStream_t stream, void **p, size_t bytes, bool zero) {
  // PERCEPNET_GUARD=1 (debugging aid): every buffer is followed by 1 MB of 0xFF (NaN as fp32 and as fp16), so that a
  // read past the end of a buffer shows up as NaN in the outputs instead of as run-to-run noise
  static const bool guard = getenv("PERCEPNET_GUARD") != NULL;
  const size_t pad = guard ? (1u << 20) : 0, body = (bytes + 255) & ~(size_t)255;
  {
    const hipError_t e = hipMalloc(p, guard ? body + pad : bytes);
    if (e != hipSuccess) {
      g_last_alloc_oom = (e == hipErrorOutOfMemory);
      (void)hipGetLastError();
      pn_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return -1;
    }
  }
  allocs.push_back(*p);
  total += bytes;
  if (zero) PN_HIP_CHECK(hipMemsetAsync(*p, 0, bytes, stream));
  if (guard) PN_HIP_CHECK(hipMemsetAsync((char *)*p + body, 0xFF, pad, stream));
  return 0;
}
static int dev_alloc(pn_ctx *c, void **p, size_t bytes, bool zero) { return dev_alloc_into(c->allocs, c->bytes, c->stream, p, bytes, zero); }
#define DEV_ALLOC(ptr, count, zero) \
  do { if (dev_alloc(c, (void **)&(ptr), sizeof(*(ptr)) * (size_t)(count), zero)) goto fail; } while (0)

static int upload(pn_ctx *c, float **dst, const float *src, size_t n) {
  if (dev_alloc(c, (void **)dst, n * sizeof(float), false)) return -1;
  PN_HIP_CHECK(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return 0;
}
// the same into the shared weight copy under construction (uploads run on the creating context's stream)
static int upload_w(pn_ctx *c, SharedWeights *w, float **dst, const float *src, size_t n) {
  if (dev_alloc_into(w->allocs, w->bytes, c->stream, (void **)dst, n * sizeof(float), false)) return -1;
  PN_HIP_CHECK(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return 0;
}

// operand shadows: 1 half per element (fp16-operand mode) or a hi and a lo plane (split-precision mode)
// (the fp32 shadows of the direct-operand family are 4 bytes per element, laid out in the same 16-byte slab entries)
static size_t shadow_halfs_per_element(const pn_ctx *c) { return (c->nn_mode == PN_NN_MFMA_X3 || c->direct) ? 2 : 1; }
static bool x3_layer(int li) { return li == PN_L_CONV1 || li == PN_L_CONV2 || li == PN_L_GRU_RB || li == PN_L_FC_GB || (li >= PN_L_GRU1 && li < PN_L_GRU1 + 4); }
static int zero_state(pn_ctx *c) {
  const size_t B = c->B;
  PN_HIP_CHECK(hipMemsetAsync(c->hist, 0, B * PN_HIST_STRIDE * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->synth, 0, B * PN_FRAME * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->yring, 0, 6 * B * PN_SPEC_BINS * sizeof(float2), c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->eyring, 0, 6 * B * 36 * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->last_gain, 0, B * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->last_period, 0, B * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->silence, 0, B * 4, c->stream));
  const size_t Bp = c->Bp;
  PN_HIP_CHECK(hipMemsetAsync(c->feat, 0, Bp * PN_FEAT_STRIDE * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->c1ring, 0, 5 * Bp * 128 * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->c2ring, 0, 3 * Bp * 512 * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->c2out, 0, Bp * 512 * 4, c->stream));
  for (int i = 0; i < 4; i++) PN_HIP_CHECK(hipMemsetAsync(c->gru[i], 0, 2 * Bp * 512 * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->rb, 0, 2 * Bp * 128 * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(c->gr, 0, B * 68 * 4, c->stream));
  if (c->c2outH) {
    const size_t hb = 2 * shadow_halfs_per_element(c);   // shadow bytes per element
    if (c->c1ringH) PN_HIP_CHECK(hipMemsetAsync(c->c1ringH, 0, 5 * Bp * 128 * hb, c->stream));
    if (c->c2ringH) PN_HIP_CHECK(hipMemsetAsync(c->c2ringH, 0, 3 * Bp * 512 * hb, c->stream));
    PN_HIP_CHECK(hipMemsetAsync(c->c2outH, 0, Bp * 512 * hb, c->stream));
    for (int i = 0; i < 4; i++) PN_HIP_CHECK(hipMemsetAsync(c->gruH[i], 0, 2 * Bp * 512 * hb, c->stream));
    PN_HIP_CHECK(hipMemsetAsync(c->rbH, 0, 2 * Bp * 128 * hb, c->stream));
  }
  c->t = 0; c->tn = 0;
  return 0;
}

// Front-end kernel family for a batch size.  PERCEPNET_FE=split|mono|g2 overrides (PERCEPNET_FE_G2=1 is the older
// spelling of g2).
static int pn_fe_mode_for(int n_streams) {
  if (const char *e = getenv("PERCEPNET_FE")) {
    if (!strcmp(e, "split")) return FE_SPLIT;
    if (!strcmp(e, "mono") || !strcmp(e, "g4")) return FE_MONO_G4;
    if (!strcmp(e, "g2")) return FE_MONO_G2;
  }
  if (const char *e = getenv("PERCEPNET_FE_G2")) return atoi(e) ? FE_MONO_G2 : FE_MONO_G4;
  (void)n_streams;
  return FE_SPLIT;      // measured: 0.102 vs 0.124 ms (g2) at 1024 streams, 0.130 vs 0.166 (g4) at 4096, 1.32 vs 2.37 at 65536 (profiles/r03e_*)
}
// narrow (34-column) dense layers on 16x16x4 MFMA tiles (pn_dense_n16_kernel) up to this batch size (measured: fc_gb 0.026 vs
// 0.056 ms at 1024 streams, 0.082 vs 0.101 at 16384, 0.327 vs 0.180 at 65536 where the batch GEMM's operand reuse wins);
// PERCEPNET_N16_ROWS overrides
static bool n48_enabled() {             // PERCEPNET_N48=0: fc_gb of large fp32 contexts back on the 32-column batch GEMM
  const char *e = getenv("PERCEPNET_N48");
  return !e || atoi(e) != 0;
}
static bool n16_rows_ok(int n_streams) {
  const char *e = getenv("PERCEPNET_N16_ROWS");
  return n_streams <= (e ? atoi(e) : 20480);
}
static int nn_selftest(pn_ctx *c);
static int dsp_selftest(pn_ctx *c);
static int nn_chains_of(const pn_ctx *c);
static int chain_streams_init(pn_ctx *c, int n);
static pn_ctx *ctx_create(const pn_model *model, int device, int n_streams, int nn_mode, void *hip_stream, bool selftest,
                          int force_small, int force_small_gru, int force_x3_rg = 0, int force_n16 = -1, int force_direct = -1);

extern "C" void pn_ctx_destroy(pn_ctx *c) {
  if (!c) return;
  DeviceGuard _dg(c->device);
  hipStreamSynchronize(c->stream);
  if (c->pipe.init) {
    hipStreamSynchronize(c->pipe.h2d); hipStreamSynchronize(c->pipe.d2h);
    for (int k = 0; k < 2; k++) { hipEventDestroy(c->pipe.in_ready[k]); hipEventDestroy(c->pipe.done[k]); hipEventDestroy(c->pipe.delivered[k]); }
    hipStreamDestroy(c->pipe.h2d); hipStreamDestroy(c->pipe.d2h);
  }
  for (int k = 1; k < 4; k++) if (c->chain_stream[k]) { hipStreamSynchronize(c->chain_stream[k]); hipStreamDestroy(c->chain_stream[k]); if (c->chain_join[k]) hipEventDestroy(c->chain_join[k]); }
  if (c->chain_fork) hipEventDestroy(c->chain_fork);
  for (auto &e : c->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
  for (void *p : c->allocs) hipFree(p);
  for (auto &sl : c->id_slot) { if (sl.h) hipHostFree(sl.h); if (sl.ev) hipEventDestroy(sl.ev); }
  for (auto &sl : c->act.slot) { if (sl.h) hipHostFree(sl.h); if (sl.ev) hipEventDestroy(sl.ev); }
  if (c->weights) {
    std::lock_guard<std::mutex> lk(g_weights_mu);
    if (--c->weights->refs == 0) {
      for (void *p : c->weights->allocs) hipFree(p);
      g_weights.erase(c->weights_key);
      delete c->weights;
    }
  }
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" pn_ctx *pn_ctx_create(const pn_model *model, int device, int n_streams, int nn_mode, void *hip_stream) {
  return ctx_create(model, device, n_streams, nn_mode, hip_stream, true, -1, -1);
}

// Biases + weights of `model` on the context's device in the layout `nn_mode` reads: STRICT the nnet_data.h arrays as they
// are, the MFMA modes re-packed tile orders (pn_pack.cpp, pn_nn_x3.hip).  Returns NULL (pn_set_error) on failure.
static SharedWeights *build_weights(pn_ctx *c, const pn_model *model, int nn_mode, bool n16, bool n48) {
  SharedWeights *w = new SharedWeights();
  memset(w->L, 0, sizeof(w->L));
  for (int li = 0; li < PN_NLAYERS; li++) {
    const PnLayerHost &H = model->L[li];
    size_t nb, nw, nr;
    pn_layer_floats(H.kind, H.nin, H.nn, H.ks, &nb, &nw, &nr);
    if (upload_w(c, w, &w->L[li].bias, H.bias, nb)) goto fail_w;
    if (nn_mode == PN_NN_STRICT) {
      if (upload_w(c, w, &w->L[li].w, H.w, nw)) goto fail_w;
      if (nr && upload_w(c, w, &w->L[li].rw, H.rw, nr)) goto fail_w;
    } else {
      const int K = H.nin * H.ks, ncols = H.nn * (H.kind == PN_KIND_GRU ? 3 : 1);
      const int k_alloc = (li == PN_L_FC) ? PN_FEAT_STRIDE : K;   // fc sweeps the zero-padded feature panel
      const int ctr = H.kind == PN_KIND_GRU ? 1 : pn_dense_nt(H.nn);
      if ((nn_mode == PN_NN_MFMA_X3 || nn_mode == PN_NN_MFMA_F16) && x3_layer(li)) {   // conv1, conv2, the GRUs and fc_gb; fc and fc_rb (K = 70 / 128) stay fp32 below
        const int np = nn_mode == PN_NN_MFMA_X3 ? 2 : 1;         // operand planes: hi + lo (split precision) or hi only (fp16 operands)
        const int ctx3 = H.kind == PN_KIND_GRU ? 1 : pn_dense_x3_nt(H.nn);
        std::vector<uint16_t> packed(pn_packed_halfs_x3(K, ncols, ctx3, np));
        if (pn_pack_weights_x3(H.w, K, K, ncols, ctx3, np, packed.data())) { pn_set_error("layer %d has a weight outside the fp16 range: the fp16-operand and split-precision modes cannot represent it", li); goto fail_w; }
        if (upload_w(c, w, &w->L[li].wp, (const float *)packed.data(), packed.size() / 2)) goto fail_w;
        if (hipStreamSynchronize(c->stream) != hipSuccess) goto fail_w;   // `packed` dies at scope end
        if (nr) {
          std::vector<uint16_t> rp(pn_packed_halfs_x3(H.nn, ncols, 1, np));
          if (pn_pack_weights_x3(H.rw, H.nn, H.nn, ncols, 1, np, rp.data())) { pn_set_error("layer %d has a recurrent weight outside the fp16 range", li); goto fail_w; }
          if (upload_w(c, w, &w->L[li].rwp, (const float *)rp.data(), rp.size() / 2)) goto fail_w;
          if (hipStreamSynchronize(c->stream) != hipSuccess) goto fail_w;
        }
      } else {
        std::vector<float> packed(pn_packed_floats(k_alloc, ncols, ctr));
        pn_pack_weights(H.w, K, k_alloc, ncols, ctr, packed.data());
        if (upload_w(c, w, &w->L[li].wp, packed.data(), packed.size())) goto fail_w;
        if (hipStreamSynchronize(c->stream) != hipSuccess) goto fail_w;   // `packed` dies at scope end
        if ((n16 || (n48 && li == PN_L_FC_GB)) && H.kind == PN_KIND_DENSE && ncols <= 48 && K % 128 == 0) {     // fc_gb, fc_rb (n48: fc_gb only)
          std::vector<float> pq(pn_packed_floats_n16(K, ncols));
          pn_pack_weights_n16(H.w, K, ncols, pq.data());
          if (upload_w(c, w, &w->L[li].wq, pq.data(), pq.size())) goto fail_w;
          if (hipStreamSynchronize(c->stream) != hipSuccess) goto fail_w;
        }
        if (nr) {
          std::vector<float> rp(pn_packed_floats(H.nn, ncols, 1));
          pn_pack_weights(H.rw, H.nn, H.nn, ncols, 1, rp.data());
          if (upload_w(c, w, &w->L[li].rwp, rp.data(), rp.size())) goto fail_w;
          if (hipStreamSynchronize(c->stream) != hipSuccess) goto fail_w;
        }
      }
    }
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) { pn_set_error("weight upload failed"); goto fail_w; }
  return w;
fail_w:
  hipStreamSynchronize(c->stream);
  for (void *p : w->allocs) hipFree(p);
  delete w;
  return NULL;
}

// force_small / force_small_gru: -1 = choose the network kernel family from the batch size (the public behaviour);
// 0 / 1 = the self-test's temporary contexts run the SAME family as the context under test whatever their own size.
static pn_ctx *ctx_create(const pn_model *model, int device, int n_streams, int nn_mode, void *hip_stream, bool selftest,
                          int force_small, int force_small_gru, int force_x3_rg, int force_n16, int force_direct) {
  if (!model) { pn_set_error("NULL model"); return NULL; }
  if (n_streams < 1) { pn_set_error("n_streams must be >= 1"); return NULL; }
  if (nn_mode != PN_NN_MFMA && nn_mode != PN_NN_STRICT && nn_mode != PN_NN_MFMA_F16 && nn_mode != PN_NN_MFMA_X3) { pn_set_error("bad nn_mode %d", nn_mode); return NULL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    pn_set_error("no HIP device available (this library has no CPU fallback)");
    return NULL;
  }
  if (device < 0 || device >= ndev) { pn_set_error("device %d out of range (%d devices)", device, ndev); return NULL; }
  DeviceGuard _dg(device);
  if (!_dg.ok) { pn_set_error("hipSetDevice(%d) failed", device); return NULL; }
  pn_ctx *c = new pn_ctx();
  c->device = device; c->B = n_streams; c->Bp = ((size_t)n_streams + 255) / 256 * 256; c->nn_mode = nn_mode; c->small = force_small >= 0 ? force_small : n_streams <= pn_small_rows(); c->small_gru = force_small_gru >= 0 ? force_small_gru : n_streams <= pn_small_gru_rows(); c->fe_mode = pn_fe_mode_for(n_streams); c->x3_rg = force_x3_rg ? force_x3_rg : pn_x3_rg_for(n_streams);
  c->direct = (nn_mode == PN_NN_MFMA && !c->small && !c->small_gru) ? (force_direct >= 0 ? force_direct : pn_direct_for(n_streams)) : 0;
  if (c->direct) c->x3_rg = force_x3_rg ? (force_x3_rg >= 2 ? 2 : 1) : pn_direct_rg_for(n_streams);
  c->t = 0; c->tn = 0; c->bytes = 0; c->profiling = false;
  memset(c->fam_ms, 0, sizeof(c->fam_ms)); memset(c->fam_n, 0, sizeof(c->fam_n));
  memset(c->L, 0, sizeof(c->L));
  c->c1ringH = c->c2ringH = c->c2outH = c->rbH = NULL; memset(c->gruH, 0, sizeof(c->gruH));

  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { pn_set_error("hipStreamCreate failed"); delete c; return NULL; }
    c->own_stream = true;
  }
  const size_t B = n_streams, Bp = c->Bp;
  {
    PnTables *ht = new PnTables();
    int rc = pn_build_tables(ht);
    if (!rc) rc = dev_alloc(c, (void **)&c->tables, sizeof(PnTables), false);
    if (!rc && hipMemcpyAsync(c->tables, ht, sizeof(PnTables), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = -1;
    if (!rc) rc = upload(c, &c->tansig, ht->tansig, 208);
    hipStreamSynchronize(c->stream);
    delete ht;
    if (rc) goto fail;
  }
  DEV_ALLOC(c->hist, B * PN_HIST_STRIDE, false);
  DEV_ALLOC(c->synth, B * PN_FRAME, false);
  DEV_ALLOC(c->last_gain, B, false);
  DEV_ALLOC(c->last_period, B, false);
  DEV_ALLOC(c->silence, B, false);
  DEV_ALLOC(c->yring, 6 * B * PN_SPEC_BINS, false);
  DEV_ALLOC(c->eyring, 6 * B * 36, false);
  DEV_ALLOC(c->Ps, B * PN_SPEC_BINS, true);
  DEV_ALLOC(c->feat, Bp * PN_FEAT_STRIDE, false);
  DEV_ALLOC(c->c1ring, 5 * Bp * 128, false);
  DEV_ALLOC(c->c2ring, 3 * Bp * 512, false);
  DEV_ALLOC(c->c2out, Bp * 512, false);
  for (int i = 0; i < 4; i++) DEV_ALLOC(c->gru[i], 2 * Bp * 512, false);
  DEV_ALLOC(c->rb, 2 * Bp * 128, false);
  DEV_ALLOC(c->gr, B * 68, false);
  if (nn_mode == PN_NN_MFMA_F16 || nn_mode == PN_NN_MFMA_X3 || c->direct) {
    const size_t hp = shadow_halfs_per_element(c);      // 1: fp16 shadow; 2: hi + lo planes (split precision) / fp32 fragments (direct-operand GRUs)
    if (!c->direct) {                                   // (the direct-operand family keeps its dense layers on the batch kernels: no shadows of the conv FIFOs)
      DEV_ALLOC(c->c1ringH, hp * 5 * Bp * 128, false);
      DEV_ALLOC(c->c2ringH, hp * 3 * Bp * 512, false);
    }
    DEV_ALLOC(c->c2outH, hp * Bp * 512, false);
    for (int i = 0; i < 4; i++) DEV_ALLOC(c->gruH[i], hp * 2 * Bp * 512, false);
    DEV_ALLOC(c->rbH, hp * 2 * Bp * 128, false);
  }
  DEV_ALLOC(c->io_in, B * PN_FRAME, false);
  DEV_ALLOC(c->io_out, B * PN_FRAME, false);
  if (zero_state(c)) goto fail;
  for (int li = 0; li < PN_NLAYERS; li++) { c->geom[li] = model->L[li]; c->geom[li].bias = c->geom[li].w = c->geom[li].rw = NULL; }
  {   // the device copy of the weights: shared with every other context of this model content on this device in this mode
    // narrow layers on the 16x16x4 kernel at small batches in every MFMA mode (in the shadow-operand modes that is fc_rb; fc_gb runs on their own kernels)
    // force_n16 (the self-tests' temporary contexts): 0 batch GEMM, 1 n16, 2 n48
    const bool n16 = (nn_mode != PN_NN_STRICT) && (force_n16 >= 0 ? force_n16 == 1 : n16_rows_ok(n_streams));
    const bool n48 = nn_mode == PN_NN_MFMA && !c->small && !n16 && (force_n16 >= 0 ? force_n16 == 2 : n48_enabled());
    c->n48 = n48;
    std::array<unsigned char, 32> dig;
    memcpy(dig.data(), model->sha256, 32);
    c->weights_key = std::make_tuple(dig, model->n_floats, device, nn_mode, n16 ? 1 : (n48 ? 2 : 0));
    std::lock_guard<std::mutex> build_lk(g_weights_build_mu[device & 15]);   // one build per device at a time; the map lock is never held across a build
    SharedWeights *hit = NULL;
    {
      std::lock_guard<std::mutex> lk(g_weights_mu);
      auto it = g_weights.find(c->weights_key);
      if (it != g_weights.end()) { hit = it->second; hit->refs++; }
    }
    if (hit) { c->weights = hit; c->weights_were_cached = true; }
    else {
      SharedWeights *w = build_weights(c, model, nn_mode, n16, n48);
      if (!w) goto fail;
      w->refs = 1;
      c->weights = w;
      std::lock_guard<std::mutex> lk(g_weights_mu);
      g_weights[c->weights_key] = w;
    }
    memcpy(c->L, c->weights->L, sizeof(c->L));
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) { pn_set_error("initial upload failed"); goto fail; }
  if (selftest && (nn_mode == PN_NN_MFMA_X3 || nn_mode == PN_NN_MFMA_F16)) {       // (not for the self-tests' temporary contexts)
    const char *e = getenv("PERCEPNET_X3_SATCOUNT");
    c->x3_sat = e && atoi(e);
  }
  if (selftest && nn_mode != PN_NN_STRICT && nn_selftest(c)) goto fail;
  if (selftest && dsp_selftest(c)) goto fail;
  if (c->x3_sat && pn_x3_sat_set(1)) { pn_set_error("cannot enable the operand-saturation counter"); goto fail; }   // after the self-tests: starts at zero
  if (selftest && nn_chains_of(c) > 1 && chain_streams_init(c, nn_chains_of(c))) goto fail;      // probed now, not inside the first frame
  return c;
fail:
  pn_ctx_destroy(c);
  return NULL;
}

static int pipe_drain(pn_ctx *c);
extern "C" int pn_ctx_reset(pn_ctx *c) { if (!c) return -1; PN_ON_DEVICE(c); if (pipe_drain(c)) return -1; return zero_state(c); }
// rnnoise_init for a subset of the streams (denoise.cpp:259-280): every row of stream s in every ring slot / ping-pong half
// of every state buffer goes to zero (pn_state.hip says why that is a fresh stream whatever the ring phases are)
extern "C" int pn_ctx_reset_streams(pn_ctx *c, const int32_t *ids, int n) {
  if (!c || n < 0 || (n > 0 && !ids)) { pn_set_error("bad argument"); return -1; }
  if (n == 0) return 0;
  for (int i = 0; i < n; i++) if (ids[i] < 0 || ids[i] >= c->B) { pn_set_error("stream id %d out of range [0, %d)", ids[i], c->B); return -1; }
  PN_ON_DEVICE(c);
  if (c->ids_cap < n) {
    // (the old, smaller buffers stay in allocs until destroy: kernels of an earlier call may still be reading them)
    const int cap = n < 1024 ? 1024 : n;
    int *p = NULL;
    PN_HIP_CHECK(hipMalloc((void **)&p, (size_t)cap * sizeof(int)));
    c->allocs.push_back(p); c->d_ids = p;
    for (auto &sl : c->id_slot) {
      if (sl.ev) PN_HIP_CHECK(hipEventSynchronize(sl.ev));
      if (sl.h) hipHostFree(sl.h);
      sl.h = NULL;
      PN_HIP_CHECK(hipHostMalloc((void **)&sl.h, (size_t)cap * sizeof(int), hipHostMallocDefault));
      if (!sl.ev) PN_HIP_CHECK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    c->ids_cap = cap;
  }
  {
    pn_ctx::IdSlot &sl = c->id_slot[c->id_calls++ & 3];
    PN_HIP_CHECK(hipEventSynchronize(sl.ev));             // the copy issued from this slot four calls ago has executed
    memcpy(sl.h, ids, (size_t)n * sizeof(int));
    PN_HIP_CHECK(hipMemcpyAsync(c->d_ids, sl.h, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    PN_HIP_CHECK(hipEventRecord(sl.ev, c->stream));
  }
  hipStream_t st = c->stream; const int *d = c->d_ids;
  const long long B = c->B, Bp = (long long)c->Bp;
  pn_launch_zero_rows(st, c->hist, PN_HIST_STRIDE, PN_HIST_STRIDE, 1, 0, d, n);
  pn_launch_zero_rows(st, c->synth, PN_FRAME, PN_FRAME, 1, 0, d, n);
  pn_launch_zero_rows(st, c->yring, 2 * PN_SPEC_BINS, 2 * PN_SPEC_BINS, 6, B * 2 * PN_SPEC_BINS, d, n);
  pn_launch_zero_rows(st, c->eyring, 36, 36, 6, B * 36, d, n);
  pn_launch_zero_rows(st, c->Ps, 2 * PN_SPEC_BINS, 2 * PN_SPEC_BINS, 1, 0, d, n);
  pn_launch_zero_rows(st, c->last_gain, 1, 1, 1, 0, d, n);
  pn_launch_zero_rows(st, c->last_period, 1, 1, 1, 0, d, n);
  pn_launch_zero_rows(st, c->silence, 1, 1, 1, 0, d, n);
  pn_launch_zero_rows(st, c->feat, PN_FEAT_STRIDE, PN_FEAT_STRIDE, 1, 0, d, n);
  pn_launch_zero_rows(st, c->c1ring, 128, 128, 5, Bp * 128, d, n);
  pn_launch_zero_rows(st, c->c2ring, 512, 512, 3, Bp * 512, d, n);
  pn_launch_zero_rows(st, c->c2out, 512, 512, 1, 0, d, n);
  for (int i = 0; i < 4; i++) pn_launch_zero_rows(st, c->gru[i], 512, 512, 2, Bp * 512, d, n);
  pn_launch_zero_rows(st, c->rb, 128, 128, 2, Bp * 128, d, n);
  pn_launch_zero_rows(st, c->gr, 68, 68, 1, 0, d, n);
  if (c->c2outH) {                                         // operand shadows (fp16-operand / split-precision modes, direct-operand GRUs); a NULL one is skipped
    const int np = (int)shadow_halfs_per_element(c);
    pn_launch_zero_shadow_rows(st, c->c1ringH, 128, np, 5, np * Bp * 128, d, n);
    pn_launch_zero_shadow_rows(st, c->c2ringH, 512, np, 3, np * Bp * 512, d, n);
    pn_launch_zero_shadow_rows(st, c->c2outH, 512, np, 1, 0, d, n);
    for (int i = 0; i < 4; i++) pn_launch_zero_shadow_rows(st, c->gruH[i], 512, np, 2, np * Bp * 512, d, n);
    pn_launch_zero_shadow_rows(st, c->rbH, 128, np, 2, np * Bp * 128, d, n);
  }
  PN_HIP_CHECK(hipGetLastError());
  return 0;
}
extern "C" int pn_ctx_n_streams(const pn_ctx *c) { return c ? c->B : -1; }
extern "C" int64_t pn_ctx_frames_done(const pn_ctx *c) { return c ? c->t : -1; }
// state (+ tables) of this context, plus the weights if this context created their device copy (a context that found them
// in the cache adds nothing: the copy is shared)
extern "C" size_t pn_ctx_device_bytes(const pn_ctx *c) { return c ? c->bytes + ((c->weights && !c->weights_were_cached) ? c->weights->bytes : 0) : 0; }
// bytes of the packed weight copy this context reads, whoever created it: a process's footprint is the sum of
// pn_ctx_device_bytes over its contexts plus every DISTINCT shared copy that no live context reports as its own
extern "C" size_t pn_ctx_weight_bytes(const pn_ctx *c) { return (c && c->weights) ? c->weights->bytes : 0; }
extern "C" int pn_ctx_describe(const pn_ctx *c, char *buf, size_t n) {
  if (!c || !buf || !n) return -1;
  const char *nn = c->nn_mode == PN_NN_STRICT ? "strict" : (c->nn_mode == PN_NN_MFMA_F16 ? "mfma_f16" : (c->nn_mode == PN_NN_MFMA_X3 ? "mfma_x3" : "mfma_f32"));
  const bool x3 = c->nn_mode == PN_NN_MFMA_X3 || c->nn_mode == PN_NN_MFMA_F16;      // shadow-operand kernels (pn_nn_x3.hip)
  const bool fam = c->nn_mode == PN_NN_MFMA || x3;      // the small-batch family exists for the fp32 MFMA kernels only (in the shadow-operand modes: fc, fc_rb)
  // rows per wave (conv1, conv2, GRUs, fc_gb); x3_rg 3 = 64 rows with the GRUs on the paired-phase kernel (pn_gru_x3p_kernel)
  const char *xk = c->nn_mode == PN_NN_MFMA_X3 ? (c->x3_rg >= 2 ? "x3_rows64" : "x3_rows32") : (c->x3_rg >= 2 ? "f16_rows64" : "f16_rows32");
  const char *xg = c->nn_mode == PN_NN_MFMA_X3 ? (c->x3_rg == 3 ? "x3_rows64_paired" : xk) : (c->x3_rg == 3 ? "f16_rows64_paired" : xk);
  const char *dk = c->x3_rg >= 2 ? "direct_rows64" : "direct_rows32";      // direct-operand fp32 GRU kernels (pn_nn_d.hip); the dense layers stay "batch"
  const int w = snprintf(buf, n, "nn=%s dense=%s gru=%s gru_rb=%s narrow=%s frontend=%s weights=%s nn_chains=%d%s%s", nn, x3 ? xk : (fam && c->small ? "small" : "batch"),
                         x3 ? xg : (c->direct ? dk : (fam && c->small_gru ? "small" : "batch")), x3 ? xg : (c->direct ? dk : (fam && c->small ? "small" : "batch")),
                         x3 ? (c->L[PN_L_FC_RB].wq ? "fc_gb:x3+fc_rb:n16" : "fc_gb:x3+fc_rb:fp32") : (c->n48 ? "fc_gb:n48+fc_rb:batch" : (c->L[PN_L_FC_GB].wq ? "n16" : (fam && c->small ? "small" : "batch"))), c->fe_mode == FE_SPLIT ? "split" : (c->fe_mode == FE_MONO_G2 ? "g2" : "g4"),
                         c->weights_were_cached ? "shared" : "own", nn_chains_of(c), nn_chains_of(c) > 1 ? ":" : "", nn_chains_of(c) > 1 ? c->chain_kind + 1 : "");
  if (w < 0 || (size_t)w >= n) return -1;
  if (c->x3_sat) {                                        // debug: operand values clamped to +-65504 so far (device-wide counter)
    DeviceGuard _dg(c->device);
    hipStreamSynchronize(c->stream);
    const int w2 = snprintf(buf + w, n - w, " x3_saturated=%lld", pn_x3_sat_read());
    return (w2 < 0 || (size_t)(w + w2) >= n) ? -1 : w + w2;
  }
  return w;
}
extern "C" int pn_ctx_synchronize(pn_ctx *c) { if (!c) return -1; PN_ON_DEVICE(c); PN_HIP_CHECK(hipStreamSynchronize(c->stream)); return 0; }

// ---- profiling ------------------------------------------------------------------------------------------
struct Scope {
  pn_ctx *c; int fam; hipEvent_t a, b; bool on; hipStream_t st;      // st: the stream the bracketed launches go to (a row-range chain's own)
  Scope(pn_ctx *c_, int fam_, hipStream_t st_ = nullptr) : c(c_), fam(fam_), on(c_->profiling), st(st_ ? st_ : c_->stream) {
    if (on) {
      auto take = [&](hipEvent_t &e) { if (c->event_pool.empty()) hipEventCreate(&e); else { e = c->event_pool.back(); c->event_pool.pop_back(); } };
      take(a); take(b); hipEventRecord(a, st);
    }
  }
  ~Scope() {
    if (on) { hipEventRecord(b, st); c->events.push_back({fam, a, b}); }
    // debugging aid: PERCEPNET_SYNC_EACH=<bit mask over kernel families, -1 = all>: host sync after those launches
    static const long sync_mask = getenv("PERCEPNET_SYNC_EACH") ? strtol(getenv("PERCEPNET_SYNC_EACH"), NULL, 0) : 0;
    if (sync_mask & (1L << fam)) hipStreamSynchronize(st);
  }
};

static int flush_events(pn_ctx *c) {
  PN_HIP_CHECK(hipStreamSynchronize(c->stream));
  for (int k = 1; k < 4; k++) if (c->chain_stream[k]) PN_HIP_CHECK(hipStreamSynchronize(c->chain_stream[k]));
  for (auto &e : c->events) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { c->fam_ms[e.fam] += ms; c->fam_n[e.fam]++; }
    c->event_pool.push_back(e.a); c->event_pool.push_back(e.b);
  }
  c->events.clear();
  return 0;
}

extern "C" int pn_ctx_set_profiling(pn_ctx *c, int enable) {
  if (!c) return -1;
  if (enable && c->event_pool.size() < 2048) {        // enough for ~75 frames between two reads; created outside any timed region
    PN_ON_DEVICE(c);
    while (c->event_pool.size() < 2048) { hipEvent_t e; PN_HIP_CHECK(hipEventCreate(&e)); c->event_pool.push_back(e); }
  }
  c->profiling = enable != 0;
  return 0;
}
extern "C" int pn_kernel_count(void) { return KF_COUNT; }
extern "C" const char *pn_kernel_name(int i) { return (i >= 0 && i < KF_COUNT) ? kKernelNames[i] : NULL; }
extern "C" int pn_ctx_kernel_time(pn_ctx *c, const char *name, double *total_ms, int64_t *launches) {
  if (!c || !name) return -1;
  if (flush_events(c)) return -1;
  for (int i = 0; i < KF_COUNT; i++)
    if (!strcmp(name, kKernelNames[i])) { if (total_ms) *total_ms = c->fam_ms[i]; if (launches) *launches = c->fam_n[i]; return 0; }
  pn_set_error("unknown kernel family '%s'", name);
  return -1;
}
extern "C" int pn_ctx_reset_profile(pn_ctx *c) {
  if (!c) return -1;
  if (flush_events(c)) return -1;
  memset(c->fam_ms, 0, sizeof(c->fam_ms)); memset(c->fam_n, 0, sizeof(c->fam_n));
  return 0;
}

// ---- the per-frame launch sequence -----------------------------------------------------------------------
static PnSegs seg1(const float *p, int ld, int width) { PnSegs s; memset(&s, 0, sizeof(s)); s.p[0] = p; s.ld[0] = ld; s.width[0] = width; s.n = 1; return s; }

// compute_rnn (rnn.cpp:42-81) for all streams; features in c->feat, result in c->gr
// fp16 shadow of an fp32 activation pointer (same element index in the twin buffer)
static uint16_t *shadow(pn_ctx *c, const float *p) {
  const size_t Bp = c->Bp;
  struct { const float *f; uint16_t *h; size_t n; } m[8] = {
      {c->c1ring, c->c1ringH, 5 * Bp * 128}, {c->c2ring, c->c2ringH, 3 * Bp * 512}, {c->c2out, c->c2outH, Bp * 512},
      {c->gru[0], c->gruH[0], 2 * Bp * 512}, {c->gru[1], c->gruH[1], 2 * Bp * 512}, {c->gru[2], c->gruH[2], 2 * Bp * 512},
      {c->gru[3], c->gruH[3], 2 * Bp * 512}, {c->rb, c->rbH, 2 * Bp * 128}};
  for (auto &e : m) if (p >= e.f && p < e.f + e.n) return e.h ? e.h + shadow_halfs_per_element(c) * (size_t)(p - e.f) : NULL;
  return NULL;
}
static PnSegs shadow_segs(pn_ctx *c, const PnSegs &A) {
  PnSegs H = A;
  for (int j = 0; j < A.n; j++) H.p[j] = reinterpret_cast<const float *>(shadow(c, A.p[j]));
  return H;
}

// Returns 0, or -1 when a launcher refused its geometry (pn_set_error names it): the refused layer is not launched (later
// layers of the frame may be — their results are never reported) and the caller fails the frame.
// The ten layers for the rows [r0, r0 + nrows) of the batch on stream `st`.  Every activation buffer is row-major, so a row range
// is the same launch with every base pointer moved down by r0 rows.  small / small_gru: the kernel family.  The shadow-operand
// and STRICT modes always run the whole batch.
static int launch_rnn_rows(pn_ctx *c, size_t r0, size_t nrows, hipStream_t st, int small, int small_gru) {
  const size_t Bp = c->Bp; const int strict = c->nn_mode == PN_NN_STRICT; const int64_t t = c->tn;
  const int B = (int)nrows;
  // x3: the layers that run on the fp16 matrix cores from operand shadows — split precision (hi + lo planes) or fp16 operands (hi only)
  const bool x3 = c->nn_mode == PN_NN_MFMA_X3 || c->nn_mode == PN_NN_MFMA_F16;
  const int np = c->nn_mode == PN_NN_MFMA_X3 ? 2 : 1;
  const bool dm = c->direct != 0;   // the GRU steps take their activations straight from fragment-order fp32 shadows (pn_nn_d.hip): written by conv2
                                    // (batch kernel, second output) and by the GRU steps themselves; every dense layer stays on the batch kernels
  const float *tab = c->tansig;
  int rc = 0;
  const int cur = (int)(t & 1), nxt = cur ^ 1;
  // every chain's launches are bracketed on the stream they go to (pn_ctx_kernel_times averages over all launches of a family;
  // with N chains the launches of one family overlap in time: bench.py prices the CONCURRENT launches together)
  struct MaybeScope { Scope s; MaybeScope(pn_ctx *c_, int fam, hipStream_t st_) : s(c_, fam, st_) {} };
  float *c1new = c->c1ring + (size_t)(t % 5) * Bp * 128 + r0 * 128;
  float *c2new = c->c2ring + (size_t)(t % 3) * Bp * 512 + r0 * 512;
  float *c2out = c->c2out + r0 * 512, *gr = c->gr + r0 * 68;
  { MaybeScope sc(c, KF_FC, st);
    PnSegs A = seg1(c->feat + r0 * PN_FEAT_STRIDE, PN_FEAT_STRIDE, strict ? PN_NFEAT : PN_FEAT_STRIDE);   // cols 70..127 are zero
    if (c->inject_bad_launch && !strict) A.width[0] = 96;   // test hook: three K-tiles, which every MFMA dense launcher refuses
    rc |= pn_launch_dense(st, strict, A, c->L[PN_L_FC].w, c->L[PN_L_FC].wp, c->L[PN_L_FC].bias, 128, c->geom[PN_L_FC].act, tab, c1new, 128, B, small);
    if (x3) rc |= pn_launch_split_x3(st, c1new, 128, 128, shadow(c, c1new), (int)Bp, np); }   // fc runs in fp32 (70 inputs); its output enters the shadow-operand layers
  { MaybeScope sc(c, KF_CONV1, st);   // causal conv as dense over [4 previous fc outputs | current] (nnet.cpp:182-200)
    PnSegs A; memset(&A, 0, sizeof(A)); A.n = 5;
    for (int j = 0; j < 5; j++) { A.p[j] = c->c1ring + (size_t)((t + 1 + j) % 5) * Bp * 128 + r0 * 128; A.ld[j] = 128; A.width[j] = 128; }
    if (x3) rc |= pn_launch_dense_x3(st, shadow_segs(c, A), c->L[PN_L_CONV1].wp, c->L[PN_L_CONV1].bias, 512, c->geom[PN_L_CONV1].act, tab, c2new, 512, shadow(c, c2new), 16, B, c->x3_rg, np);
    else rc |= pn_launch_dense(st, strict, A, c->L[PN_L_CONV1].w, c->L[PN_L_CONV1].wp, c->L[PN_L_CONV1].bias, 512, c->geom[PN_L_CONV1].act, tab, c2new, 512, B, small); }
  { MaybeScope sc(c, KF_CONV2, st);
    PnSegs A; memset(&A, 0, sizeof(A)); A.n = 3;
    for (int j = 0; j < 3; j++) { A.p[j] = c->c2ring + (size_t)((t + 1 + j) % 3) * Bp * 512 + r0 * 512; A.ld[j] = 512; A.width[j] = 512; }
    if (x3) rc |= pn_launch_dense_x3(st, shadow_segs(c, A), c->L[PN_L_CONV2].wp, c->L[PN_L_CONV2].bias, 512, c->geom[PN_L_CONV2].act, tab, c2out, 512, c->c2outH, 16, B, c->x3_rg, np);
    else if (dm) rc |= pn_launch_dense(st, 0, A, NULL, c->L[PN_L_CONV2].wp, c->L[PN_L_CONV2].bias, 512, c->geom[PN_L_CONV2].act, tab, c2out, 512, B, 0, shadow(c, c2out), 16);   // + the shadow the GRUs read
    else rc |= pn_launch_dense(st, strict, A, c->L[PN_L_CONV2].w, c->L[PN_L_CONV2].wp, c->L[PN_L_CONV2].bias, 512, c->geom[PN_L_CONV2].act, tab, c2out, 512, B, small); }
  const float *x = c2out;
  for (int i = 0; i < 4 && !rc; i++) {    // gru1 -> gru2 -> gru3 -> gru_gb, each fed the UPDATED state of its predecessor
    MaybeScope sc(c, KF_GRU512, st);
    const int li = PN_L_GRU1 + i;
    float *ho = c->gru[i] + (size_t)cur * Bp * 512 + r0 * 512, *hn = c->gru[i] + (size_t)nxt * Bp * 512 + r0 * 512;
    PnSegs X = seg1(x, 512, 512);
    if (x3) rc |= pn_launch_gru_x3(st, shadow_segs(c, X), ho, shadow(c, ho), c->L[li].wp, c->L[li].rwp, c->L[li].bias, 512, c->geom[li].act, tab, hn, shadow(c, hn), B, c->x3_rg, np);
    else if (dm) rc |= pn_launch_gru_d(st, shadow_segs(c, X), ho, shadow(c, ho), c->L[li].wp, c->L[li].rwp, c->L[li].bias, 512, c->geom[li].act, tab, hn, shadow(c, hn), B, c->x3_rg);
    else rc |= pn_launch_gru(st, strict, X, ho, c->L[li].w, c->L[li].rw, c->L[li].wp, c->L[li].rwp, c->L[li].bias, 512, c->geom[li].act, tab, hn, B, small_gru);
    x = hn;
  }
  const float *g1 = c->gru[0] + (size_t)nxt * Bp * 512 + r0 * 512, *g2 = c->gru[1] + (size_t)nxt * Bp * 512 + r0 * 512,
              *g3 = c->gru[2] + (size_t)nxt * Bp * 512 + r0 * 512, *gb = c->gru[3] + (size_t)nxt * Bp * 512 + r0 * 512;
  float *rbo = c->rb + (size_t)cur * Bp * 128 + r0 * 128, *rbn = c->rb + (size_t)nxt * Bp * 128 + r0 * 128;
  { MaybeScope sc(c, KF_GRU_RB, st);   // input = [gru3 | conv2 out] (rnn.cpp:67-69)
    PnSegs X; memset(&X, 0, sizeof(X)); X.n = 2;
    X.p[0] = g3; X.ld[0] = 512; X.width[0] = 512; X.p[1] = c2out; X.ld[1] = 512; X.width[1] = 512;
    const int li = PN_L_GRU_RB;
    if (x3) rc |= pn_launch_gru_x3(st, shadow_segs(c, X), rbo, shadow(c, rbo), c->L[li].wp, c->L[li].rwp, c->L[li].bias, 128, c->geom[li].act, tab, rbn, shadow(c, rbn), B, c->x3_rg, np);
    else if (dm) rc |= pn_launch_gru_d(st, shadow_segs(c, X), rbo, shadow(c, rbo), c->L[li].wp, c->L[li].rwp, c->L[li].bias, 128, c->geom[li].act, tab, rbn, shadow(c, rbn), B, c->x3_rg);
    else rc |= pn_launch_gru(st, strict, X, rbo, c->L[li].w, c->L[li].rw, c->L[li].wp, c->L[li].rwp, c->L[li].bias, 128, c->geom[li].act, tab, rbn, B, small); }   // gru_rb (1024->128) crosses over with the dense layers
  { MaybeScope sc(c, KF_FC_GB, st);    // input = [conv2 out | gru1 | gru2 | gru3 | gru_gb] (rnn.cpp:72-77)
    PnSegs A; memset(&A, 0, sizeof(A)); A.n = 5;
    const float *ps[5] = {c2out, g1, g2, g3, gb};
    for (int j = 0; j < 5; j++) { A.p[j] = ps[j]; A.ld[j] = 512; A.width[j] = 512; }
    if (x3) rc |= pn_launch_dense_x3(st, shadow_segs(c, A), c->L[PN_L_FC_GB].wp, c->L[PN_L_FC_GB].bias, 34, c->geom[PN_L_FC_GB].act, tab, gr, 68, NULL, 0, B, c->x3_rg, np);
    else if (c->n48) rc |= pn_launch_dense_n48(st, A, c->L[PN_L_FC_GB].wq, c->L[PN_L_FC_GB].bias, 34, c->geom[PN_L_FC_GB].act, tab, gr, 68, B);
    else if (c->L[PN_L_FC_GB].wq) rc |= pn_launch_dense_n16(st, A, c->L[PN_L_FC_GB].wq, c->L[PN_L_FC_GB].bias, 34, c->geom[PN_L_FC_GB].act, tab, gr, 68, B);
    else rc |= pn_launch_dense(st, strict, A, c->L[PN_L_FC_GB].w, c->L[PN_L_FC_GB].wp, c->L[PN_L_FC_GB].bias, 34, c->geom[PN_L_FC_GB].act, tab, gr, 68, B, small); }
  { MaybeScope sc(c, KF_FC_RB, st);
    PnSegs A = seg1(rbn, 128, 128);
    if (c->L[PN_L_FC_RB].wq) rc |= pn_launch_dense_n16(st, A, c->L[PN_L_FC_RB].wq, c->L[PN_L_FC_RB].bias, 34, c->geom[PN_L_FC_RB].act, tab, gr + 34, 68, B);
    else rc |= pn_launch_dense(st, strict, A, c->L[PN_L_FC_RB].w, c->L[PN_L_FC_RB].wp, c->L[PN_L_FC_RB].bias, 34, c->geom[PN_L_FC_RB].act, tab, gr + 34, 68, B, small); }
  return rc ? -1 : 0;
}

// compute_rnn for the whole batch.  Large fp32 MFMA contexts run it as ROW-RANGE CHAINS (round 6, round-5 verdict item 3).
// The batch-GEMM kernels run in rounds of 512 co-resident blocks (two per CU) — 4096 rows of a 512-wide layer, 65 536 rows of a
// 34-wide one — and on ONE in-order stream every layer pays whole rounds: 65 536 streams = 16 rounds per 512-wide layer, 66 048 = 17
// (+0.45 ms per frame; the reference has no such step, its cost is per stream: nnet.cpp:120-180), and even an exact fit leaves the
// ramp and drain of ten launches idle.  No layer mixes rows, so the batch is cut into PN_NN_CHAINS row ranges (multiples of 128
// rows) whose ten layers are independent chains of the SAME kernels on streams of their own: while one chain drains a layer the
// blocks of another fill the slots.  One fork (the features are ready) and one join (before the back end) per frame; results
// are bit-identical by construction (the same launches over sub-ranges of the rows).  Measured (profiles/r06_row_chains.log).
// The first attempts — the rows past the last whole round on the small-batch kernels, in the same stream or beside the body —
// cost 2-3x the tail's share: a small block holds a block slot for a single latency-bound MFMA chain, and any slot taken from
// an exactly fitting body pushes that layer into an extra round.
#define PN_MAX_CHAINS 4
static int nn_chains_of(const pn_ctx *c) {
  if (c->nn_mode != PN_NN_MFMA || c->small || c->small_gru) return 1;
  const char *e = getenv("PN_NN_CHAINS");
  // default: two chains unless the batch fits EVERY layer's rounds exactly — a multiple of 32 768 streams (8 rounds of the 512-wide
  // layers, 2 of the 128-wide GRU) whose 128-row tile count also fits the 34-wide layers' single column block (<= 512 tiles or a
  // multiple of 512): 32 768, 65 536, 131 072, ...  Measured (profiles/r06_row_chains.log): two chains win 0.6-2 % at 20 480, 49 152,
  // 61 440, 69 632 and every size off the 4096-stream grid, and change nothing at 32 768 / 65 536 (+-0.03 ms) — where a second compute
  // stream would only be one more hardware queue for the pipelined host path's copy streams to stay clear of (HIP has four by default)
  const size_t mt = ((size_t)c->B + 127) / 128;
  // (second session of round 6: with the direct-operand GRU kernels — 256-row blocks, 8 instead of 16 rounds per 512-wide layer at
  // 65 536 streams, longer drains — two chains win at the exact fits from 65 536 streams too: 9.02 / 9.03 / 9.07 -> 8.93 / 8.99 / 8.96 ms
  // per frame at 65 536, 17.99 -> 17.85 at 131 072, even at 32 768; profiles/r06_direct_operand_gru.log H)
  const bool exact = (c->B % 32768 == 0) && (mt <= 512 || mt % 512 == 0) && !(c->direct && c->B >= 65536);
  int n = e ? atoi(e) : ((c->B > 16384 && !exact) ? 2 : 1);
  if (n < 1) n = 1;
  if (n > PN_MAX_CHAINS) n = PN_MAX_CHAINS;
  while (n > 1 && (size_t)c->B < (size_t)n * 4096) n--;      // a chain of fewer than 4096 rows is the small-batch regime: not worth a stream
  return n;
}
// Every extra chain's stream must have a HARDWARE queue of its own: HIP multiplexes the streams of one priority over a few queues,
// and on a queue shared with the context's stream a chain runs in front of the others instead of beside them.  Same remedy as for
// the copy streams of the pipelined host path: default-priority streams PROBED against the streams they must not share a queue
// with, a high-priority one as the fallback (pipe_make_stream).
static int pipe_make_stream(pn_ctx *c, hipStream_t *out, char how, int prio, char fallback, const std::vector<hipStream_t> &others, char *kind);
// every stream of this context that carries kernels or copies of a frame: a new one must share a hardware queue with none of them
static std::vector<hipStream_t> busy_streams(const pn_ctx *c) {
  std::vector<hipStream_t> v{c->stream};
  for (int k = 1; k < 4; k++) if (c->chain_stream[k]) v.push_back(c->chain_stream[k]);
  if (c->pipe.h2d) v.push_back(c->pipe.h2d);
  if (c->pipe.d2h) v.push_back(c->pipe.d2h);
  return v;
}
static int chain_streams_init(pn_ctx *c, int n) {
  int lo = 0, hi = 0;
  PN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  for (int k = 1; k < n; k++) {
    if (c->chain_stream[k]) continue;
    if (pipe_make_stream(c, &c->chain_stream[k], 'a', hi, 'h', busy_streams(c), &c->chain_kind[k])) return -1;
    PN_HIP_CHECK(hipEventCreateWithFlags(&c->chain_join[k], hipEventDisableTiming));
  }
  if (!c->chain_fork) PN_HIP_CHECK(hipEventCreateWithFlags(&c->chain_fork, hipEventDisableTiming));
  return 0;
}
static int launch_rnn(pn_ctx *c) {
  const int n = nn_chains_of(c);
  if (n <= 1) return launch_rnn_rows(c, 0, c->B, c->stream, c->small, c->small_gru);
  if (chain_streams_init(c, n)) return -1;
  // row ranges: equal shares rounded up to whole 128-row tiles; the last chain takes what is left
  const size_t tile = c->direct ? 128 * (size_t)c->x3_rg : 128;            // rows per block of the family's kernels
  const size_t B = c->B, share = ((B + n - 1) / n + tile - 1) / tile * tile;
  PN_HIP_CHECK(hipEventRecord(c->chain_fork, c->stream));                 // the front end's features (and last frame's state) are in place
  int rc = 0;
  for (int k = n - 1; k >= 0; k--) {                                      // the context's own stream last: the others are already queued
    const size_t r0 = share * k, nr = r0 >= B ? 0 : (B - r0 < share ? B - r0 : share);
    if (!nr) continue;
    hipStream_t st = k ? c->chain_stream[k] : c->stream;
    if (k) PN_HIP_CHECK(hipStreamWaitEvent(st, c->chain_fork, 0));
    rc |= launch_rnn_rows(c, r0, nr, st, c->small, c->small_gru);
    if (k) PN_HIP_CHECK(hipEventRecord(c->chain_join[k], st));             // also after a refused launch: nothing stays unordered
  }
  for (int k = 1; k < n; k++) PN_HIP_CHECK(hipStreamWaitEvent(c->stream, c->chain_join[k], 0));
  return rc ? -1 : 0;
}

// Known-answer self-test of the MFMA network kernels (PERCEPNET_SELFTEST=0 skips it).
// The MFMA paths (fp32 and fp16 operands) depend on the compiler's wait-state insertion and on pinned instruction
// order; a toolchain that schedules them differently could lose accumulator updates silently (the
// failure once seen hit output rows 27/31 mod 32 only).  So the first context of every (device, nn_mode, kernel
// family) in a process triggers one check of THE KERNELS — not of the caller's model: a fixed built-in synthetic weight
// set (uniform +-3/sqrt(fan_in), LCG-generated: gates from saturated to linear; the expected MFMA-vs-reference-order
// difference over two steps from the zero state is known and small; PERCEPNET_SELFTEST=2 prints it) is
// run for two network steps over 192 rows (six 32-row wave tiles, two M tiles) through two temporary contexts — the
// kernel family under test and the reference-order STRICT kernels — and the context is refused if any g/r output
// differs by more than 2e-5 (fp32 operands) / 4e-3 (fp16 operands, whose rounding the x3 weights amplify).  The verdict is cached for
// the process; a self-test that cannot allocate its ~70 MB of temporaries is reported as SKIPPED, not as a failure.
static std::mutex g_selftest_mu;
static std::map<std::tuple<int, int, int, int, int, int>, int> g_selftest_done;     // key -> 0 passed, 1 skipped

pn_model *pn_model_from_sources(const struct PnLayerSrc *src);
static pn_model *selftest_model() {
  static std::vector<float> store;
  PnLayerSrc s[PN_NLAYERS];
  size_t total = 0, nb, nw, nr;
  for (int li = 0; li < PN_NLAYERS; li++) total += pn_layer_floats(pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks, &nb, &nw, &nr);
  store.resize(total);
  unsigned x = 2463534242u;
  size_t off = 0;
  static const int act[PN_NLAYERS] = {3, 3, 2, 2, 2, 2, 2, 2, 1, 1};        // relu relu tanh tanh*5 sigmoid sigmoid (rnn_train.py:105-121)
  for (int li = 0; li < PN_NLAYERS; li++) {
    pn_layer_floats(pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks, &nb, &nw, &nr);
    const float bound_w = 1.f / sqrtf((float)(pn_kGeom[li].kind == PN_KIND_GRU ? pn_kGeom[li].nn : pn_kGeom[li].nin * pn_kGeom[li].ks));
    for (size_t i = 0; i < nb + nw + nr; i++) {
      x = x * 1664525u + 1013904223u;
      // x3: a good share of the GRU gates and tanh outputs saturate, so the clamped end of the activation table
      // (indices 192..200: a 192-thread block once failed to stage them) is exercised, not only its linear middle
      store[off + i] = ((int)(x >> 8) % 20001 - 10000) * 1e-4f * bound_w * (i < nb ? 1.f : 3.f);
    }
    s[li] = {pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks, act[li], 1, &store[off], &store[off + nb], nr ? &store[off + nb + nw] : NULL};
    off += nb + nw + nr;
  }
  pn_model *m = pn_model_from_sources(s);
  store.clear(); store.shrink_to_fit();
  return m;
}

static int nn_selftest(pn_ctx *c) {
  const char *env = getenv("PERCEPNET_SELFTEST");
  if (env && !atoi(env)) return 0;
  const int n16 = c->n48 ? 2 : (c->L[PN_L_FC_RB].wq != NULL);   // narrow layers: 1 = the 16x16x4 kernel (small batches; fc_rb in every MFMA mode, fc_gb in the fp32 one), 2 = fc_gb on its batch form, 0 = the batch GEMM
  const auto key = std::make_tuple(c->device, c->nn_mode, c->small, c->small_gru, (c->nn_mode == PN_NN_MFMA_X3 || c->nn_mode == PN_NN_MFMA_F16) ? c->x3_rg : (c->direct ? 10 + c->x3_rg : 0), n16);
  std::lock_guard<std::mutex> lk(g_selftest_mu);
  if (g_selftest_done.count(key)) return 0;
  const int rows = 192;
  const float tol = c->nn_mode == PN_NN_MFMA_F16 ? 4e-3f : 2e-5f;    // measured on the built-in set: 8.3e-7 (fp32), 1.03e-3 (fp16 operands); a lost k-step is O(0.1)
  pn_model *m = selftest_model();
  pn_ctx *cx[2] = {NULL, NULL};
  std::vector<float> feat((size_t)rows * PN_NFEAT), gr[2][2];
  int rc = m ? 0 : -1;
  bool oom = false;
  for (int pass = 0; pass < 2 && !rc; pass++) {          // pass 0: the kernel family under test; pass 1: STRICT kernels
    g_last_alloc_oom = false;
    cx[pass] = ctx_create(m, c->device, rows, pass ? PN_NN_STRICT : c->nn_mode, NULL, false, c->small, c->small_gru, c->x3_rg, n16, pass ? 0 : c->direct);
    if (!cx[pass]) { rc = -1; oom = g_last_alloc_oom; break; }
    unsigned x = 12345u;
    for (int step = 0; step < 2 && !rc; step++) {
      for (float &v : feat) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) % 2001 - 1000) * 1.5e-3f; }
      gr[pass][step].resize((size_t)rows * 68);
      if (pn_ctx_compute_rnn_host(cx[pass], feat.data(), gr[pass][step].data())) rc = -1;
    }
  }
  pn_ctx_destroy(cx[0]); pn_ctx_destroy(cx[1]); pn_model_free(m);
  if (rc && oom) {
    fprintf(stderr, "percepnet_hip: network self-test SKIPPED on device %d (not enough free memory for its temporaries): %s\n", c->device, pn_last_error());
    g_selftest_done[key] = 1;
    return 0;
  }
  if (rc) { std::string why = pn_last_error(); pn_set_error("network self-test could not run: %s", why.c_str()); return -1; }
  float worst = 0; int wrow = 0, wcol = 0;
  for (int step = 0; step < 2; step++)
    for (size_t i = 0; i < gr[0][step].size(); i++) {
      const float d = fabsf(gr[0][step][i] - gr[1][step][i]);
      if (!(d <= worst)) { worst = d; wrow = (int)(i / 68); wcol = (int)(i % 68); }     // NaN lands here too
    }
  if (env && atoi(env) >= 2)
    fprintf(stderr, "percepnet_hip: network self-test device %d nn_mode %d dense=%s gru=%s: worst |delta g,r| %g (tolerance %g) at row %d output %d\n",
            c->device, c->nn_mode, c->small ? "small" : "batch", c->small_gru ? "small" : "batch", (double)worst, (double)tol, wrow, wcol);
  if (!(worst <= tol)) {
    pn_set_error("network self-test FAILED (nn_mode %d, dense=%s gru=%s): the MFMA kernels differ from the reference-order kernels by %g "
                 "(> %g) at row %d (row %% 32 = %d), output %d on the built-in weight set — the build's instruction schedule is "
                 "not the validated one (DESIGN.md 4.3); refusing to run", c->nn_mode, c->small ? "small" : "batch",
                 c->small_gru ? "small" : "batch", (double)worst, (double)tol, wrow, wrow % 32, wcol);
    return -1;
  }
  g_selftest_done[key] = 0;
  return 0;
}


// Known-answer self-test of the DSP kernels, the counterpart of nn_selftest (PERCEPNET_SELFTEST=0 skips both).
// The first context of every (device, front-end family) in a process runs a fixed integer-generated waveform
// (two triangle waves + LCG noise, quiet and clipping stretches) through a temporary 40-stream context whose DSP
// launches are capped at ONE block (pn_ctx::dsp_grid_cap, an argument of the DSP launchers): every stream is fed the same PCM, so the 40 streams of 3 to 10
// grid-stride rounds must agree with each other word for word, the silence flags of all 14 frames (a full wrap of
// the 12-frame history ring) and the 70 features of the last frame must equal the CPU oracle's bit patterns stored in
// pn_selftest_golden.h (tools/make_dsp_selftest_golden.py; the features never touch the network).
static void selftest_pcm(std::vector<int16_t> &out) {     // in step with tools/make_dsp_selftest_golden.py
  const int n = PN_SELFTEST_FRAMES * PN_FRAME;
  out.resize(n);
  uint32_t x = 2463534242u;
  for (int i = 0; i < n; i++) {
    const int p1 = (i * 7) % 960, t1 = p1 < 480 ? p1 - 480 : 1440 - p1 - 480;
    const int p2 = (i * 31) % 960, t2 = p2 < 480 ? p2 - 480 : 1440 - p2 - 480;
    x = x * 1664525u + 1013904223u;
    const int noise = (int)((x >> 16) % 2001u) - 1000;
    const int amp = (i / 2400) % 2 == 1 ? 200 : 24;
    int v = amp * t1 + (amp / 3) * t2 + noise;
    v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
    out[i] = (int16_t)v;
  }
}

static int dsp_selftest(pn_ctx *c) {
  const char *env = getenv("PERCEPNET_SELFTEST");
  if (env && !atoi(env)) return 0;
  static std::map<std::pair<int, int>, int> done;
  const auto key = std::make_pair(c->device, c->fe_mode);
  std::lock_guard<std::mutex> lk(g_selftest_mu);
  if (done.count(key)) return 0;
  const int Bt = 40;
  std::vector<int16_t> pcm;
  selftest_pcm(pcm);
  pn_model *m = selftest_model();
  g_last_alloc_oom = false;
  pn_ctx *t = m ? ctx_create(m, c->device, Bt, PN_NN_MFMA, NULL, false, -1, -1) : NULL;
  if (!t) {
    const bool oom = g_last_alloc_oom;
    pn_model_free(m);
    if (oom) { fprintf(stderr, "percepnet_hip: DSP self-test SKIPPED on device %d (no memory for its temporaries)\n", c->device); done[key] = 1; return 0; }
    std::string why = pn_last_error(); pn_set_error("DSP self-test could not run: %s", why.c_str()); return -1;
  }
  std::vector<int16_t> in((size_t)Bt * PN_FRAME), out((size_t)Bt * PN_FRAME);
  std::vector<float> feat((size_t)Bt * PN_NFEAT);
  std::vector<int32_t> sil(Bt);
  int rc = 0; std::string msg;
  t->dsp_grid_cap = 1;
  for (int f = 0; f < PN_SELFTEST_FRAMES && !rc; f++) {
    for (int s = 0; s < Bt; s++) memcpy(&in[(size_t)s * PN_FRAME], &pcm[(size_t)f * PN_FRAME], PN_FRAME * sizeof(int16_t));
    if (pn_process_host_i16(t, in.data(), out.data(), NULL) || pn_ctx_read_features(t, feat.data(), sil.data())) { rc = -1; msg = pn_last_error(); break; }
    for (int s = 0; s < Bt && !rc; s++) {
      if (sil[s] != kSelftestSilence[f]) { rc = -2; msg = "silence flag of frame " + std::to_string(f) + ", stream " + std::to_string(s); }
      if (memcmp(&feat[(size_t)s * PN_NFEAT], &feat[0], PN_NFEAT * 4)) { rc = -2; msg = "stream " + std::to_string(s) + " differs from stream 0 at frame " + std::to_string(f) + " (same input)"; }
    }
    if (!rc && f == PN_SELFTEST_FRAMES - 1)
      for (int k = 0; k < PN_NFEAT; k++) {
        uint32_t w; memcpy(&w, &feat[k], 4);
        if (w != kSelftestFeat[k]) { rc = -2; msg = "feature " + std::to_string(k) + " of the last frame"; break; }
      }
  }
  pn_ctx_destroy(t); pn_model_free(m);
  if (env && atoi(env) >= 2) fprintf(stderr, "percepnet_hip: DSP self-test device %d front end %d: %s\n", c->device, c->fe_mode, rc ? msg.c_str() : "70 features + 14 silence flags bit-equal to the CPU oracle, 40 streams identical");
  if (rc == -1) { pn_set_error("DSP self-test could not run: %s", msg.c_str()); return -1; }
  if (rc) {
    pn_set_error("DSP self-test FAILED (front end %s): %s does not match the CPU reference's known answer — this build of the DSP "
                 "kernels is not bit-exact (DESIGN.md 4.4); refusing to run", c->fe_mode == FE_SPLIT ? "split" : (c->fe_mode == FE_MONO_G2 ? "g2" : "g4"), msg.c_str());
    return -1;
  }
  done[key] = 0;
  return 0;
}

static int process_dev(pn_ctx *c, const void *d_in, void *d_out, float *d_gr, int is_i16) {
  if (!c || !d_in || !d_out) { pn_set_error("NULL argument"); return -1; }
  PN_ON_DEVICE(c);
  if (c->fe_mode == FE_SPLIT) {
    { Scope sc(c, KF_FE_SPEC_IN);
      pn_launch_fe_spec_in(c->stream, c->tables, c->B, c->t, d_in, is_i16, PN_FRAME, 1.f / 32768.f, c->hist, c->yring, c->eyring, c->dsp_grid_cap); }
    { Scope sc(c, KF_FE_PITCH);
      pn_launch_fe_pitch(c->stream, c->B, c->t, c->hist, c->feat, c->last_period, c->last_gain, nullptr, c->dsp_grid_cap); }
    { Scope sc(c, KF_FE_SPEC_OUT);
      pn_launch_fe_spec_out(c->stream, c->tables, c->B, c->t, c->hist, c->yring, c->eyring, c->last_period, c->Ps, c->feat,
                            c->silence, nullptr, c->dsp_grid_cap); }
  } else {
    Scope sc(c, KF_FRONTEND);
    (c->fe_mode == FE_MONO_G2 ? pn_launch_frontend_g2 : pn_launch_frontend)(c->stream, c->tables, c->B, c->t, d_in, is_i16, PN_FRAME, 1.f / 32768.f,
        c->hist, c->yring, c->eyring, c->Ps, c->feat, c->silence, c->last_period, c->last_gain, nullptr, c->dsp_grid_cap);
  }
  if (launch_rnn(c)) return -1;                        // a refused launch fails the frame (pn_last_error says which layer)
  { Scope sc(c, KF_BACKEND);
    // X(t) == the look-ahead spectrum of frame t-5 (pn_dsp_fe.hip): ring slot (t+1)%6
    const size_t slot = (size_t)((c->t + 1) % 6);
    const float2 *Xs = c->yring + slot * c->B * PN_SPEC_BINS;
    const float *Ex = c->postfilter ? c->eyring + slot * c->B * 36 : nullptr;      // Ex(t) = Ey_lookahead(t-5)
    pn_launch_backend(c->stream, c->tables, c->B, Xs, c->Ps, c->gr, Ex, c->silence, c->synth, d_out, is_i16, c->dsp_grid_cap); }
  if (d_gr) PN_HIP_CHECK(hipMemcpyAsync(d_gr, c->gr, (size_t)c->B * 68 * 4, hipMemcpyDeviceToDevice, c->stream));
  PN_HIP_CHECK(hipGetLastError());
  c->t++; c->tn++;
  if (c->events.size() >= 4096 && flush_events(c)) return -1;   // profiling left on: bound the pending events
  return 0;
}

// Test hook (tests/test_gpu_lifecycle.py): while enabled, every frame asks the fc layer's launcher for a geometry it refuses —
// pn_process_* / pn_submit_host_* / pn_ctx_compute_rnn_host must then return -1 with pn_last_error() naming the launcher,
// never 0 with stale outputs (STRICT contexts have no such refusal: their kernels take any geometry).
extern "C" int pn_ctx_debug_inject_launch_failure(pn_ctx *c, int enable) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  c->inject_bad_launch = enable != 0;
  return 0;
}

extern "C" int pn_ctx_set_postfilter(pn_ctx *c, int enable) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  c->postfilter = enable != 0;
  return 0;
}

extern "C" int pn_process_f32(pn_ctx *c, const float *d_in, float *d_out, float *d_gr) { return process_dev(c, d_in, d_out, d_gr, 0); }
extern "C" int pn_process_i16(pn_ctx *c, const int16_t *d_in, int16_t *d_out, float *d_gr) { return process_dev(c, d_in, d_out, d_gr, 1); }
extern "C" int pn_process_i16_multi(pn_ctx *c, const int16_t *d_in, int16_t *d_out, float *d_gr, int n_frames) {
  if (!c) return -1;
  const size_t fs = (size_t)c->B * PN_FRAME;
  for (int f = 0; f < n_frames; f++)
    if (process_dev(c, d_in + f * fs, d_out + f * fs, d_gr ? d_gr + (size_t)f * c->B * 68 : NULL, 1)) return -1;
  return 0;
}

// ---- per-call active set (pn_active.hip) -----------------------------------------------------------------------------
// ids[0..n): the streams that receive a frame in this call (distinct, any order).  Every other stream keeps ALL of its
// state — as if its rnnoise_process_frame had not been called (denoise.cpp:508-547) — and its rows of d_out / d_gr are
// left as they were; its row of d_in is ignored.  n == n_streams is exactly pn_process_*.
// the active list must name distinct streams of this context; leaves c->act.mark[s] = 1 for the listed ones
static int active_check(pn_ctx *c, const int32_t *ids, int n) {
  if (!c || n < 0 || (n > 0 && !ids)) { pn_set_error("bad argument"); return -1; }
  const int B = c->B;
  if (n > B) { pn_set_error("%d active streams in a context of %d", n, B); return -1; }
  pn_ctx::Active &A = c->act;
  A.mark.assign((size_t)B, 0);
  for (int i = 0; i < n; i++) {
    if (ids[i] < 0 || ids[i] >= B) { pn_set_error("stream id %d out of range [0, %d)", ids[i], B); return -1; }
    if (A.mark[ids[i]]) { pn_set_error("stream id %d listed twice", ids[i]); return -1; }
    A.mark[ids[i]] = 1;
  }
  return 0;
}
static int process_active(pn_ctx *c, const void *d_in, void *d_out, float *d_gr, int is_i16, const int32_t *ids, int n) {
  if (!c || !d_in || !d_out) { pn_set_error("NULL argument"); return -1; }
  if (active_check(c, ids, n)) return -1;
  const int B = c->B;
  pn_ctx::Active &A = c->act;
  if (n == B) return process_dev(c, d_in, d_out, d_gr, is_i16);
  A.inactive.clear();
  for (int s = 0; s < B; s++) if (!A.mark[s]) A.inactive.push_back(s);
  const int ni = (int)A.inactive.size();
  PN_ON_DEVICE(c);
  if (A.cap < ni) {
    // (the old, smaller buffers stay in allocs until destroy: kernels of an earlier call may still be using them)
    int cap = 1024; while (cap < ni) cap *= 2; if (cap > B) cap = B;
    if (dev_alloc(c, (void **)&A.d_ids, (size_t)cap * 4, false) || dev_alloc(c, (void **)&A.save_synth, (size_t)cap * PN_FRAME * 4, false) ||
        dev_alloc(c, (void **)&A.save_out, (size_t)cap * PN_FRAME * 4, false) || dev_alloc(c, (void **)&A.save_gr, (size_t)cap * 68 * 4, false) ||
        dev_alloc(c, (void **)&A.save_period, (size_t)cap * 4, false) || dev_alloc(c, (void **)&A.save_gain, (size_t)cap * 4, false)) return -1;
    for (auto &sl : A.slot) {
      if (sl.ev) PN_HIP_CHECK(hipEventSynchronize(sl.ev));
      if (sl.h) hipHostFree(sl.h);
      sl.h = NULL;
      PN_HIP_CHECK(hipHostMalloc((void **)&sl.h, (size_t)cap * sizeof(int), hipHostMallocDefault));
      if (!sl.ev) PN_HIP_CHECK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    A.cap = cap;
  }
  {
    pn_ctx::IdSlot &sl = A.slot[A.calls++ & 3];
    PN_HIP_CHECK(hipEventSynchronize(sl.ev));             // the copy issued from this slot four calls ago has executed
    memcpy(sl.h, A.inactive.data(), (size_t)ni * sizeof(int));
    PN_HIP_CHECK(hipMemcpyAsync(A.d_ids, sl.h, (size_t)ni * sizeof(int), hipMemcpyHostToDevice, c->stream));
    PN_HIP_CHECK(hipEventRecord(sl.ev, c->stream));
  }
  PnActiveArgs a; memset(&a, 0, sizeof(a));
  a.ids = A.d_ids; a.synth = c->synth; a.last_period = c->last_period; a.last_gain = c->last_gain;
  a.out = d_out; a.out_row_words = is_i16 ? PN_FRAME / 2 : PN_FRAME; a.d_gr = d_gr;
  a.save_synth = A.save_synth; a.save_out = A.save_out; a.save_gr = A.save_gr; a.save_period = A.save_period; a.save_gain = A.save_gain;
  a.hist = c->hist; a.yring = c->yring; a.eyring = c->eyring; a.c1ring = c->c1ring; a.c2ring = c->c2ring; a.rb = c->rb;
  for (int i = 0; i < 4; i++) { a.gru[i] = c->gru[i]; a.gruH[i] = (uint4 *)c->gruH[i]; }
  a.c1ringH = (uint4 *)c->c1ringH; a.c2ringH = (uint4 *)c->c2ringH; a.rbH = (uint4 *)c->rbH;
  a.np = c->c2outH ? (int)shadow_halfs_per_element(c) : 0;     // (a NULL shadow is skipped)
  a.B = B; a.Bp = (long long)c->Bp; a.t = c->t; a.tn = c->tn;
  pn_launch_inactive_save(c->stream, a, ni);
  if (process_dev(c, d_in, d_out, d_gr, is_i16)) {
    // a refused launch: the frame did not complete and the counters did not advance, but the front end may already have
    // written last_period / last_gain and the caller's rows may hold anything — the SKIPPED streams still get their in-place
    // state and their output rows back, as the header promises (the error message of the refusal is kept)
    a.restore_only = 1;
    pn_launch_inactive_fixup(c->stream, a, ni);
    return -1;
  }
  pn_launch_inactive_fixup(c->stream, a, ni);           // a.t / a.tn: the counters the frame above ran with
  PN_HIP_CHECK(hipGetLastError());
  return 0;
}
extern "C" int pn_process_f32_active(pn_ctx *c, const float *d_in, float *d_out, float *d_gr, const int32_t *ids, int n) { return process_active(c, d_in, d_out, d_gr, 0, ids, n); }
extern "C" int pn_process_i16_active(pn_ctx *c, const int16_t *d_in, int16_t *d_out, float *d_gr, const int32_t *ids, int n) { return process_active(c, d_in, d_out, d_gr, 1, ids, n); }

static int process_host(pn_ctx *c, const void *h_in, void *h_out, float *h_gr, int is_i16) {
  if (!c || !h_in || !h_out) { pn_set_error("NULL argument"); return -1; }
  PN_ON_DEVICE(c);
  if (pipe_drain(c)) return -1;                      // frames still in flight on the pipelined path use io_in/io_out
  const size_t nbytes = (size_t)c->B * PN_FRAME * (is_i16 ? 2 : 4);
  PN_HIP_CHECK(hipMemcpyAsync(c->io_in, h_in, nbytes, hipMemcpyHostToDevice, c->stream));
  if (process_dev(c, c->io_in, c->io_out, NULL, is_i16)) return -1;
  PN_HIP_CHECK(hipMemcpyAsync(h_out, c->io_out, nbytes, hipMemcpyDeviceToHost, c->stream));
  if (h_gr) PN_HIP_CHECK(hipMemcpyAsync(h_gr, c->gr, (size_t)c->B * 68 * 4, hipMemcpyDeviceToHost, c->stream));
  PN_HIP_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int pn_process_host_f32(pn_ctx *c, const float *h_in, float *h_out, float *h_gr) { return process_host(c, h_in, h_out, h_gr, 0); }
extern "C" int pn_process_host_i16(pn_ctx *c, const int16_t *h_in, int16_t *h_out, float *h_gr) { return process_host(c, h_in, h_out, h_gr, 1); }

// ---- pipelined host-buffer path ----------------------------------------------------------------------------
// Copy-in, the 13 launches and copy-out of consecutive frames on three streams with double-buffered device staging:
//   h2d stream:     H2D(t) ........ H2D(t+1) ......
//   compute stream:        frame(t) ........ frame(t+1) ...
//   d2h stream:                     D2H(t) ......... D2H(t+1)
// Slot k = t & 1 is reused by frame t+2 only after the host has seen frame t delivered, which also bounds the frames
// in flight to two.
// Do `busy` and `cand` share a hardware queue?  A 1 ms sleeper goes to `busy`, then a 64-byte copy to `cand`: on a queue of
// its own the copy lands while the sleeper runs; on a shared queue it lands after it.
static int pipe_streams_share(pn_ctx *c, hipStream_t busy, hipStream_t cand, void *d_scratch, void *h_scratch, bool *shared) {
  struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) hipEventDestroy(e); } } ek, ec;     // destroyed on every exit path
  PN_HIP_CHECK(hipEventCreateWithFlags(&ek.e, hipEventDisableTiming));
  PN_HIP_CHECK(hipEventCreateWithFlags(&ec.e, hipEventDisableTiming));
  if (pn_launch_spin(busy, 100000)) { pn_set_error("queue probe: launch failed"); return -1; }
  PN_HIP_CHECK(hipEventRecord(ek.e, busy));
  PN_HIP_CHECK(hipMemcpyAsync(d_scratch, h_scratch, 64, hipMemcpyHostToDevice, cand));
  PN_HIP_CHECK(hipEventRecord(ec.e, cand));
  PN_HIP_CHECK(hipEventSynchronize(ec.e));
  *shared = hipEventQuery(ek.e) == hipSuccess;
  PN_HIP_CHECK(hipEventSynchronize(ek.e));
  (void)c;
  return 0;
}

// One copy stream.  how: 'n' default priority unprobed, 'h' / 'l' a priority stream, 'a' (the default) a default-priority
// stream that shares its queue with none of `others` — up to 6 candidates (the rejected ones stay alive until the end, so
// that the runtime's least-used-queue choice moves on), else the priority stream `fallback`.
static int pipe_make_stream(pn_ctx *c, hipStream_t *out, char how, int prio, char fallback, const std::vector<hipStream_t> &others, char *kind) {
  if (how == 'h' || how == 'l') {
    int lo = 0, hi = 0;
    PN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    PN_HIP_CHECK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, how == 'h' ? hi : lo));
    *kind = how;
    return 0;
  }
  if (how == 'n') { PN_HIP_CHECK(hipStreamCreateWithFlags(out, hipStreamNonBlocking)); *kind = 'n'; return 0; }
  // everything the probe owns is released on EVERY exit path (advisor, round 5: the early returns of PN_HIP_CHECK leaked the
  // scratch buffers and the rejected streams)
  struct Probe {
    void *h_scratch = NULL, *d_scratch = NULL; std::vector<hipStream_t> rejected; hipStream_t cur = nullptr;
    ~Probe() { for (hipStream_t s : rejected) hipStreamDestroy(s); if (cur) hipStreamDestroy(cur); if (d_scratch) hipFree(d_scratch); if (h_scratch) hipHostFree(h_scratch); }
  } pr;
  PN_HIP_CHECK(hipHostMalloc(&pr.h_scratch, 64, hipHostMallocDefault));
  memset(pr.h_scratch, 0, 64);
  PN_HIP_CHECK(hipMalloc(&pr.d_scratch, 64));
  for (int attempt = 0; attempt < 6; attempt++) {
    PN_HIP_CHECK(hipStreamCreateWithFlags(&pr.cur, hipStreamNonBlocking));
    PN_HIP_CHECK(hipMemcpyAsync(pr.d_scratch, pr.h_scratch, 64, hipMemcpyHostToDevice, pr.cur));       // first use of the stream, not timed
    PN_HIP_CHECK(hipStreamSynchronize(pr.cur));
    bool bad = false;
    for (hipStream_t o : others) {
      bool sh = false;
      if (pipe_streams_share(c, o, pr.cur, pr.d_scratch, pr.h_scratch, &sh)) return -1;
      if (sh) { bad = true; break; }
    }
    if (!bad) { *out = pr.cur; pr.cur = nullptr; *kind = 'n'; return 0; }
    pr.rejected.push_back(pr.cur); pr.cur = nullptr;
  }
  PN_HIP_CHECK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio));
  *kind = fallback;
  return 0;
}

static int pipe_init_body(pn_ctx *c) {
  pn_ctx::Pipe &P = c->pipe;
  // Each of the three streams of the pipeline needs a hardware queue of its own.  HIP multiplexes the streams of one
  // priority over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default): in a process that already owns a handful of
  // streams (torch's pools) a copy stream can land on the queue of the compute stream, and the copy of frame t - 1 then
  // runs BEHIND the kernels of frame t instead of beside them (10.6 instead of 9.5 ms per frame at 65 536 streams,
  // profiles/r04q_host_pipeline_queues.log).  Queues of different priorities are never shared — but two copy streams at
  // the non-default priorities cost every kernel of the compute stream ~50 us (back-to-back frames 10.06 instead of
  // 9.48 ms at 65 536 streams; one priority stream costs nothing: profiles/r05_host_pipeline.log).  So: default-priority
  // streams, each PROBED against the streams it must not share a queue with (pipe_make_stream), a priority stream only
  // as the fallback.  PN_PIPE_PRIO = two letters (h2d, d2h) of h / n / l overrides (tools/host_pipeline_probe.py).
  int prio_least = 0, prio_greatest = 0;
  PN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  const char *pp = getenv("PN_PIPE_PRIO");
  if (pp && strlen(pp) != 2) pp = NULL;
  if (pipe_make_stream(c, &P.h2d, pp ? pp[0] : 'a', prio_greatest, 'h', busy_streams(c), &P.kind[0])) return -1;     // incl. the row-range chains' streams
  if (pipe_make_stream(c, &P.d2h, pp ? pp[1] : 'a', prio_least, 'l', busy_streams(c), &P.kind[1])) return -1;
  for (int k = 0; k < 2; k++) {
    PN_HIP_CHECK(hipEventCreateWithFlags(&P.in_ready[k], hipEventDisableTiming));
    PN_HIP_CHECK(hipEventCreateWithFlags(&P.done[k], hipEventDisableTiming));
    PN_HIP_CHECK(hipEventCreateWithFlags(&P.delivered[k], hipEventDisableTiming));
  }
  const size_t io_bytes = (size_t)c->B * PN_FRAME * 4, gr_bytes = (size_t)c->B * 68 * 4;
  P.in[0] = c->io_in; P.out[0] = c->io_out;
  if (!P.in[1] && dev_alloc(c, &P.in[1], io_bytes, false)) return -1;          // device buffers belong to the context (freed with it): a retry reuses them
  if (!P.out[1] && dev_alloc(c, &P.out[1], io_bytes, false)) return -1;
  for (int k = 0; k < 2; k++) if (!P.gr[k] && dev_alloc(c, (void **)&P.gr[k], gr_bytes, false)) return -1;
  return 0;
}
// The first pn_submit_host_* call (or pn_host_pipeline_prepare) builds the pipeline: up to 6 attempts x 3 pairings of a 1 ms
// probe on the context's stream — tens of milliseconds, and not legal while that stream is being captured.  A caller on a
// real-time clock calls pn_host_pipeline_prepare once, before its first frame arrives.  A failed build leaves NOTHING behind
// (streams and events of the partial pipeline are destroyed; the next call starts over).
static int pipe_init(pn_ctx *c) {
  pn_ctx::Pipe &P = c->pipe;
  if (P.init) return 0;
  if (pipe_init_body(c)) {
    if (P.h2d) { hipStreamDestroy(P.h2d); P.h2d = nullptr; }
    if (P.d2h) { hipStreamDestroy(P.d2h); P.d2h = nullptr; }
    for (int k = 0; k < 2; k++) {
      if (P.in_ready[k]) { hipEventDestroy(P.in_ready[k]); P.in_ready[k] = nullptr; }
      if (P.done[k]) { hipEventDestroy(P.done[k]); P.done[k] = nullptr; }
      if (P.delivered[k]) { hipEventDestroy(P.delivered[k]); P.delivered[k] = nullptr; }
    }
    return -1;
  }
  P.init = true;
  return 0;
}
extern "C" int pn_host_pipeline_prepare(pn_ctx *c) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  PN_ON_DEVICE(c);
  return pipe_init(c);
}

static int pipe_drain(pn_ctx *c) {
  if (!c->pipe.init) return 0;
  PN_HIP_CHECK(hipStreamSynchronize(c->pipe.h2d));
  PN_HIP_CHECK(hipStreamSynchronize(c->stream));
  PN_HIP_CHECK(hipStreamSynchronize(c->pipe.d2h));
  return 0;
}

static int process_active(pn_ctx *c, const void *d_in, void *d_out, float *d_gr, int is_i16, const int32_t *ids, int n);
static int active_check(pn_ctx *c, const int32_t *ids, int n);
// ids != NULL or n >= 0 with active = true: only the listed streams advance (pn_submit_host_*_active)
static int submit_host(pn_ctx *c, const void *h_in, void *h_out, float *h_gr, int is_i16, bool active = false, const int32_t *ids = NULL, int n = 0) {
  if (!c || !h_in || !h_out) { pn_set_error("NULL argument"); return -1; }
  if (active && active_check(c, ids, n)) return -1;          // refused before the frame takes a pipeline slot
  PN_ON_DEVICE(c);
  if (pipe_init(c)) return -1;
  pn_ctx::Pipe &P = c->pipe;
  const int k = (int)(P.submitted & 1);
  if (P.submitted >= 2) PN_HIP_CHECK(hipEventSynchronize(P.delivered[k]));     // frame submitted-2 delivered: slot k is free
  const size_t nbytes = (size_t)c->B * PN_FRAME * (is_i16 ? 2 : 4);
  PN_HIP_CHECK(hipMemcpyAsync(P.in[k], h_in, nbytes, hipMemcpyHostToDevice, P.h2d));
  PN_HIP_CHECK(hipEventRecord(P.in_ready[k], P.h2d));
  PN_HIP_CHECK(hipStreamWaitEvent(c->stream, P.in_ready[k], 0));
  if (active ? process_active(c, P.in[k], P.out[k], h_gr ? P.gr[k] : NULL, is_i16, ids, n)
             : process_dev(c, P.in[k], P.out[k], h_gr ? P.gr[k] : NULL, is_i16)) return -1;
  PN_HIP_CHECK(hipEventRecord(P.done[k], c->stream));
  PN_HIP_CHECK(hipStreamWaitEvent(P.d2h, P.done[k], 0));
  PN_HIP_CHECK(hipMemcpyAsync(h_out, P.out[k], nbytes, hipMemcpyDeviceToHost, P.d2h));
  if (h_gr) PN_HIP_CHECK(hipMemcpyAsync(h_gr, P.gr[k], (size_t)c->B * 68 * 4, hipMemcpyDeviceToHost, P.d2h));
  PN_HIP_CHECK(hipEventRecord(P.delivered[k], P.d2h));
  P.submitted++;
  return 0;
}
// "nn" / "hl" / ...: how the two copy streams of the pipelined path were obtained (pipe_init); "" before the first submit
extern "C" const char *pn_ctx_pipe_streams(pn_ctx *c) { return (c && c->pipe.init) ? c->pipe.kind : ""; }
extern "C" int pn_submit_host_f32(pn_ctx *c, const float *h_in, float *h_out, float *h_gr) { return submit_host(c, h_in, h_out, h_gr, 0); }
extern "C" int pn_submit_host_i16(pn_ctx *c, const int16_t *h_in, int16_t *h_out, float *h_gr) { return submit_host(c, h_in, h_out, h_gr, 1); }
// The pipelined path with a per-call active set: rows of h_in of skipped streams are ignored; their rows of h_out / h_gr are
// UNSPECIFIED (the device staging rows are restored to what they held two frames earlier and copied out with the rest).
extern "C" int pn_submit_host_f32_active(pn_ctx *c, const float *h_in, float *h_out, float *h_gr, const int32_t *ids, int n) { return submit_host(c, h_in, h_out, h_gr, 0, true, ids, n); }
extern "C" int pn_submit_host_i16_active(pn_ctx *c, const int16_t *h_in, int16_t *h_out, float *h_gr, const int32_t *ids, int n) { return submit_host(c, h_in, h_out, h_gr, 1, true, ids, n); }
extern "C" int pn_host_wait(pn_ctx *c) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  PN_ON_DEVICE(c);
  return pipe_drain(c);
}
// Frames of the pipelined host path whose output copy has landed in the caller's buffer (non-blocking: event queries on the
// at most two frames in flight; delivery is in order).  A caller on a real-time clock polls this between arrivals to
// timestamp each frame's delivery (bench.py: arrival-to-delivery latency), which the blocking pn_submit_host_* cannot show.
extern "C" int64_t pn_host_frames_delivered(pn_ctx *c) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  pn_ctx::Pipe &P = c->pipe;
  if (!P.init || P.submitted == 0) return 0;
  DeviceGuard _dg(c->device);
  if (!_dg.ok) { pn_set_error("hipSetDevice(%d) failed", c->device); return -1; }
  int64_t done = P.submitted >= 2 ? P.submitted - 2 : 0;     // everything older than the two newest was waited for by a submit
  for (int64_t f = done; f < P.submitted; f++) {
    const hipError_t e = hipEventQuery(P.delivered[f & 1]);
    if (e == hipSuccess) done = f + 1;
    else { if (e != hipErrorNotReady) { pn_set_error("hipEventQuery failed: %s", hipGetErrorString(e)); return -1; } (void)hipGetLastError(); break; }
  }
  return done;
}
// NUMA placement of a host thread that feeds one device: bind the CALLING THREAD to the CPUs of the NUMA node the device hangs
// off (read from /sys/bus/pci/devices/<bdf>/numa_node), BEFORE it allocates its pinned buffers — first touch then places
// them next to the GPU's root port.  Returns the node (>= 0) when bound, -1 when nothing was changed (msg says why: no
// affinity reported, sysfs unreadable, ...).  Never an error for the caller: an unbound thread is merely slower.
#include <sched.h>
extern "C" int pn_bind_thread_to_device_numa(int device, char *msg, size_t msg_bytes) {
#define PN_SAY(...) do { if (msg && msg_bytes) snprintf(msg, msg_bytes, __VA_ARGS__); } while (0)
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) { (void)hipGetLastError(); PN_SAY("device %d: no PCI bus id", device); return -1; }
  for (char *p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells it lower-case
  char path[256];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  int node = -1;
  if (!f || fscanf(f, "%d", &node) != 1) { if (f) fclose(f); PN_SAY("device %d (%s): %s unreadable", device, bdf, path); return -1; }
  fclose(f);
  if (node < 0) { PN_SAY("device %d (%s): the platform reports no NUMA affinity (numa_node = -1)", device, bdf); return -1; }
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  f = fopen(path, "r");
  char list[4096] = {0};
  if (!f || !fgets(list, sizeof(list), f)) { if (f) fclose(f); PN_SAY("device %d (%s): node %d has no cpulist", device, bdf, node); return -1; }
  fclose(f);
  cpu_set_t allowed, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) { PN_SAY("sched_getaffinity failed"); return -1; }
  int n = 0;
  for (char *p = list; *p && *p != '\n';) {              // "0-3,8,10-11"
    char *e; const long lo = strtol(p, &e, 10); long hi = lo;
    if (e == p) break;
    if (*e == '-') { p = e + 1; hi = strtol(p, &e, 10); }
    for (long c = lo; c <= hi && c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) { CPU_SET(c, &want); n++; }
    p = (*e == ',') ? e + 1 : e;
    if (*e != ',') break;
  }
  if (!n) { PN_SAY("device %d (%s): node %d has no CPU inside this thread's affinity mask", device, bdf, node); return -1; }
  if (sched_setaffinity(0, sizeof(want), &want) != 0) { PN_SAY("device %d (%s): sched_setaffinity failed", device, bdf); return -1; }
  PN_SAY("device %d (%s): thread bound to the %d CPUs of NUMA node %d", device, bdf, n, node);
  return node;
}
#undef PN_SAY
extern "C" void *pn_host_alloc(size_t bytes) {
  void *p = NULL;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { pn_set_error("hipHostMalloc(%zu) failed", bytes); return NULL; }
  return p;
}
extern "C" void pn_host_free(void *p) { if (p) hipHostFree(p); }

extern "C" int pn_ctx_read_features(pn_ctx *c, float *h_feat, int32_t *h_silence) {
  if (!c) return -1;
  PN_ON_DEVICE(c);
  if (h_feat)
    PN_HIP_CHECK(hipMemcpy2DAsync(h_feat, PN_NFEAT * 4, c->feat, PN_FEAT_STRIDE * 4, PN_NFEAT * 4, c->B, hipMemcpyDeviceToHost, c->stream));
  if (h_silence) PN_HIP_CHECK(hipMemcpyAsync(h_silence, c->silence, (size_t)c->B * 4, hipMemcpyDeviceToHost, c->stream));
  PN_HIP_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

// Device-side twin of pn_ctx_read_features: asynchronous copies on the context's stream into caller-owned device
// buffers (d_feat [n_streams][70], d_silence [n_streams] int32; either may be NULL).
extern "C" int pn_ctx_read_features_dev(pn_ctx *c, float *d_feat, int32_t *d_silence) {
  if (!c) return -1;
  PN_ON_DEVICE(c);
  if (d_feat)
    PN_HIP_CHECK(hipMemcpy2DAsync(d_feat, PN_NFEAT * 4, c->feat, PN_FEAT_STRIDE * 4, PN_NFEAT * 4, c->B, hipMemcpyDeviceToDevice, c->stream));
  if (d_silence) PN_HIP_CHECK(hipMemcpyAsync(d_silence, c->silence, (size_t)c->B * 4, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

extern "C" int pn_ctx_compute_rnn_host(pn_ctx *c, const float *h_feat, float *h_gr) {
  if (!c || !h_feat || !h_gr) { pn_set_error("NULL argument"); return -1; }
  PN_ON_DEVICE(c);
  if (pipe_drain(c)) return -1;                      // frames in flight on the pipelined path own feat/gr
  PN_HIP_CHECK(hipMemcpy2DAsync(c->feat, PN_FEAT_STRIDE * 4, h_feat, PN_NFEAT * 4, PN_NFEAT * 4, c->B, hipMemcpyHostToDevice, c->stream));
  if (launch_rnn(c)) return -1;
  PN_HIP_CHECK(hipMemcpyAsync(h_gr, c->gr, (size_t)c->B * 68 * 4, hipMemcpyDeviceToHost, c->stream));
  PN_HIP_CHECK(hipStreamSynchronize(c->stream));
  PN_HIP_CHECK(hipGetLastError());
  c->tn++;                                           // only the network's rings advance; the DSP rings keep their frame
  return 0;
}

// ---- network state <-> host arrays in the reference's RNNState layout (nnet_data.h:28-38) ---------------------------
// The conv FIFOs are rings here: the ks-1 previous layer inputs, oldest first, live in slots (tn+1+j) % ks, j = 0..ks-2
// (launch_rnn reads panels (tn+1+j) % ks for j = 0..ks-1, the last one being the slot the current step writes).
// The GRU states are ping-pong pairs: buffer tn & 1 holds the state the next step reads.
static int rnn_state_copy(pn_ctx *c, bool to_device, float *conv1, float *conv2, float *const gru[4], float *rb) {
  PN_ON_DEVICE(c);
  if (pipe_drain(c)) return -1;
  // fp16-operand and split-precision modes: the fp32 buffers are complete (every layer stores fp32 next to its operand
  // shadow), so a store reads them as in the fp32 modes and a load re-derives the shadows from the loaded fp32 values
  const bool x3 = c->nn_mode == PN_NN_MFMA_X3 && to_device, f16 = c->nn_mode == PN_NN_MFMA_F16 && to_device;
  const size_t B = c->B, Bp = c->Bp; const int64_t t = c->tn;
  const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
  auto cp2d = [&](float *host, size_t hpitch, float *dev, size_t dpitch, size_t width) -> hipError_t {
    return to_device ? hipMemcpy2DAsync(dev, dpitch * 4, host, hpitch * 4, width * 4, B, kind, c->stream)
                     : hipMemcpy2DAsync(host, hpitch * 4, dev, dpitch * 4, width * 4, B, kind, c->stream);
  };
  int split_rc = 0;
  auto resplit = [&](float *dev, int width) {
    if (x3) split_rc |= pn_launch_split_x3(c->stream, dev, width, width, shadow(c, dev), (int)Bp, 2);
    if (f16) split_rc |= pn_launch_split_x3(c->stream, dev, width, width, shadow(c, dev), (int)Bp, 1);
    if (c->direct && to_device && shadow(c, dev)) split_rc |= pn_launch_split_d(c->stream, dev, width, width, shadow(c, dev), (int)Bp);   // (GRU states; the conv FIFOs have no shadow there)
  };
  if (conv1) for (int j = 0; j < 4; j++) {
    float *d = c->c1ring + (size_t)((t + 1 + j) % 5) * Bp * 128;
    PN_HIP_CHECK(cp2d(conv1 + j * 128, 4 * 128, d, 128, 128)); resplit(d, 128);
  }
  if (conv2) for (int j = 0; j < 2; j++) {
    float *d = c->c2ring + (size_t)((t + 1 + j) % 3) * Bp * 512;
    PN_HIP_CHECK(cp2d(conv2 + j * 512, 2 * 512, d, 512, 512)); resplit(d, 512);
  }
  for (int i = 0; i < 4; i++)
    if (gru[i]) { float *d = c->gru[i] + (size_t)(t & 1) * Bp * 512; PN_HIP_CHECK(cp2d(gru[i], 512, d, 512, 512)); resplit(d, 512); }
  if (rb) { float *d = c->rb + (size_t)(t & 1) * Bp * 128; PN_HIP_CHECK(cp2d(rb, 128, d, 128, 128)); resplit(d, 128); }
  PN_HIP_CHECK(hipStreamSynchronize(c->stream));
  return split_rc ? -1 : 0;                  // a refused shadow-operand split (pn_launch_split_x3) fails the call, like any refused launch
}
extern "C" int pn_ctx_set_rnn_state_host(pn_ctx *c, const float *conv1, const float *conv2, const float *gru1, const float *gru2,
                                         const float *gru3, const float *gru_gb, const float *gru_rb) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  float *g[4] = {(float *)gru1, (float *)gru2, (float *)gru3, (float *)gru_gb};
  return rnn_state_copy(c, true, (float *)conv1, (float *)conv2, g, (float *)gru_rb);
}
extern "C" int pn_ctx_get_rnn_state_host(pn_ctx *c, float *conv1, float *conv2, float *gru1, float *gru2, float *gru3,
                                         float *gru_gb, float *gru_rb) {
  if (!c) { pn_set_error("NULL argument"); return -1; }
  float *g[4] = {gru1, gru2, gru3, gru_gb};
  return rnn_state_copy(c, false, conv1, conv2, g, gru_rb);
}

// Debug tap (tests/tools only): copy an internal device buffer to the host.
// which: 0 feat[B][128], 1 c1ring[5][B][128], 2 c2ring[3][B][512], 3 c2out[B][512],
//        4..7 gru[i][2][B][512], 8 rb[2][B][128], 9 gr[B][68], 10 look-ahead spectra ring, 11 comb-filtered spectrum,
//        12 history ring, 13 last_period int32 [B].  Returns the byte count.
extern "C" long long pn_ctx_debug_copy(pn_ctx *c, int which, void *dst, long long max_bytes) {
  if (!c || !dst) return -1;
  const size_t B = c->B, Bp = c->Bp;
  const void *src = NULL; size_t n = 0;
  switch (which) {
    case 0: src = c->feat; n = Bp * PN_FEAT_STRIDE * 4; break;
    case 1: src = c->c1ring; n = 5 * Bp * 128 * 4; break;
    case 2: src = c->c2ring; n = 3 * Bp * 512 * 4; break;
    case 3: src = c->c2out; n = Bp * 512 * 4; break;
    case 4: case 5: case 6: case 7: src = c->gru[which - 4]; n = 2 * Bp * 512 * 4; break;
    case 8: src = c->rb; n = 2 * Bp * 128 * 4; break;
    case 9: src = c->gr; n = B * 68 * 4; break;
    case 10: src = c->yring; n = 6 * B * PN_SPEC_BINS * sizeof(float2); break;     // look-ahead spectra ring [6][B][400]
    case 11: src = c->Ps; n = B * PN_SPEC_BINS * sizeof(float2); break;            // comb-filtered spectrum [B][400]
    case 12: src = c->hist; n = B * PN_HIST_STRIDE * 4; break;                     // history ring
    case 13: src = c->last_period; n = B * 4; break;                               // pitch period of the last frame, int32 [B]
    default: pn_set_error("bad debug buffer id"); return -1;
  }
  if ((long long)n > max_bytes) { pn_set_error("debug buffer needs %zu bytes", n); return -1; }
  PN_ON_DEVICE(c);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
  if (hipMemcpy(dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) { pn_set_error("debug copy failed"); return -1; }
  return (long long)n;
}
