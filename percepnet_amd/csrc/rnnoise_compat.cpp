// The reference's frame-engine interface on top of the batched HIP engine.
//
// The reference declares its API in src/rnnoise.h:49-68 WITHOUT extern "C" and builds it from
// .cpp files, so the symbols an existing caller (src/main.cpp:30-39) links against are the
// Itanium-mangled ones (_Z14rnnoise_createP8RNNModel, _Z21rnnoise_process_frameP12DenoiseState
// PfPKfP8_IO_FILE, ...).  This file defines functions with exactly those prototypes in the global
// C++ namespace — the compiler emits exactly those symbols — plus extern "C" `_c` spellings.
//
// One DenoiseState = a batch-of-one context.  That is correct but launch-bound (15 kernel
// launches and two PCIe hops per 10 ms frame); throughput lives in the batched pn_* API.  N handles of one model share
// ONE DEVICE copy of the weights (pn_context.cpp: SharedWeights, found by the model's SHA-256).  Host side (round 6): a handle
// keeps NO copy of the model.  The host pn_model a context is created from comes from
//   * the OwnedModel behind an RNNModel that rnnoise_model_from_file returned (the arrays ARE that pn_model's: zero copies);
//   * one process-wide pn_model of the link-time percepnet_model_orig (immutable .rodata) / of the PERCEPNET_MODEL file
//     (keyed by path, size and mtime), built on first use and kept;
//   * a temporary built from a caller's own RNNModel — borrowed, mutable memory, so nothing is keyed on its address — and freed
//     as soon as the context exists (a context retains no host pointer).
// Round 5 copied 32 MB and hashed them twice for every rnnoise_init and kept the copy for the life of the handle.
#include "pn_common.h"
#include "../../include/percepnet_hip.h"
// the reference's own (C++-mangled) entry points: exported like the C-ABI (the library is built with -fvisibility=hidden)
#define PN_REF_EXPORT __attribute__((visibility("default")))
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <map>
#include <mutex>
#include <string>

// link-time model of the reference (denoise.cpp:49-51): generated nnet_data.cpp defines it.
// Weak: a caller may instead pass a model explicitly or set PERCEPNET_MODEL=<file.pnw>.
extern const RNNModel percepnet_model_orig __attribute__((weak));

struct DenoiseState {
  uint32_t magic;
  pn_ctx *ctx;
  pn_model *unused_model;        // (round 5 kept a private 32 MB host copy here)
  float gr[68];
  int failed;
};
#define DS_MAGIC 0x504e4453u

PN_REF_EXPORT int rnnoise_get_size() { return (int)sizeof(DenoiseState); }

// rnnoise_init (denoise.cpp:259-280): zero state, bind the model.  Returns 0; on failure the
// state is left inert (process_frame then outputs silence) and pn_last_error() says why —
// the reference has no error path at all here.
// Still returns 0 (existing callers ignore the value and the reference cannot fail here), but an inert state is never
// silent: one line on stderr says why no audio will come out.
static int env_device() { const char *d = getenv("PERCEPNET_DEVICE"); return d ? atoi(d) : 0; }
// PERCEPNET_STRICT=1: reference-order network (bit-exact); PERCEPNET_X3=1: split-precision network; default fp32 MFMA
static int env_mode() {
  const char *s = getenv("PERCEPNET_STRICT");
  if (s && atoi(s)) return PN_NN_STRICT;
  const char *x = getenv("PERCEPNET_X3");
  return (x && atoi(x)) ? PN_NN_MFMA_X3 : PN_NN_MFMA;
}

// ---- host models shared between handles ------------------------------------------------------------------------------
struct OwnedModel;
static pn_model *owned_model_of(const RNNModel *m);          // the pn_model behind an RNNModel of rnnoise_model_from_file, or NULL
static std::mutex g_host_mu;                                 // guards the two process-wide models and the live-state registry
static pn_model *g_linked_model = NULL;                      // of percepnet_model_orig
static pn_model *g_env_model = NULL; static std::string g_env_key;   // of the PERCEPNET_MODEL file: path | size | mtime
static std::map<const DenoiseState *, pn_ctx *> g_live;      // states that own a context, and which (the registry, not the state's
                                                             // memory, is the authority: the memory may be fresh from malloc)

// -> the host model to create a context from; *temp = it is the caller's to free once the context exists
static pn_model *host_model_for(RNNModel *model, bool *temp) {
  *temp = false;
  if (model) {
    if (pn_model *pm = owned_model_of(model)) return pm;
    if (&percepnet_model_orig && model == &percepnet_model_orig) {
      if (!g_linked_model) g_linked_model = pn_model_from_rnnmodel(model);
      return g_linked_model;
    }
    *temp = true;
    return pn_model_from_rnnmodel(model);
  }
  if (&percepnet_model_orig) {
    if (!g_linked_model) g_linked_model = pn_model_from_rnnmodel(&percepnet_model_orig);
    return g_linked_model;
  }
  if (const char *path = getenv("PERCEPNET_MODEL")) {
    struct stat sb;
    if (stat(path, &sb) != 0) { fprintf(stderr, "percepnet_hip: rnnoise_init: cannot open PERCEPNET_MODEL=%s\n", path); return NULL; }
    const std::string key = std::string(path) + "|" + std::to_string((long long)sb.st_size) + "|" + std::to_string((long long)sb.st_mtime);
    if (g_env_model && key == g_env_key) return g_env_model;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "percepnet_hip: rnnoise_init: cannot open PERCEPNET_MODEL=%s\n", path); return NULL; }
    pn_model *pm = pn_model_from_file(f);
    fclose(f);
    if (pm) { g_env_model = pm; g_env_key = key; }      // the previous file's model (if any) may still be creating a context on another
    return pm;                                            // thread only under g_host_mu, which we hold: safe to replace; it is leaked by design
  }                                                       // (one per CHANGED file per process) rather than freed under a reader
  return NULL;
}

PN_REF_EXPORT int rnnoise_init(DenoiseState *st, RNNModel *model) {
  std::lock_guard<std::mutex> lk(g_host_mu);
  // (re-)initialising memory the registry knows as a live state — rnnoise_init twice, or a state that was free()d without
  // rnnoise_destroy and whose address came back from malloc: the reference leaks nothing in either case (its state is plain
  // memory); a context is device memory and a stream, so the old one is released first
  { auto it = g_live.find(st); if (it != g_live.end()) { pn_ctx_destroy(it->second); g_live.erase(it); } }
  memset(st, 0, sizeof(*st));
  st->magic = DS_MAGIC;
  bool temp = false;
  pn_model *pm = host_model_for(model, &temp);
  if (!pm) {
    fprintf(stderr, "percepnet_hip: rnnoise_init: no model (pass an RNNModel, link a generated nnet_data.cpp defining "
                    "percepnet_model_orig, or set PERCEPNET_MODEL=<file.pnw>)%s%s: the state is INERT, "
                    "rnnoise_process_frame will output silence\n", pn_last_error()[0] ? ": " : "", pn_last_error());
    return 0;
  }
  st->ctx = pn_ctx_create(pm, env_device(), 1, env_mode(), NULL);
  if (temp) pn_model_free(pm);                            // a context keeps no host pointer into the model it was created from
  if (!st->ctx)
    fprintf(stderr, "percepnet_hip: rnnoise_init: %s: the state is INERT, rnnoise_process_frame will output silence "
                    "(this library has no CPU fallback)\n", pn_last_error());
  else g_live[st] = st->ctx;
  return 0;
}

PN_REF_EXPORT DenoiseState *rnnoise_create(RNNModel *model) {
  DenoiseState *st = (DenoiseState *)malloc(rnnoise_get_size());
  if (st) rnnoise_init(st, model);
  return st;
}

PN_REF_EXPORT void rnnoise_destroy(DenoiseState *st) {
  if (!st) return;
  {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_live.find(st);
    if (it != g_live.end()) { pn_ctx_destroy(it->second); g_live.erase(it); }
  }
  free(st);
}

// rnnoise_process_frame (denoise.cpp:508-547): 480 floats in -> 480 floats out (may alias), the
// 68-float g|r tap appended to f_feature (the reference requires it non-NULL; NULL is accepted
// here).  Always returns 0 like the reference.
PN_REF_EXPORT float rnnoise_process_frame(DenoiseState *st, float *out, const float *in, FILE *f_feature) {
  if (!st || st->magic != DS_MAGIC || !st->ctx) { if (out) memset(out, 0, PN_FRAME_SIZE * sizeof(float)); return 0; }
  float tmp[PN_FRAME_SIZE];
  memcpy(tmp, in, sizeof(tmp));
  if (pn_process_host_f32(st->ctx, tmp, out, st->gr) != 0) {     // HIP error: defined output, no stale tap, one message
    memset(out, 0, PN_FRAME_SIZE * sizeof(float));
    if (!st->failed) { st->failed = 1; fprintf(stderr, "percepnet_hip: rnnoise_process_frame: %s\n", pn_last_error()); }
    return 0;
  }
  if (f_feature) { fwrite(st->gr, sizeof(float), 34, f_feature); fwrite(st->gr + 34, sizeof(float), 34, f_feature); }
  return 0;
}

// compute_rnn (rnnoise.h:68, rnn.cpp:42-81) on a caller-owned RNNState: exported with the reference's prototype (the
// compiler emits _Z11compute_rnnP8RNNStatePfS1_PKf).  The state lives in the caller's host arrays as in the reference;
// each call uploads it to a batch-of-one context cached per model, runs the ten layers on the GPU and writes the new
// state back.  Correct but launch- and PCIe-bound by construction — batched callers use pn_ctx_compute_rnn_host.
// The cache is keyed on the RNNModel's address; rnnoise_model_free drops the entry (a new model allocated at a freed
// address must not run with the old weights); a failed creation is retried on the next call, not cached.
// PERCEPNET_DEVICE / PERCEPNET_STRICT are read when an entry is created.  Entries of models that are never freed (the
// link-time percepnet_model_orig) live until process exit: tearing HIP objects down from a library destructor races the
// HIP runtime's own exit handlers, so that is deliberately left to the OS.
static std::mutex g_rnn_mu;
struct RnnEntry { pn_model *m; pn_ctx *c; };
static std::map<const RNNModel *, RnnEntry> g_rnn_ctx;
static void rnn_cache_drop(const RNNModel *key) {          // caller holds g_rnn_mu
  auto it = g_rnn_ctx.find(key);
  if (it == g_rnn_ctx.end()) return;
  pn_ctx_destroy(it->second.c); pn_model_free(it->second.m);
  g_rnn_ctx.erase(it);
}

PN_REF_EXPORT void compute_rnn(RNNState *rnn, float *gains, float *strengths, const float *input) {
  if (!rnn || !rnn->model || !gains || !strengths || !input) return;
  std::lock_guard<std::mutex> lk(g_rnn_mu);
  auto it = g_rnn_ctx.find(rnn->model);
  if (it == g_rnn_ctx.end()) {
    pn_model *m = pn_model_from_rnnmodel(rnn->model);
    pn_ctx *c = m ? pn_ctx_create(m, env_device(), 1, env_mode(), NULL) : NULL;
    if (!c) {                                              // not cached: the next call tries again
      fprintf(stderr, "percepnet_hip: compute_rnn: %s (no CPU fallback: outputs are zero)\n", pn_last_error());
      pn_model_free(m);
      memset(gains, 0, 34 * sizeof(float)); memset(strengths, 0, 34 * sizeof(float));
      return;
    }
    it = g_rnn_ctx.emplace(rnn->model, RnnEntry{m, c}).first;
  }
  pn_ctx *c = it->second.c;
  float gr[68];
  if (!c ||
      pn_ctx_set_rnn_state_host(c, rnn->first_conv1d_state, rnn->second_conv1d_state, rnn->gru1_state, rnn->gru2_state,
                                rnn->gru3_state, rnn->gb_gru_state, rnn->rb_gru_state) ||
      pn_ctx_compute_rnn_host(c, input, gr) ||
      pn_ctx_get_rnn_state_host(c, rnn->first_conv1d_state, rnn->second_conv1d_state, rnn->gru1_state, rnn->gru2_state,
                                rnn->gru3_state, rnn->gb_gru_state, rnn->rb_gru_state)) {
    memset(gains, 0, 34 * sizeof(float)); memset(strengths, 0, 34 * sizeof(float));
    return;
  }
  memcpy(gains, gr, 34 * sizeof(float)); memcpy(strengths, gr + 34, 34 * sizeof(float));
}

// rnnoise_model_from_file / rnnoise_model_free are declared by the reference (rnnoise.h:62-64) but defined nowhere; here
// they read / free a PNW1 container materialised as nnet_data.h records.  One parser: the container is validated by
// pn_model_from_file (fixed topology checked before any size is used), the records point into that model's storage.
// rnnoise_model_free only frees what rnnoise_model_from_file returned (a registry, not a guess about the memory in front
// of the pointer): a foreign RNNModel — the link-time percepnet_model_orig, a caller's own struct — is left alone.
struct OwnedModel { RNNModel m; DenseLayer d[3]; Conv1DLayer c[2]; GRULayer g[5]; pn_model *pm; };
static std::mutex g_owned_mu;
static std::map<const RNNModel *, OwnedModel *> g_owned;
static pn_model *owned_model_of(const RNNModel *m) {
  std::lock_guard<std::mutex> lk(g_owned_mu);
  auto it = g_owned.find(m);
  return it == g_owned.end() ? NULL : it->second->pm;
}

PN_REF_EXPORT RNNModel *rnnoise_model_from_file(FILE *f) {
  pn_model *pm = pn_model_from_file(f);
  if (!pm) return NULL;
  OwnedModel *o = (OwnedModel *)calloc(1, sizeof(OwnedModel));
  if (!o) { pn_model_free(pm); return NULL; }
  o->pm = pm;
  const PnLayerHost *L = pm->L;
  const int di[3] = {PN_L_FC, PN_L_FC_GB, PN_L_FC_RB};
  for (int i = 0; i < 3; i++) { const PnLayerHost &H = L[di[i]]; o->d[i] = {H.bias, H.w, H.nin, H.nn, H.act}; }
  for (int i = 0; i < 2; i++) { const PnLayerHost &H = L[PN_L_CONV1 + i]; o->c[i] = {H.bias, H.w, H.nin, H.ks, H.nn, H.act}; }
  for (int i = 0; i < 5; i++) { const PnLayerHost &H = L[PN_L_GRU1 + i]; o->g[i] = {H.bias, H.w, H.rw, H.nin, H.nn, H.act, H.reset_after}; }
  o->m = {&o->d[0], &o->c[0], &o->c[1], &o->g[0], &o->g[1], &o->g[2], &o->g[3], &o->g[4], &o->d[1], &o->d[2]};
  std::lock_guard<std::mutex> lk(g_owned_mu);
  g_owned[&o->m] = o;
  return &o->m;
}

PN_REF_EXPORT void rnnoise_model_free(RNNModel *model) {
  if (!model) return;
  OwnedModel *o = NULL;
  {
    std::lock_guard<std::mutex> lk(g_owned_mu);
    auto it = g_owned.find(model);
    if (it == g_owned.end()) return;                       // not ours: nothing to free, nothing to corrupt
    o = it->second;
    g_owned.erase(it);
  }
  { std::lock_guard<std::mutex> lk(g_rnn_mu); rnn_cache_drop(model); }   // a later model at this address starts clean
  pn_model_free(o->pm);
  free(o);
}

extern "C" {
int rnnoise_get_size_c(void) { return rnnoise_get_size(); }
int rnnoise_init_c(DenoiseState *st, RNNModel *model) { return rnnoise_init(st, model); }
DenoiseState *rnnoise_create_c(RNNModel *model) { return rnnoise_create(model); }
void rnnoise_destroy_c(DenoiseState *st) { rnnoise_destroy(st); }
float rnnoise_process_frame_c(DenoiseState *st, float *out, const float *in, FILE *f) { return rnnoise_process_frame(st, out, in, f); }
RNNModel *rnnoise_model_from_file_c(FILE *f) { return rnnoise_model_from_file(f); }
void rnnoise_model_free_c(RNNModel *m) { rnnoise_model_free(m); }
void rnnoise_compute_rnn_c(RNNState *rnn, float *gains, float *strengths, const float *input) { compute_rnn(rnn, gains, strengths, input); }
}

// ---- train(), rnnoise.h:66 / denoise.cpp:603-787: the body of the `percepNet` binary -----------------
// Same argv contract (<speech> <noisy> <count> <output>) and the same by-products in the cwd
// (test_output.pcm, test_input.pcm); one job = a batch of one pair — for dataset-scale runs use
// pn_featgen_run_files / the percepnet_featgen CLI with many jobs per call.
PN_REF_EXPORT int train(int argc, char **argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s <speech> <noisy> <count> <output>\n", argv[0]);
    return 1;
  }
  int dev = 0;
  if (const char *d = getenv("PERCEPNET_DEVICE")) dev = atoi(d);
  const char *sp = argv[1], *no = argv[2], *out = argv[4], *to = "test_output.pcm", *ti = "test_input.pcm";
  const int count = atoi(argv[3]);
  if (pn_featgen_run_files(dev, 1, &sp, &no, &count, &out, &to, &ti)) {
    fprintf(stderr, "percepnet_hip train: %s\n", pn_last_error());
    return 1;
  }
  return 0;
}
extern "C" int rnnoise_train_c(int argc, char **argv) { return train(argc, argv); }
