// The reference's frame-engine interface on top of the batched HIP engine.
//
// The reference declares its API in src/rnnoise.h:49-68 WITHOUT extern "C" and builds it from
// .cpp files, so the symbols an existing caller (src/main.cpp:30-39) links against are the
// Itanium-mangled ones (_Z14rnnoise_createP8RNNModel, _Z21rnnoise_process_frameP12DenoiseState
// PfPKfP8_IO_FILE, ...).  This file defines functions with exactly those prototypes in the global
// C++ namespace — the compiler emits exactly those symbols — plus extern "C" `_c` spellings.
//
// One DenoiseState = a batch-of-one context.  That is correct but launch-bound (13 kernel
// launches and two PCIe hops per 10 ms frame); throughput lives in the batched pn_* API.
#include "../../include/percepnet_hip.h"
#include <stdlib.h>
#include <string.h>

// link-time model of the reference (denoise.cpp:49-51): generated nnet_data.cpp defines it.
// Weak: a caller may instead pass a model explicitly or set PERCEPNET_MODEL=<file.pnw>.
extern const RNNModel percepnet_model_orig __attribute__((weak));

struct DenoiseState {
  uint32_t magic;
  pn_ctx *ctx;
  pn_model *model;
  float gr[68];
};
#define DS_MAGIC 0x504e4453u

int rnnoise_get_size() { return (int)sizeof(DenoiseState); }

// rnnoise_init (denoise.cpp:259-280): zero state, bind the model.  Returns 0; on failure the
// state is left inert (process_frame then outputs silence) and pn_last_error() says why —
// the reference has no error path at all here.
int rnnoise_init(DenoiseState *st, RNNModel *model) {
  memset(st, 0, sizeof(*st));
  st->magic = DS_MAGIC;
  const RNNModel *m = model ? model : (&percepnet_model_orig ? &percepnet_model_orig : NULL);
  if (m) st->model = pn_model_from_rnnmodel(m);
  else if (const char *path = getenv("PERCEPNET_MODEL")) {
    FILE *f = fopen(path, "rb");
    if (f) { st->model = pn_model_from_file(f); fclose(f); }
  }
  if (!st->model) return 0;
  int dev = 0;
  if (const char *d = getenv("PERCEPNET_DEVICE")) dev = atoi(d);
  int mode = PN_NN_MFMA;
  if (const char *s = getenv("PERCEPNET_STRICT")) mode = atoi(s) ? PN_NN_STRICT : PN_NN_MFMA;
  st->ctx = pn_ctx_create(st->model, dev, 1, mode, NULL);
  return 0;
}

DenoiseState *rnnoise_create(RNNModel *model) {
  DenoiseState *st = (DenoiseState *)malloc(rnnoise_get_size());
  if (st) rnnoise_init(st, model);
  return st;
}

void rnnoise_destroy(DenoiseState *st) {
  if (!st) return;
  if (st->magic == DS_MAGIC) { pn_ctx_destroy(st->ctx); pn_model_free(st->model); }
  free(st);
}

// rnnoise_process_frame (denoise.cpp:508-547): 480 floats in -> 480 floats out (may alias), the
// 68-float g|r tap appended to f_feature (the reference requires it non-NULL; NULL is accepted
// here).  Always returns 0 like the reference.
float rnnoise_process_frame(DenoiseState *st, float *out, const float *in, FILE *f_feature) {
  if (!st || st->magic != DS_MAGIC || !st->ctx) { if (out) memset(out, 0, PN_FRAME_SIZE * sizeof(float)); return 0; }
  float tmp[PN_FRAME_SIZE];
  memcpy(tmp, in, sizeof(tmp));
  pn_process_host_f32(st->ctx, tmp, out, st->gr);
  if (f_feature) { fwrite(st->gr, sizeof(float), 34, f_feature); fwrite(st->gr + 34, sizeof(float), 34, f_feature); }
  return 0;
}

// rnnoise_model_from_file / rnnoise_model_free are declared by the reference (rnnoise.h:62-64)
// but defined nowhere; here they read/free a PNW1 container materialised as nnet_data.h records.
struct OwnedModel { RNNModel m; DenseLayer d[3]; Conv1DLayer c[2]; GRULayer g[5]; float *data; };

RNNModel *rnnoise_model_from_file(FILE *f) {
  if (!f) return NULL;
  size_t cap = 1 << 20, n = 0;
  unsigned char *buf = (unsigned char *)malloc(cap);
  for (;;) {
    size_t r = fread(buf + n, 1, cap - n, f);
    n += r;
    if (r == 0) break;
    if (n == cap) { cap *= 2; buf = (unsigned char *)realloc(buf, cap); }
  }
  if (n < 8 || memcmp(buf, "PNW1", 4) != 0) { free(buf); return NULL; }
  OwnedModel *o = (OwnedModel *)calloc(1, sizeof(OwnedModel));
  o->data = (float *)malloc(n);
  size_t off = 8, fo = 0; int nd = 0, nc = 0, ng = 0; bool ok = true;
  for (int li = 0; li < 10 && ok; li++) {
    uint32_t h[6];
    if (off + 24 > n) { ok = false; break; }
    memcpy(h, buf + off, 24); off += 24;
    const size_t nb = h[0] == 2 ? 6 * (size_t)h[2] : h[2];
    const size_t nw = (size_t)h[1] * h[3] * h[2] * (h[0] == 2 ? 3 : 1);
    const size_t nr = h[0] == 2 ? (size_t)h[2] * 3 * h[2] : 0;
    if (off + 4 * (nb + nw + nr) > n) { ok = false; break; }
    memcpy(o->data + fo, buf + off, 4 * (nb + nw + nr)); off += 4 * (nb + nw + nr);
    const float *b = o->data + fo, *w = b + nb, *rw = w + nw; fo += nb + nw + nr;
    if (h[0] == 0 && nd < 3) o->d[nd++] = {b, w, (int)h[1], (int)h[2], (int)h[4]};
    else if (h[0] == 1 && nc < 2) o->c[nc++] = {b, w, (int)h[1], (int)h[3], (int)h[2], (int)h[4]};
    else if (h[0] == 2 && ng < 5) o->g[ng++] = {b, w, rw, (int)h[1], (int)h[2], (int)h[4], (int)h[5]};
    else ok = false;
  }
  free(buf);
  if (!ok || nd != 3 || nc != 2 || ng != 5) { free(o->data); free(o); return NULL; }
  o->m = {&o->d[0], &o->c[0], &o->c[1], &o->g[0], &o->g[1], &o->g[2], &o->g[3], &o->g[4], &o->d[1], &o->d[2]};
  return &o->m;   // RNNModel is the first member: the same address frees the whole record
}

void rnnoise_model_free(RNNModel *model) {
  if (!model) return;
  OwnedModel *o = (OwnedModel *)model;
  free(o->data); free(o);
}

extern "C" {
int rnnoise_get_size_c(void) { return rnnoise_get_size(); }
int rnnoise_init_c(DenoiseState *st, RNNModel *model) { return rnnoise_init(st, model); }
DenoiseState *rnnoise_create_c(RNNModel *model) { return rnnoise_create(model); }
void rnnoise_destroy_c(DenoiseState *st) { rnnoise_destroy(st); }
float rnnoise_process_frame_c(DenoiseState *st, float *out, const float *in, FILE *f) { return rnnoise_process_frame(st, out, in, f); }
RNNModel *rnnoise_model_from_file_c(FILE *f) { return rnnoise_model_from_file(f); }
void rnnoise_model_free_c(RNNModel *m) { rnnoise_model_free(m); }
}

// ---- train(), rnnoise.h:66 / denoise.cpp:603-787: the body of the `percepNet` binary -----------------
// Same argv contract (<speech> <noisy> <count> <output>) and the same by-products in the cwd
// (test_output.pcm, test_input.pcm); one job = a batch of one pair — for dataset-scale runs use
// pn_featgen_run_files / the percepnet_featgen CLI with many jobs per call.
int train(int argc, char **argv) {
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
