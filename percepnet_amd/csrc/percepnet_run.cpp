// percepnet_run — the reference's `percepNet_run <noisy.pcm> <out.pcm>` CLI (src/main.cpp:11-44) on
// top of libpercepnet_hip's batched C-ABI, extended to N file pairs processed as N concurrent
// streams.  Same I/O contract per stream: raw little-endian int16 mono 48 kHz
// in; (frames-1)*480 samples out (first output frame dropped, main.cpp:37; partial tail frame
// dropped, main.cpp:32-33); with a single pair ./feature_test.raw gets 68 floats per frame.
//
//   percepnet_run [--model model.pnw] [--strict] [--device N] in0.pcm out0.pcm [in1.pcm out1.pcm ...]
#include "../../include/percepnet_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

extern const RNNModel percepnet_model_orig __attribute__((weak));

int main(int argc, char **argv) {
  const char *model_path = getenv("PERCEPNET_MODEL");
  int strict = 0, device = 0, postfilter = 0, ai = 1;
  for (; ai < argc; ai++) {
    if (!strcmp(argv[ai], "--model") && ai + 1 < argc) model_path = argv[++ai];
    else if (!strcmp(argv[ai], "--strict")) strict = 1;
    else if (!strcmp(argv[ai], "--postfilter")) postfilter = 1;      // optional envelope post-filter (denoise.cpp:216-250)
    else if (!strcmp(argv[ai], "--device") && ai + 1 < argc) device = atoi(argv[++ai]);
    else break;
  }
  const int nfiles = argc - ai;
  if (nfiles < 2 || (nfiles & 1)) {
    fprintf(stderr, "usage: %s [--model model.pnw] [--strict] [--postfilter] [--device N] <noisy speech> <output denoised> [...more pairs]\n", argv[0]);
    return 1;
  }
  const int B = nfiles / 2;
  pn_model *m = NULL;
  if (model_path) { FILE *f = fopen(model_path, "rb"); if (f) { m = pn_model_from_file(f); fclose(f); } }
  else if (&percepnet_model_orig) m = pn_model_from_rnnmodel(&percepnet_model_orig);
  if (!m) { fprintf(stderr, "no model: pass --model file.pnw (or link a generated nnet_data.cpp): %s\n", pn_last_error()); return 2; }
  pn_ctx *cx = pn_ctx_create(m, device, B, strict ? PN_NN_STRICT : PN_NN_MFMA, NULL);
  if (!cx) { fprintf(stderr, "pn_ctx_create: %s\n", pn_last_error()); return 3; }
  if (postfilter) pn_ctx_set_postfilter(cx, 1);
  std::vector<FILE *> fin(B), fout(B);
  for (int s = 0; s < B; s++) {
    fin[s] = fopen(argv[ai + 2 * s], "rb"); fout[s] = fopen(argv[ai + 2 * s + 1], "wb");
    if (!fin[s] || !fout[s]) { fprintf(stderr, "cannot open %s / %s\n", argv[ai + 2 * s], argv[ai + 2 * s + 1]); return 4; }
  }
  FILE *ftap = (B == 1) ? fopen("feature_test.raw", "wb") : NULL;
  // Three rotating pinned buffer sets on the pipelined entry point: the files of frame t+1 are read while the GPU
  // works on frame t, and frame t-2's output is on the host once pn_submit_host_i16(t) has returned.
  struct Slot { int16_t *in, *out; float *gr; std::vector<char> alive; };
  Slot slot[3];
  for (Slot &sl : slot) {
    sl.in = (int16_t *)pn_host_alloc((size_t)B * PN_FRAME_SIZE * sizeof(int16_t));
    sl.out = (int16_t *)pn_host_alloc((size_t)B * PN_FRAME_SIZE * sizeof(int16_t));
    sl.gr = (float *)pn_host_alloc((size_t)B * 68 * sizeof(float));
    if (!sl.in || !sl.out || !sl.gr) { fprintf(stderr, "%s\n", pn_last_error()); return 5; }
    sl.alive.assign(B, 0);
  }
  std::vector<char> alive(B, 1), first(B, 1);
  auto flush = [&](const Slot &sl) {                 // main.cpp:36-38 for every stream that supplied this frame
    for (int s = 0; s < B; s++) {
      if (!sl.alive[s]) continue;
      if (ftap) fwrite(&sl.gr[(size_t)s * 68], sizeof(float), 68, ftap);
      if (!first[s]) fwrite(&sl.out[(size_t)s * PN_FRAME_SIZE], sizeof(int16_t), PN_FRAME_SIZE, fout[s]);
      first[s] = 0;
    }
  };
  int n_alive = B;
  long t = 0;
  for (;; t++) {
    Slot &sl = slot[t % 3];
    for (int s = 0; s < B; s++) {
      int16_t *x = sl.in + (size_t)s * PN_FRAME_SIZE;
      if (alive[s] && fread(x, sizeof(int16_t), PN_FRAME_SIZE, fin[s]) != PN_FRAME_SIZE) { alive[s] = 0; n_alive--; }
      if (!alive[s]) memset(x, 0, PN_FRAME_SIZE * sizeof(int16_t));
    }
    if (n_alive == 0) break;
    sl.alive = alive;
    if (pn_submit_host_i16(cx, sl.in, sl.out, sl.gr)) { fprintf(stderr, "%s\n", pn_last_error()); return 5; }
    if (t >= 2) flush(slot[(t - 2) % 3]);
  }
  if (pn_host_wait(cx)) { fprintf(stderr, "%s\n", pn_last_error()); return 5; }
  for (long u = (t >= 2 ? t - 2 : 0); u < t; u++) flush(slot[u % 3]);     // the last two frames in flight
  for (Slot &sl : slot) { pn_host_free(sl.in); pn_host_free(sl.out); pn_host_free(sl.gr); }
  for (int s = 0; s < B; s++) { fclose(fin[s]); fclose(fout[s]); }
  if (ftap) fclose(ftap);
  pn_ctx_destroy(cx); pn_model_free(m);
  return 0;
}
