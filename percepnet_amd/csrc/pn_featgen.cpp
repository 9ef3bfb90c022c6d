// Host side of the batched training-feature generator: the reference's
// `percepNet <speech> <noisy> <count> <output>` binary (train(), denoise.cpp:603-787) for n_pairs
// (speech, noisy) file pairs advanced in lock-step on one GPU.  Per frame: the inference path's
// front-end kernel once over the speech streams and once over the noisy streams, the target kernel,
// and — when the caller wants the TEST build's test_output.pcm — the inference back-end kernel on the
// noisy spectrum with the ideal gains, through the speech state's synthesis memory (744-757).
#include "pn_launch.h"
#include "../../include/percepnet_hip.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

struct FgSide {                 // one DenoiseState's worth of DSP state per stream (no network state)
  float *hist; float2 *yring; float *eyring; float2 *Ps; float *feat; int *silence; int *last_period;
  float *last_gain; float *aux;
};

struct pn_featgen {
  int device, B; int64_t t; size_t bytes;
  hipStream_t stream; bool own_stream;
  PnTables *tables;
  FgSide clean, noisy;
  float *synth;                 // st->synthesis_mem of the speech state (frame_synthesis(st, ...), 753)
  float *gr, *tmp_out;          // [B][68], [B][480] for the TEST synthesis
  std::vector<void *> allocs;
};

static int fg_alloc(pn_featgen *c, void **p, size_t bytes) {
  PN_HIP_CHECK(hipMalloc(p, bytes));
  c->allocs.push_back(*p); c->bytes += bytes;
  PN_HIP_CHECK(hipMemsetAsync(*p, 0, bytes, c->stream));
  return 0;
}
#define FG_ALLOC(ptr, count) \
  do { if (fg_alloc(c, (void **)&(ptr), sizeof(*(ptr)) * (size_t)(count))) goto fail; } while (0)

static int fg_zero_side(pn_featgen *c, FgSide &s) {
  const size_t B = c->B;
  PN_HIP_CHECK(hipMemsetAsync(s.hist, 0, B * PN_HIST_STRIDE * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(s.yring, 0, 6 * B * PN_SPEC_BINS * sizeof(float2), c->stream));
  PN_HIP_CHECK(hipMemsetAsync(s.eyring, 0, 6 * B * 36 * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(s.last_gain, 0, B * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(s.last_period, 0, B * 4, c->stream));
  PN_HIP_CHECK(hipMemsetAsync(s.silence, 0, B * 4, c->stream));
  return 0;
}

extern "C" void pn_featgen_destroy(pn_featgen *c) {
  if (!c) return;
  DeviceGuard _dg(c->device);
  hipStreamSynchronize(c->stream);
  for (void *p : c->allocs) hipFree(p);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}

extern "C" pn_featgen *pn_featgen_create(int device, int n_pairs, void *hip_stream) {
  if (n_pairs < 1) { pn_set_error("n_pairs must be >= 1"); return NULL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    pn_set_error("no HIP device available (this library has no CPU fallback)");
    return NULL;
  }
  if (device < 0 || device >= ndev) { pn_set_error("device %d out of range (%d devices)", device, ndev); return NULL; }
  DeviceGuard _dg(device);
  if (!_dg.ok) { pn_set_error("hipSetDevice(%d) failed", device); return NULL; }
  pn_featgen *c = new pn_featgen();
  c->device = device; c->B = n_pairs; c->t = 0; c->bytes = 0;
  if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { pn_set_error("hipStreamCreate failed"); delete c; return NULL; }
    c->own_stream = true;
  }
  {
    const size_t B = n_pairs;
    PnTables *ht = (PnTables *)malloc(sizeof(PnTables));
    if (!ht) { pn_set_error("out of host memory"); goto fail; }
    if (pn_build_tables(ht) || fg_alloc(c, (void **)&c->tables, sizeof(PnTables))) { free(ht); goto fail; }
    hipError_t e = hipMemcpyAsync(c->tables, ht, sizeof(PnTables), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    free(ht);
    if (e != hipSuccess) { pn_set_error("table upload failed: %s", hipGetErrorString(e)); goto fail; }
    FgSide *sides[2] = {&c->clean, &c->noisy};
    for (FgSide *s : sides) {
      FG_ALLOC(s->hist, B * PN_HIST_STRIDE);
      FG_ALLOC(s->yring, 6 * B * PN_SPEC_BINS);
      FG_ALLOC(s->eyring, 6 * B * 36);
      FG_ALLOC(s->Ps, B * PN_SPEC_BINS);
      FG_ALLOC(s->feat, B * PN_FEAT_STRIDE);
      FG_ALLOC(s->silence, B);
      FG_ALLOC(s->last_period, B);
      FG_ALLOC(s->last_gain, B);
      FG_ALLOC(s->aux, B * PN_AUX_STRIDE);
    }
    FG_ALLOC(c->synth, B * PN_FRAME);
    FG_ALLOC(c->gr, B * 68);
    FG_ALLOC(c->tmp_out, B * PN_FRAME);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { pn_set_error("featgen init failed"); goto fail; }
  }
  return c;
fail:
  pn_featgen_destroy(c);
  return NULL;
}

extern "C" int pn_featgen_reset(pn_featgen *c) {
  if (!c) return -1;
  PN_ON_DEVICE(c);
  if (fg_zero_side(c, c->clean) || fg_zero_side(c, c->noisy)) return -1;
  PN_HIP_CHECK(hipMemsetAsync(c->synth, 0, (size_t)c->B * PN_FRAME * 4, c->stream));
  c->t = 0;
  return 0;
}
extern "C" int pn_featgen_n_pairs(const pn_featgen *c) { return c ? c->B : -1; }
extern "C" int64_t pn_featgen_frames_done(const pn_featgen *c) { return c ? c->t : -1; }
extern "C" size_t pn_featgen_device_bytes(const pn_featgen *c) { return c ? c->bytes : 0; }
extern "C" int pn_featgen_synchronize(pn_featgen *c) { if (!c) return -1; PN_HIP_CHECK(hipStreamSynchronize(c->stream)); return 0; }

static int fg_frame(pn_featgen *c, const int16_t *sp, const int16_t *no, long long in_stride, float *rec,
                    long long rec_stride, int16_t *pcm, long long pcm_stride) {
  const int slot_w = (int)(c->t % 6), slot_r = (int)((c->t + 1) % 6);
  const size_t B = c->B;
  // train() analyses the noisy frame first (730) and the speech frame second (731); the two states are
  // independent, so the order of the launches is immaterial
  // the phase-split front end (three launches per analysed signal) unless PERCEPNET_FE=mono asks for the single-launch kernel
  static const bool mono = getenv("PERCEPNET_FE") && (!strcmp(getenv("PERCEPNET_FE"), "mono") || !strcmp(getenv("PERCEPNET_FE"), "g4"));
  auto fe = mono ? pn_launch_frontend : pn_launch_frontend_split;
  fe(c->stream, c->tables, c->B, c->t, no, 1, in_stride, 1.f, c->noisy.hist, c->noisy.yring, c->noisy.eyring, c->noisy.Ps,
     c->noisy.feat, c->noisy.silence, c->noisy.last_period, c->noisy.last_gain, c->noisy.aux, 0);
  fe(c->stream, c->tables, c->B, c->t, sp, 1, in_stride, 1.f, c->clean.hist, c->clean.yring, c->clean.eyring, c->clean.Ps,
     c->clean.feat, c->clean.silence, c->clean.last_period, c->clean.last_gain, c->clean.aux, 0);
  pn_launch_targets(c->stream, c->tables, c->B, c->clean.eyring + (size_t)slot_r * B * 36,
                    c->noisy.eyring + (size_t)slot_r * B * 36, c->noisy.eyring + (size_t)slot_w * B * 36, c->clean.aux,
                    c->noisy.aux, c->noisy.last_period, rec, rec_stride, c->gr);
  // the synthesis memory must advance every frame whether or not the caller keeps the PCM (753)
  pn_launch_backend(c->stream, c->tables, c->B, c->noisy.yring + (size_t)slot_r * B * PN_SPEC_BINS, c->noisy.Ps, c->gr,
                    nullptr /* the targets kernel already post-filtered g (743) */, c->noisy.silence, c->synth, c->tmp_out, 0, 0);
  if (pcm) pn_launch_saturate_i16(c->stream, c->B, c->tmp_out, pcm, pcm_stride);
  PN_HIP_CHECK(hipGetLastError());
  c->t++;
  return 0;
}

extern "C" int pn_featgen_process_i16(pn_featgen *c, const int16_t *d_speech, const int16_t *d_noisy, float *d_records,
                                      int16_t *d_test_pcm) {
  if (!c || !d_speech || !d_noisy || !d_records) { pn_set_error("NULL argument"); return -1; }
  PN_ON_DEVICE(c);
  return fg_frame(c, d_speech, d_noisy, PN_FRAME, d_records, 138, d_test_pcm, PN_FRAME);
}

extern "C" int pn_featgen_process_i16_files(pn_featgen *c, const int16_t *d_speech, const int16_t *d_noisy,
                                            int n_frames, float *d_records, int16_t *d_test_pcm) {
  if (!c || !d_speech || !d_noisy || !d_records || n_frames < 0) { pn_set_error("bad argument"); return -1; }
  PN_ON_DEVICE(c);
  const long long in_stride = (long long)n_frames * PN_FRAME;
  for (int f = 0; f < n_frames; f++)
    if (fg_frame(c, d_speech + (size_t)f * PN_FRAME, d_noisy + (size_t)f * PN_FRAME, in_stride,
                 d_records + (size_t)f * 138, (long long)n_frames * 138,
                 d_test_pcm ? d_test_pcm + (size_t)f * PN_FRAME : NULL, in_stride))
      return -1;
  return 0;
}

extern "C" int pn_featgen_process_host_i16_files(pn_featgen *c, const int16_t *h_speech, const int16_t *h_noisy,
                                                 int n_frames, float *h_records, int16_t *h_test_pcm) {
  if (!c || !h_speech || !h_noisy || !h_records || n_frames < 0) { pn_set_error("bad argument"); return -1; }
  PN_ON_DEVICE(c);
  const size_t n_in = (size_t)c->B * n_frames * PN_FRAME, n_rec = (size_t)c->B * n_frames * 138;
  int16_t *d_sp = NULL, *d_no = NULL, *d_pcm = NULL; float *d_rec = NULL;
  int rc = -1;
  do {
    if (hipMalloc((void **)&d_sp, n_in * 2) != hipSuccess || hipMalloc((void **)&d_no, n_in * 2) != hipSuccess ||
        hipMalloc((void **)&d_rec, n_rec * 4) != hipSuccess ||
        (h_test_pcm && hipMalloc((void **)&d_pcm, n_in * 2) != hipSuccess)) { pn_set_error("hipMalloc failed (featgen staging)"); break; }
    if (hipMemcpyAsync(d_sp, h_speech, n_in * 2, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(d_no, h_noisy, n_in * 2, hipMemcpyHostToDevice, c->stream) != hipSuccess) { pn_set_error("H2D failed"); break; }
    if (pn_featgen_process_i16_files(c, d_sp, d_no, n_frames, d_rec, d_pcm)) break;
    if (hipMemcpyAsync(h_records, d_rec, n_rec * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        (h_test_pcm && hipMemcpyAsync(h_test_pcm, d_pcm, n_in * 2, hipMemcpyDeviceToHost, c->stream) != hipSuccess)) { pn_set_error("D2H failed"); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { pn_set_error("stream sync failed: %s", hipGetErrorString(hipGetLastError())); break; }
    rc = 0;
  } while (0);
  hipStreamSynchronize(c->stream);
  hipFree(d_sp); hipFree(d_no); hipFree(d_rec); hipFree(d_pcm);
  return rc;
}

// ---- file-level driver: the `percepNet` binary for n_jobs (speech, noisy, count, output) jobs --------
// File semantics of train() (693-715): frames are read 480 shorts at a time; when a read hits EOF the
// file is rewound and the read repeated, i.e. the whole frames of a file are cycled and a partial tail
// is never used.  Jobs are advanced in lock-step to the largest count; each output receives its own
// count records.  test_out_paths / test_in_paths (arrays or NULL, entries may be NULL) receive what the
// TEST build writes to ./test_output.pcm and ./test_input.pcm.
static bool fg_read_pcm(const char *path, std::vector<int16_t> &v) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  v.resize(n > 0 ? (size_t)n / 2 : 0);
  const size_t got = v.empty() ? 0 : fread(v.data(), 2, v.size(), f);
  fclose(f);
  v.resize(got / PN_FRAME * PN_FRAME);          // whole frames only
  return true;
}

extern "C" int pn_featgen_run_files(int device, int n_jobs, const char *const *speech_paths,
                                    const char *const *noisy_paths, const int *counts, const char *const *out_paths,
                                    const char *const *test_out_paths, const char *const *test_in_paths) {
  if (n_jobs < 1 || !speech_paths || !noisy_paths || !counts || !out_paths) { pn_set_error("bad argument"); return -1; }
  std::vector<std::vector<int16_t>> sp(n_jobs), no(n_jobs);
  std::vector<FILE *> fo(n_jobs, NULL), fto(n_jobs, NULL), fti(n_jobs, NULL);
  int rc = -1, max_count = 0;
  pn_featgen *fg = NULL;
  auto close_all = [&]() {
    for (int j = 0; j < n_jobs; j++) { if (fo[j]) fclose(fo[j]); if (fto[j]) fclose(fto[j]); if (fti[j]) fclose(fti[j]); }
  };
  for (int j = 0; j < n_jobs; j++) {
    if (!fg_read_pcm(speech_paths[j], sp[j]) || !fg_read_pcm(noisy_paths[j], no[j])) {
      pn_set_error("cannot read %s / %s", speech_paths[j], noisy_paths[j]); close_all(); return -1; }
    if (counts[j] > 0 && (sp[j].size() < PN_FRAME || no[j].size() < PN_FRAME)) {
      pn_set_error("job %d: inputs must hold at least one 480-sample frame", j); close_all(); return -1; }
    fo[j] = fopen(out_paths[j], "wb");
    if (!fo[j]) { pn_set_error("cannot create %s", out_paths[j]); close_all(); return -1; }
    if (test_out_paths && test_out_paths[j]) fto[j] = fopen(test_out_paths[j], "wb");
    if (test_in_paths && test_in_paths[j]) fti[j] = fopen(test_in_paths[j], "wb");
    if (counts[j] > max_count) max_count = counts[j];
  }
  bool want_pcm = false;
  for (int j = 0; j < n_jobs; j++) want_pcm = want_pcm || fto[j];
  if (max_count > 0) {
    fg = pn_featgen_create(device, n_jobs, NULL);
    if (!fg) { close_all(); return -1; }
    // chunk so that the staging buffers stay around 256 MB
    long long chunk = (256ll << 20) / ((long long)n_jobs * PN_FRAME * 2);
    if (chunk < 1) chunk = 1;
    if (chunk > 1024) chunk = 1024;
    std::vector<int16_t> hs, hn, hp; std::vector<float> hr;
    for (int f0 = 0; f0 < max_count; f0 += (int)chunk) {
      const int F = max_count - f0 < chunk ? max_count - f0 : (int)chunk;
      hs.assign((size_t)n_jobs * F * PN_FRAME, 0); hn.assign(hs.size(), 0); hr.resize((size_t)n_jobs * F * 138);
      if (want_pcm) hp.resize(hs.size());
      for (int j = 0; j < n_jobs; j++) {
        if (sp[j].empty() || no[j].empty()) continue;
        const size_t ns = sp[j].size() / PN_FRAME, nn = no[j].size() / PN_FRAME;
        for (int f = 0; f < F; f++) {
          memcpy(&hs[((size_t)j * F + f) * PN_FRAME], &sp[j][((size_t)(f0 + f) % ns) * PN_FRAME], PN_FRAME * 2);
          memcpy(&hn[((size_t)j * F + f) * PN_FRAME], &no[j][((size_t)(f0 + f) % nn) * PN_FRAME], PN_FRAME * 2);
        }
      }
      if (pn_featgen_process_host_i16_files(fg, hs.data(), hn.data(), F, hr.data(), want_pcm ? hp.data() : NULL)) goto done;
      for (int j = 0; j < n_jobs; j++) {
        const int keep = counts[j] - f0 < F ? (counts[j] - f0 > 0 ? counts[j] - f0 : 0) : F;
        if (!keep) continue;
        fwrite(&hr[(size_t)j * F * 138], sizeof(float), (size_t)keep * 138, fo[j]);
        if (fto[j]) fwrite(&hp[(size_t)j * F * PN_FRAME], 2, (size_t)keep * PN_FRAME, fto[j]);
        if (fti[j]) fwrite(&hn[(size_t)j * F * PN_FRAME], 2, (size_t)keep * PN_FRAME, fti[j]);   // 722-728: the noisy frame, saturated = itself
      }
    }
  }
  rc = 0;
done:
  pn_featgen_destroy(fg);
  close_all();
  return rc;
}
