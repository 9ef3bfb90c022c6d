// percepnet_run — the reference's `percepNet_run <noisy.pcm> <out.pcm>` CLI (src/main.cpp:11-44) on
// top of libpercepnet_hip's batched C-ABI, extended to N file pairs processed as N concurrent
// streams on one or several GPUs.  Same I/O contract per stream: raw little-endian int16
// mono 48 kHz in; (frames-1)*480 samples out (first output frame dropped, main.cpp:37; partial tail frame
// dropped, main.cpp:32-33); with a single pair ./feature_test.raw gets 68 floats per frame.
//
//   percepnet_run [--model model.pnw] [--strict | --x3] [--postfilter] [--slots N] [--device N | --devices 0,1,..|all]
//                 [--no-numa] [--verbose]
//                 in0.pcm out0.pcm [in1.pcm out1.pcm ...]
//
// --slots N: at most N concurrent streams per device; further pairs wait and take over the slot of a pair that has ended
// (per-stream re-initialisation on the device, pn_ctx_reset_streams) — a directory of recordings of different lengths goes
// through a fixed-size context without padding the short ones with silence.
//
// Multi-GPU: streams are independent, so the pairs are cut into contiguous balanced shards, one per
// device; every device gets its own host thread, its own context (a replica of the weights and tables) and its own
// pinned buffers, and the threads never talk to each other — the one-process counterpart of the reference's shell
// fan-out (utils/run.sh:49,65,99).  No collective is involved.  Each device's thread binds itself to the CPUs of that GPU's NUMA
// node before it creates its context and pinned buffers (pn_bind_thread_to_device_numa; --no-numa leaves the affinity alone,
// --verbose prints the binding).
#include "../../include/percepnet_hip.h"
#include "pn_cli_util.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

extern const RNNModel percepnet_model_orig __attribute__((weak));

struct Shard { int device, first, count, rc; std::string err; };

// Everything a shard holds; released on every exit path of run_shard.
struct ShardRes {
  pn_ctx *cx = NULL;
  std::vector<FILE *> fin, fout;        // per SLOT: the pair currently playing there
  FILE *ftap = NULL;
  // one of three rotating pinned buffer sets; file[s] = the output file the frame of slot s belongs to (NULL: slot idle),
  // skip[s] = that frame is the pair's first output frame, which main.cpp:37 drops
  struct Slot { int16_t *in = NULL, *out = NULL; float *gr = NULL; std::vector<FILE *> file; std::vector<char> skip, last; } slot[3];
  ~ShardRes() {
    for (auto &sl : slot) { pn_host_free(sl.in); pn_host_free(sl.out); pn_host_free(sl.gr); }
    for (FILE *f : fin) if (f) fclose(f);
    for (FILE *f : fout) if (f) fclose(f);
    if (ftap) fclose(ftap);
    pn_ctx_destroy(cx);
  }
};

// One device: pairs [first, first+count) of argv-style (in, out) paths through `n_slots` concurrent streams (n_slots =
// count: every pair has its own stream from the start, the round-1 behaviour).  With fewer slots than pairs the shard is a
// queue: when the pair playing in a slot runs out of input, the slot is re-initialised on the device
// (pn_ctx_reset_streams = rnnoise_destroy + rnnoise_create of the reference, denoise.cpp:252-280,326-331) and the next
// waiting pair starts there on the following frame, while the other slots keep running.
static bool g_numa = true, g_verbose = false;
static void run_shard(Shard *sh, const pn_model *m, char **paths, int nn_mode, int postfilter, bool tap, int n_slots) {
  const int P = sh->count, B = n_slots > 0 && n_slots < P ? n_slots : P;
  auto fail = [&](int rc, const std::string &msg) { sh->rc = rc; sh->err = msg; };
  ShardRes R;
  // this thread owns the device from here on: run on the CPUs of the GPU's NUMA node BEFORE the context and the pinned
  // buffers exist (first touch places them), so that N threads feeding N GPUs do not all pull through one socket
  if (g_numa) {
    char msg[256];
    pn_bind_thread_to_device_numa(sh->device, msg, sizeof(msg));
    if (g_verbose) fprintf(stderr, "percepnet_run: %s\n", msg);
  }
  R.cx = pn_ctx_create(m, sh->device, B, nn_mode, NULL);
  pn_ctx *cx = R.cx;
  if (!cx) return fail(3, std::string("pn_ctx_create: ") + pn_last_error());
  if (postfilter) pn_ctx_set_postfilter(cx, 1);
  std::vector<FILE *> &fin = R.fin, &fout = R.fout;
  fin.assign(B, NULL); fout.assign(B, NULL);
  int next_pair = 0;
  auto open_pair = [&](int s) -> bool {                 // the next waiting pair starts playing in slot s
    const char *pi = paths[2 * (sh->first + next_pair)], *po = paths[2 * (sh->first + next_pair) + 1];
    next_pair++;
    fin[s] = fopen(pi, "rb"); fout[s] = fopen(po, "wb");
    if (!fin[s] || !fout[s]) { fail(4, std::string("cannot open ") + pi + " / " + po); return false; }
    return true;
  };
  for (int s = 0; s < B; s++) if (!open_pair(s)) return;
  R.ftap = tap ? fopen("feature_test.raw", "wb") : NULL;
  FILE *ftap = R.ftap;
  // Three rotating pinned buffer sets on the pipelined entry point: the files of frame t+1 are read while the GPU
  // works on frame t, and frame t-2's output is on the host once pn_submit_host_i16(t) has returned.
  typedef ShardRes::Slot Slot;
  Slot *slot = R.slot;
  for (int k = 0; k < 3; k++) {
    Slot &sl = slot[k];
    sl.in = (int16_t *)pn_host_alloc((size_t)B * PN_FRAME_SIZE * sizeof(int16_t));
    sl.out = (int16_t *)pn_host_alloc((size_t)B * PN_FRAME_SIZE * sizeof(int16_t));
    sl.gr = (float *)pn_host_alloc((size_t)B * 68 * sizeof(float));
    if (!sl.in || !sl.out || !sl.gr) return fail(5, pn_last_error());
    sl.file.assign(B, NULL); sl.skip.assign(B, 0); sl.last.assign(B, 0);
  }
  std::vector<char> first(B, 1);
  auto flush = [&](Slot &sl) {                       // main.cpp:36-38 for every stream that supplied this frame
    for (int s = 0; s < B; s++) {
      if (!sl.file[s]) continue;
      if (ftap) fwrite(&sl.gr[(size_t)s * 68], sizeof(float), 68, ftap);
      if (!sl.skip[s]) fwrite(&sl.out[(size_t)s * PN_FRAME_SIZE], sizeof(int16_t), PN_FRAME_SIZE, sl.file[s]);
      if (sl.last[s]) { fclose(sl.file[s]); }        // the pair's last frame has been written: its output file is complete
    }
  };
  std::vector<int32_t> restart;
  int n_alive = B;
  long t = 0;
  for (;; t++) {
    Slot &sl = slot[t % 3];
    restart.clear();
    for (int s = 0; s < B; s++) {
      int16_t *x = sl.in + (size_t)s * PN_FRAME_SIZE;
      sl.file[s] = NULL; sl.skip[s] = 0; sl.last[s] = 0;
      if (fin[s] && fread(x, sizeof(int16_t), PN_FRAME_SIZE, fin[s]) != PN_FRAME_SIZE) {
        // this pair is finished (partial tail dropped, main.cpp:32-33): mark the frame it supplied last as its final one
        fclose(fin[s]); fin[s] = NULL;
        Slot &prev = slot[(t + 2) % 3];                // = frame t - 1
        if (t >= 1 && prev.file[s] == fout[s]) prev.last[s] = 1; else if (fout[s]) fclose(fout[s]);
        fout[s] = NULL;
        while (next_pair < P) {                        // the slot starts over with the next waiting pair, from this frame on
          if (!open_pair(s)) return;
          if (fread(x, sizeof(int16_t), PN_FRAME_SIZE, fin[s]) == PN_FRAME_SIZE) { restart.push_back(s); first[s] = 1; break; }
          fclose(fin[s]); fin[s] = NULL; fclose(fout[s]); fout[s] = NULL;       // shorter than one frame: an empty output, next pair
        }
        if (!fin[s]) n_alive--;
      }
      if (fin[s]) { sl.file[s] = fout[s]; sl.skip[s] = first[s]; first[s] = 0; }
      else memset(x, 0, PN_FRAME_SIZE * sizeof(int16_t));
    }
    if (n_alive == 0) break;
    if (!restart.empty() && pn_ctx_reset_streams(cx, restart.data(), (int)restart.size())) return fail(5, pn_last_error());
    if (pn_submit_host_i16(cx, sl.in, sl.out, sl.gr)) return fail(5, pn_last_error());
    if (t >= 2) flush(slot[(t - 2) % 3]);
  }
  if (pn_host_wait(cx)) return fail(5, pn_last_error());
  for (long u = (t >= 2 ? t - 2 : 0); u < t; u++) flush(slot[u % 3]);     // the last two frames in flight
  for (int s = 0; s < B; s++) fout[s] = NULL;        // every output file was closed with its last frame
}

int main(int argc, char **argv) {
  const char *model_path = getenv("PERCEPNET_MODEL");
  int nn_mode = PN_NN_MFMA, postfilter = 0, ai = 1, n_slots = 0;
  std::vector<int> devices;
  for (; ai < argc; ai++) {
    if (!strcmp(argv[ai], "--model") && ai + 1 < argc) model_path = argv[++ai];
    else if (!strcmp(argv[ai], "--strict")) nn_mode = PN_NN_STRICT;      // reference-order network, bit-exact to the CPU path
    else if (!strcmp(argv[ai], "--x3")) nn_mode = PN_NN_MFMA_X3;         // split-precision network (same +-1 LSB bound, ~2x the rate)
    else if (!strcmp(argv[ai], "--postfilter")) postfilter = 1;      // optional envelope post-filter (denoise.cpp:216-250)
    else if (!strcmp(argv[ai], "--no-numa")) g_numa = false;         // leave the host threads' CPU affinity alone
    else if (!strcmp(argv[ai], "--verbose")) g_verbose = true;       // one line per device: its NUMA binding
    else if (!strcmp(argv[ai], "--slots") && ai + 1 < argc) n_slots = atoi(argv[++ai]);   // concurrent streams per device: pairs queue for them
    else if (!strcmp(argv[ai], "--device") && ai + 1 < argc) devices.assign(1, atoi(argv[++ai]));
    else if (!strcmp(argv[ai], "--devices") && ai + 1 < argc) {
      if (!pn_cli_parse_devices(argv[++ai], pn_device_count(), devices)) {
        fprintf(stderr, "--devices: expected a comma-separated list of device ordinals in [0,%d) or 'all', got '%s'\n", pn_device_count(), argv[ai]);
        return 1;
      }
    }
    else break;
  }
  if (devices.empty()) devices.push_back(0);
  const int nfiles = argc - ai;
  if (nfiles < 2 || (nfiles & 1)) {
    fprintf(stderr, "usage: %s [--model model.pnw] [--strict | --x3] [--postfilter] [--slots N] [--device N | --devices 0,1,..|all] <noisy speech> <output denoised> [...more pairs]\n", argv[0]);
    return 1;
  }
  const int B = nfiles / 2;
  pn_model *m = NULL;
  if (model_path) { FILE *f = fopen(model_path, "rb"); if (f) { m = pn_model_from_file(f); fclose(f); } }
  else if (&percepnet_model_orig) m = pn_model_from_rnnmodel(&percepnet_model_orig);
  if (!m) { fprintf(stderr, "no model: pass --model file.pnw (or link a generated nnet_data.cpp): %s\n", pn_last_error()); return 2; }
  // contiguous balanced shards (the same rule as percepnet_amd/sharding.py: shard_streams); devices beyond the number
  // of pairs stay idle
  const int W = (int)devices.size() < B ? (int)devices.size() : B;
  std::vector<Shard> shards(W);
  for (int r = 0; r < W; r++) {
    int first, count;
    pn_cli_shard(B, W, r, &first, &count);
    shards[r] = {devices[r], first, count, 0, ""};
  }
  const bool tap = B == 1;
  if (W == 1) run_shard(&shards[0], m, argv + ai, nn_mode, postfilter, tap, n_slots);
  else {
    std::vector<std::thread> th;
    for (int r = 0; r < W; r++) th.emplace_back(run_shard, &shards[r], m, argv + ai, nn_mode, postfilter, false, n_slots);
    for (auto &t : th) t.join();
  }
  int rc = 0;
  for (const Shard &sh : shards)
    if (sh.rc) { fprintf(stderr, "device %d (pairs %d..%d): %s\n", sh.device, sh.first, sh.first + sh.count - 1, sh.err.c_str()); if (sh.rc > rc) rc = sh.rc; }
  if (rc && W > 1)         // some shards may have finished: say which outputs are complete and which are not
    for (const Shard &sh : shards)
      fprintf(stderr, "  outputs of pairs %d..%d (device %d): %s\n", sh.first, sh.first + sh.count - 1, sh.device,
              sh.rc ? "INCOMPLETE - discard" : "complete");
  pn_model_free(m);
  return rc;
}
