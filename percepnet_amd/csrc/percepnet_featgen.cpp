// percepnet_featgen — the reference's `percepNet <speech> <noisy> <count> <output>` training-feature
// binary (src/main.cpp:13-16 -> train(), denoise.cpp:603-787) for many jobs at once on one or several GPUs:
//
//   percepnet_featgen [--device N | --devices 0,1,..|all] [--test-pcm] <speech> <noisy> <count> <output> [...more jobs]
//
// --devices: the jobs are cut into contiguous balanced shards (the rule of percepnet_run / sharding.py), one host
// thread + one generator per device, no communication — the counterpart of the reference's 8-way shell fan-out of
// its preparation loop (utils/run.sh:95-117).
//
// Each <output> receives count records of 138 float32, byte-compatible with the reference's output
// (consumer: rnn_train.py:44-53).  --test-pcm also writes <output>.test_output.pcm and
// <output>.test_input.pcm (what the reference drops into its cwd for its single job).
#include "../../include/percepnet_hip.h"
#include "pn_cli_util.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>

int main(int argc, char **argv) {
  int test_pcm = 0, ai = 1;
  std::vector<int> devices;
  for (; ai < argc; ai++) {
    if (!strcmp(argv[ai], "--device") && ai + 1 < argc) devices.assign(1, atoi(argv[++ai]));
    else if (!strcmp(argv[ai], "--devices") && ai + 1 < argc) {
      if (!pn_cli_parse_devices(argv[++ai], pn_device_count(), devices)) {
        fprintf(stderr, "--devices: expected a comma-separated list of device ordinals in [0,%d) or 'all', got '%s'\n", pn_device_count(), argv[ai]);
        return 1;
      }
    }
    else if (!strcmp(argv[ai], "--test-pcm")) test_pcm = 1;
    else break;
  }
  const int nargs = argc - ai;
  if (nargs < 4 || nargs % 4) {
    fprintf(stderr, "usage: %s [--device N | --devices 0,1,..|all] [--test-pcm] <speech> <noisy> <count> <output> [...more jobs]\n", argv[0]);
    return 1;
  }
  const int J = nargs / 4;
  std::vector<const char *> sp(J), no(J), out(J), to(J, nullptr), ti(J, nullptr);
  std::vector<int> counts(J);
  std::vector<std::string> names(2 * J);
  for (int j = 0; j < J; j++) {
    sp[j] = argv[ai + 4 * j]; no[j] = argv[ai + 4 * j + 1]; counts[j] = atoi(argv[ai + 4 * j + 2]); out[j] = argv[ai + 4 * j + 3];
    if (test_pcm) {
      names[2 * j] = std::string(out[j]) + ".test_output.pcm"; names[2 * j + 1] = std::string(out[j]) + ".test_input.pcm";
      to[j] = names[2 * j].c_str(); ti[j] = names[2 * j + 1].c_str();
    }
  }
  if (devices.empty()) devices.push_back(0);
  const int W = (int)devices.size() < J ? (int)devices.size() : J;
  struct Shard { int device, first, count, rc; std::string err; };
  std::vector<Shard> shards(W);
  auto run = [&](Shard *sh) {
    sh->rc = pn_featgen_run_files(sh->device, sh->count, sp.data() + sh->first, no.data() + sh->first, counts.data() + sh->first,
                                  out.data() + sh->first, to.data() + sh->first, ti.data() + sh->first);
    if (sh->rc) sh->err = pn_last_error();          // thread-local: read it on the thread that failed
  };
  for (int r = 0; r < W; r++) { shards[r].device = devices[r]; shards[r].rc = 0; pn_cli_shard(J, W, r, &shards[r].first, &shards[r].count); }
  if (W == 1) run(&shards[0]);
  else {
    std::vector<std::thread> th;
    for (int r = 0; r < W; r++) th.emplace_back(run, &shards[r]);
    for (auto &t : th) t.join();
  }
  int rc = 0;
  for (const Shard &sh : shards)
    if (sh.rc) { fprintf(stderr, "percepnet_featgen: device %d (jobs %d..%d): %s\n", sh.device, sh.first, sh.first + sh.count - 1, sh.err.c_str()); rc = 2; }
  if (rc && W > 1)
    for (const Shard &sh : shards)
      fprintf(stderr, "  outputs of jobs %d..%d (device %d): %s\n", sh.first, sh.first + sh.count - 1, sh.device, sh.rc ? "INCOMPLETE - discard" : "complete");
  return rc;
}
