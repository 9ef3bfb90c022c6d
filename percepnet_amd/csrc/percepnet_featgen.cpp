// percepnet_featgen — the reference's `percepNet <speech> <noisy> <count> <output>` training-feature
// binary (src/main.cpp:13-16 -> train(), denoise.cpp:603-787) for many jobs at once on one GPU:
//
//   percepnet_featgen [--device N] [--test-pcm] <speech> <noisy> <count> <output> [<speech> <noisy> <count> <output> ...]
//
// Each <output> receives count records of 138 float32, byte-compatible with the reference's output
// (consumer: rnn_train.py:44-53).  --test-pcm also writes <output>.test_output.pcm and
// <output>.test_input.pcm (what the reference drops into its cwd for its single job).
#include "../../include/percepnet_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

int main(int argc, char **argv) {
  int device = 0, test_pcm = 0, ai = 1;
  for (; ai < argc; ai++) {
    if (!strcmp(argv[ai], "--device") && ai + 1 < argc) device = atoi(argv[++ai]);
    else if (!strcmp(argv[ai], "--test-pcm")) test_pcm = 1;
    else break;
  }
  const int nargs = argc - ai;
  if (nargs < 4 || nargs % 4) {
    fprintf(stderr, "usage: %s [--device N] [--test-pcm] <speech> <noisy> <count> <output> [...more jobs]\n", argv[0]);
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
  if (pn_featgen_run_files(device, J, sp.data(), no.data(), counts.data(), out.data(), to.data(), ti.data())) {
    fprintf(stderr, "percepnet_featgen: %s\n", pn_last_error());
    return 2;
  }
  return 0;
}
