// Shared by the two command-line front ends (percepnet_run, percepnet_featgen): device-list parsing and the shard rule.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <vector>

// "0,1,3" | "all" -> device ordinals.  Every element must be a non-empty decimal number in [0, n_visible):
// "1,,2", "1," and "x" are rejected (false), never read as device 0.
static inline bool pn_cli_parse_devices(const char *s, int n_visible, std::vector<int> &out) {
  out.clear();
  if (!s || !*s) return false;
  if (!strcmp(s, "all")) { for (int d = 0; d < n_visible; d++) out.push_back(d); return !out.empty(); }
  for (const char *p = s;;) {
    char *end = NULL;
    const long v = strtol(p, &end, 10);
    if (end == p || v < 0 || v >= n_visible) { out.clear(); return false; }
    out.push_back((int)v);
    if (*end == '\0') return true;
    if (*end != ',') { out.clear(); return false; }
    p = end + 1;
  }
}

// Contiguous balanced shards, the same rule as percepnet_amd/sharding.py shard_streams(): shard r of w over n units.
static inline void pn_cli_shard(int n, int w, int r, int *first, int *count) {
  const int base = n / w, rem = n % w;
  *first = r * base + (r < rem ? r : rem);
  *count = base + (r < rem ? 1 : 0);
}
