// CPU-only sanitizer harness (tests/test_hardening.py builds it with g++ -fsanitize=address,undefined -DPN_NO_HIP): the
// HIP-free host pieces of libpercepnet_hip — the PNW1 / RNNModel parsers (pn_model.cpp), the table builder (pn_tables.cpp),
// the weight packers (pn_pack.cpp) and the CLI helpers (pn_cli_util.h) — driven with valid, truncated, oversized and
// corrupted inputs.  Any out-of-bounds access, overflow or leak-free violation aborts; the process prints "ok" and exits 0.
#include "../../percepnet_amd/csrc/pn_model.cpp"
#include "../../percepnet_amd/csrc/pn_pack.cpp"
#include "../../percepnet_amd/csrc/pn_tables.cpp"
#include "../../percepnet_amd/csrc/pn_cli_util.h"
#include <stdio.h>
#include <string>
#include <vector>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_sanitize: CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #c, pn_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: host_sanitize <model.pnw>\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  CHECK(f);
  std::vector<unsigned char> blob;
  unsigned char tmp[65536]; size_t n;
  while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) blob.insert(blob.end(), tmp, tmp + n);
  rewind(f);

  // tables
  { PnTables *t = new PnTables(); CHECK(pn_build_tables(t) == 0); CHECK(t->border[PN_NB - 1] == PN_SPEC_BINS); CHECK(t->tansig[200] > 0.999f); delete t; }

  // CLI helpers
  { std::vector<int> d;
    CHECK(pn_cli_parse_devices("0,1", 2, d) && d.size() == 2); CHECK(!pn_cli_parse_devices("1,,2", 4, d)); CHECK(!pn_cli_parse_devices("1,", 4, d));
    CHECK(!pn_cli_parse_devices("x", 4, d)); CHECK(!pn_cli_parse_devices("4", 4, d)); CHECK(pn_cli_parse_devices("all", 3, d) && d.size() == 3);
    CHECK(!pn_cli_parse_devices("all", 0, d)); CHECK(!pn_cli_parse_devices("", 4, d)); CHECK(!pn_cli_parse_devices(NULL, 4, d));
    int tot = 0; for (int r = 0; r < 7; r++) { int a, c; pn_cli_shard(100, 7, r, &a, &c); CHECK(a == tot); tot += c; } CHECK(tot == 100); }

  // the valid container, from memory and from the FILE*
  pn_model *m = pn_model_from_blob(blob.data(), blob.size());
  CHECK(m);
  { pn_model *m2 = pn_model_from_file(f); CHECK(m2 && m2->n_floats == m->n_floats); pn_model_free(m2); }
  fclose(f);
  CHECK(pn_model_from_file(NULL) == NULL); CHECK(pn_model_from_blob(NULL, 0) == NULL);

  // every truncation near the headers, coarse steps elsewhere; one trailing byte
  { size_t off = 8; std::vector<size_t> cuts;
    for (size_t c = 0; c < 64; c++) cuts.push_back(c);
    for (int li = 0; li < PN_NLAYERS; li++) {
      size_t nb, nw, nr; const size_t tot = pn_layer_floats(pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks, &nb, &nw, &nr);
      for (size_t c = off > 4 ? off - 4 : 0; c < off + 28; c++) cuts.push_back(c);
      off += 24 + 4 * tot;
      cuts.push_back(off - 1); cuts.push_back(off - 4);
    }
    CHECK(off == blob.size());
    for (size_t c : cuts) if (c < blob.size()) {
      std::vector<unsigned char> t(blob.begin(), blob.begin() + c);          // exact-size copy: a read past the end is caught
      CHECK(pn_model_from_blob(t.data(), t.size()) == NULL);
    }
    std::vector<unsigned char> t(blob); t.push_back(0);
    CHECK(pn_model_from_blob(t.data(), t.size()) == NULL); }

  // header fields: absurd dimensions must be refused before any size is derived from them
  { size_t off = 8;
    const uint32_t vals[] = {0u, 1u, 2u, 3u, 7u, 0x10000u, 0x7fffffffu, 0x80000000u, 0xffffffffu};
    for (int li = 0; li < PN_NLAYERS; li++) {
      size_t nb, nw, nr; const size_t tot = pn_layer_floats(pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks, &nb, &nw, &nr);
      for (int fld = 0; fld < 6; fld++)
        for (uint32_t v : vals) {
          std::vector<unsigned char> t(blob);
          uint32_t orig; memcpy(&orig, &t[off + 4 * fld], 4);
          memcpy(&t[off + 4 * fld], &v, 4);
          pn_model *x = pn_model_from_blob(t.data(), t.size());
          if (fld < 4) CHECK((x != NULL) == (v == orig));                    // kind / inputs / neurons / kernel size are the topology
          pn_model_free(x);
        }
      off += 24 + 4 * tot;
    } }
  // wrong magic / layer count
  { std::vector<unsigned char> t(blob); t[0] = 'X'; CHECK(pn_model_from_blob(t.data(), t.size()) == NULL);
    t = blob; uint32_t nl = 11; memcpy(&t[4], &nl, 4); CHECK(pn_model_from_blob(t.data(), t.size()) == NULL);
    nl = 0xffffffffu; memcpy(&t[4], &nl, 4); CHECK(pn_model_from_blob(t.data(), t.size()) == NULL); }
  // random byte flips in the first 64 KB (headers of fc and conv1 + arrays): never a crash
  { uint32_t x = 2463534242u;
    for (int it = 0; it < 300; it++) {
      std::vector<unsigned char> t(blob);
      for (int k = 0; k < 4; k++) { x = x * 1664525u + 1013904223u; t[(x >> 8) % 65536] ^= (unsigned char)(x >> 24); }
      pn_model_free(pn_model_from_blob(t.data(), t.size()));
    } }

  // RNNModel path: a correct record set, then one with a wrong geometry, then NULL
  { DenseLayer d[3]; Conv1DLayer c[2]; GRULayer g[5];
    const int di[3] = {PN_L_FC, PN_L_FC_GB, PN_L_FC_RB};
    for (int i = 0; i < 3; i++) { const PnLayerHost &H = m->L[di[i]]; d[i] = {H.bias, H.w, H.nin, H.nn, H.act}; }
    for (int i = 0; i < 2; i++) { const PnLayerHost &H = m->L[PN_L_CONV1 + i]; c[i] = {H.bias, H.w, H.nin, H.ks, H.nn, H.act}; }
    for (int i = 0; i < 5; i++) { const PnLayerHost &H = m->L[PN_L_GRU1 + i]; g[i] = {H.bias, H.w, H.rw, H.nin, H.nn, H.act, H.reset_after}; }
    RNNModel r = {&d[0], &c[0], &c[1], &g[0], &g[1], &g[2], &g[3], &g[4], &d[1], &d[2]};
    pn_model *x = pn_model_from_rnnmodel(&r); CHECK(x); CHECK(!memcmp(x->storage, m->storage, m->n_floats * 4)); pn_model_free(x);
    g[2].nb_neurons = 1 << 30; CHECK(pn_model_from_rnnmodel(&r) == NULL); g[2].nb_neurons = 512;
    g[1].reset_after = 0; CHECK(pn_model_from_rnnmodel(&r) == NULL);
    CHECK(pn_model_from_rnnmodel(NULL) == NULL); }

  // packers: exactly-sized destinations, every layer, both tile orders
  for (int li = 0; li < PN_NLAYERS; li++) {
    const PnLayerHost &H = m->L[li];
    const int K = H.nin * H.ks, ncols = H.nn * (H.kind == PN_KIND_GRU ? 3 : 1), k_alloc = li == PN_L_FC ? PN_FEAT_STRIDE : K;
    const int ctr = H.kind == PN_KIND_GRU ? 1 : pn_dense_nt(H.nn);
    std::vector<float> wp(pn_packed_floats(k_alloc, ncols, ctr));
    pn_pack_weights(H.w, K, k_alloc, ncols, ctr, wp.data());
    if (H.rw) { std::vector<float> rp(pn_packed_floats(H.nn, ncols, 1)); pn_pack_weights(H.rw, H.nn, H.nn, ncols, 1, rp.data()); }
    if (H.kind == PN_KIND_DENSE && ncols <= 48) { std::vector<float> q(pn_packed_floats_n16(K, ncols)); pn_pack_weights_n16(H.w, K, ncols, q.data()); }
  }
  pn_model_free(m);
  pn_model_free(NULL);
  puts("ok");
  return 0;
}
