// Models: the error channel, the fixed PercepNet topology, and the three ways a weight set enters the library (an in-memory
// RNNModel in nnet_data.h layout, a PNW1 container, a FILE*).  No HIP in this file: it is also built alone, with
// -fsanitize=address,undefined, by the CPU test suite (tests/c/host_sanitize.cpp).
#include "pn_common.h"
#include "../../include/percepnet_hip.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// ---- errors -----------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void pn_set_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char *pn_last_error(void) { return g_err; }
extern "C" const char *pn_version(void) { return "percepnet_hip 0.2 (gfx950)"; }

// ---- models -------------------------------------------------------------------------------------------
const PnGeom pn_kGeom[PN_NLAYERS] = {
  {PN_KIND_DENSE, 70, 128, 1}, {PN_KIND_CONV1D, 128, 512, 5}, {PN_KIND_CONV1D, 512, 512, 3},
  {PN_KIND_GRU, 512, 512, 1}, {PN_KIND_GRU, 512, 512, 1}, {PN_KIND_GRU, 512, 512, 1}, {PN_KIND_GRU, 512, 512, 1},
  {PN_KIND_GRU, 1024, 128, 1}, {PN_KIND_DENSE, 2560, 34, 1}, {PN_KIND_DENSE, 128, 34, 1}};

size_t pn_layer_floats(int kind, int nin, int nn, int ks, size_t *nb, size_t *nw, size_t *nr) {
  *nb = kind == PN_KIND_GRU ? 6 * (size_t)nn : (size_t)nn;
  *nw = (size_t)nin * ks * nn * (kind == PN_KIND_GRU ? 3 : 1);
  *nr = kind == PN_KIND_GRU ? (size_t)nn * 3 * nn : 0;
  return *nb + *nw + *nr;
}

static int check_geometry(int li, int kind, int nin, int nn, int ks) {
  if (kind != pn_kGeom[li].kind || nin != pn_kGeom[li].nin || nn != pn_kGeom[li].nn || ks != pn_kGeom[li].ks) {
    pn_set_error("layer %d: geometry %d/%d/%d/%d differs from the PercepNet topology (rnn.cpp:42-81 hard-codes it)",
                 li, kind, nin, nn, ks);
    return -1;
  }
  return 0;
}



// ---- SHA-256 (FIPS 180-4) of a model's content: the key of the per-device cache of packed weights (pn_context.cpp).  A strong
// digest, computed ONCE per model, instead of two 64-bit hashes plus a retained 32 MB host copy that every cache hit was compared
// against byte for byte under the build lock (advisor, round 5): a PNW1 file cannot be crafted to share another model's entry.
namespace {
struct Sha256 {
  uint32_t h[8]; uint64_t n; unsigned char buf[64]; size_t fill;
  Sha256() : n(0), fill(0) {
    static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h, iv, sizeof(h));
  }
  static uint32_t ror(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
  void block(const unsigned char *p) {
    static const uint32_t K[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
      0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
      0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
      0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
      0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      const uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      const uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  void update(const void *data, size_t len) {
    const unsigned char *p = (const unsigned char *)data;
    n += len;
    if (fill) {
      const size_t take = len < 64 - fill ? len : 64 - fill;
      memcpy(buf + fill, p, take); fill += take; p += take; len -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
    for (; len >= 64; p += 64, len -= 64) block(p);
    if (len) { memcpy(buf, p, len); fill = len; }
  }
  void finish(unsigned char out[32]) {
    const uint64_t bits = n * 8;
    const unsigned char one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    unsigned char lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = (unsigned char)(bits >> (56 - 8 * i));
    update(lenb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (unsigned char)(h[i] >> 24); out[4 * i + 1] = (unsigned char)(h[i] >> 16); out[4 * i + 2] = (unsigned char)(h[i] >> 8); out[4 * i + 3] = (unsigned char)h[i]; }
  }
};
}  // namespace
extern "C" void pn_debug_sha256(const void *data, size_t len, unsigned char out[32]) { Sha256 s; s.update(data, len); s.finish(out); }

pn_model *pn_model_from_sources(const PnLayerSrc *src) {
  size_t total = 0;
  for (int li = 0; li < PN_NLAYERS; li++) {
    size_t nb, nw, nr;
    if (check_geometry(li, src[li].kind, src[li].nin, src[li].nn, src[li].ks)) return NULL;
    if (src[li].kind == PN_KIND_GRU && !src[li].reset_after) { pn_set_error("only reset_after GRUs are supported (dump_percepnet.py:94-98)"); return NULL; }
    total += pn_layer_floats(src[li].kind, src[li].nin, src[li].nn, src[li].ks, &nb, &nw, &nr);
  }
  pn_model *m = (pn_model *)calloc(1, sizeof(pn_model));
  if (m) m->storage = (float *)malloc(total * sizeof(float));
  if (!m || !m->storage) { free(m); pn_set_error("out of host memory for the model (%zu bytes)", total * sizeof(float)); return NULL; }
  m->n_floats = total;
  float *p = m->storage;
  for (int li = 0; li < PN_NLAYERS; li++) {
    size_t nb, nw, nr;
    pn_layer_floats(src[li].kind, src[li].nin, src[li].nn, src[li].ks, &nb, &nw, &nr);
    PnLayerHost &L = m->L[li];
    L.kind = src[li].kind; L.nin = src[li].nin; L.nn = src[li].nn; L.ks = src[li].ks; L.act = src[li].act;
    L.reset_after = src[li].reset_after;
    memcpy(p, src[li].bias, nb * 4); L.bias = p; p += nb;
    memcpy(p, src[li].w, nw * 4); L.w = p; p += nw;
    if (nr) { memcpy(p, src[li].rw, nr * 4); L.rw = p; p += nr; } else L.rw = NULL;
  }
  // content digest: every array byte in storage order, then every layer's (activation, reset_after) — what the packed device
  // copy depends on besides the fixed topology
  {
    Sha256 sh;
    sh.update(m->storage, total * sizeof(float));
    for (int li = 0; li < PN_NLAYERS; li++) { const int32_t d[2] = {m->L[li].act, m->L[li].reset_after}; sh.update(d, sizeof(d)); }
    sh.finish(m->sha256);
  }
  return m;
}

extern "C" pn_model *pn_model_from_rnnmodel(const RNNModel *r) {
  if (!r) { pn_set_error("NULL RNNModel"); return NULL; }
  PnLayerSrc s[PN_NLAYERS];
  const DenseLayer *d[3] = {r->fc, r->fc_gb, r->fc_rb};
  const int di[3] = {PN_L_FC, PN_L_FC_GB, PN_L_FC_RB};
  for (int i = 0; i < 3; i++) s[di[i]] = {PN_KIND_DENSE, d[i]->nb_inputs, d[i]->nb_neurons, 1, d[i]->activation, 0, d[i]->bias, d[i]->input_weights, NULL};
  const Conv1DLayer *c[2] = {r->conv1, r->conv2};
  for (int i = 0; i < 2; i++) s[PN_L_CONV1 + i] = {PN_KIND_CONV1D, c[i]->nb_inputs, c[i]->nb_neurons, c[i]->kernel_size, c[i]->activation, 0, c[i]->bias, c[i]->input_weights, NULL};
  const GRULayer *g[5] = {r->gru1, r->gru2, r->gru3, r->gru_gb, r->gru_rb};
  for (int i = 0; i < 5; i++) s[PN_L_GRU1 + i] = {PN_KIND_GRU, g[i]->nb_inputs, g[i]->nb_neurons, 1, g[i]->activation, g[i]->reset_after, g[i]->bias, g[i]->input_weights, g[i]->recurrent_weights};
  return pn_model_from_sources(s);
}

// PNW1 container (percepnet_amd/weights.py): "PNW1" | u32 n_layers | n_layers x { u32 kind, nin, nn, ks, act, reset_after |
// f32 bias[] | f32 weights[] | f32 recurrent[] }.  The topology is fixed (rnn.cpp:42-81 hard-codes it), so every header
// is checked against kGeom BEFORE any size is derived from it: no arithmetic on untrusted dimensions, nothing to overflow.
extern "C" pn_model *pn_model_from_blob(const void *blob, size_t nbytes) {
  const unsigned char *p = (const unsigned char *)blob;
  if (!p || nbytes < 8 || memcmp(p, "PNW1", 4) != 0) { pn_set_error("not a PNW1 weight container"); return NULL; }
  uint32_t n; memcpy(&n, p + 4, 4);
  if (n != PN_NLAYERS) { pn_set_error("PNW1: %u layers, expected %d", n, PN_NLAYERS); return NULL; }
  // arrays inside the blob are only 4-byte aligned relative to its start; copy through an aligned staging buffer
  std::vector<float> stage;
  try { stage.resize(nbytes / 4 + 1); } catch (...) { pn_set_error("PNW1: out of host memory (%zu bytes)", nbytes); return NULL; }
  PnLayerSrc s[PN_NLAYERS];
  size_t off = 8, fo = 0;
  for (uint32_t li = 0; li < n; li++) {
    if (nbytes - off < 24) { pn_set_error("PNW1: truncated in the header of layer %u", li); return NULL; }
    uint32_t h[6]; memcpy(h, p + off, 24); off += 24;
    if (h[0] != (uint32_t)pn_kGeom[li].kind || h[1] != (uint32_t)pn_kGeom[li].nin || h[2] != (uint32_t)pn_kGeom[li].nn || h[3] != (uint32_t)pn_kGeom[li].ks) {
      pn_set_error("PNW1: layer %u is kind %u, %u inputs, %u neurons, kernel %u; the PercepNet topology has %d / %d / %d / %d there "
                   "(rnn.cpp:42-81 hard-codes it)", li, h[0], h[1], h[2], h[3], pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks);
      return NULL;
    }
    if (h[4] > 3) { pn_set_error("PNW1: layer %u has activation %u (0..3 = linear, sigmoid, tanh, relu)", li, h[4]); return NULL; }
    size_t nb, nw, nr;
    const size_t tot = pn_layer_floats(pn_kGeom[li].kind, pn_kGeom[li].nin, pn_kGeom[li].nn, pn_kGeom[li].ks, &nb, &nw, &nr);   // <= 4 M floats
    if ((nbytes - off) / 4 < tot) { pn_set_error("PNW1: truncated in the arrays of layer %u", li); return NULL; }
    memcpy(&stage[fo], p + off, tot * 4); off += tot * 4;
    s[li] = {(int)h[0], (int)h[1], (int)h[2], (int)h[3], (int)h[4], (int)h[5], &stage[fo], &stage[fo + nb], nr ? &stage[fo + nb + nw] : NULL};
    fo += tot;
  }
  if (off != nbytes) { pn_set_error("PNW1: %zu trailing bytes", nbytes - off); return NULL; }
  return pn_model_from_sources(s);
}

extern "C" pn_model *pn_model_from_file(FILE *f) {
  if (!f) { pn_set_error("NULL FILE"); return NULL; }
  // the largest valid container is ~32 MB; refuse to slurp an arbitrarily large stream before looking at it
  const size_t limit = (size_t)64 << 20;
  std::vector<unsigned char> buf;
  unsigned char tmp[1 << 16];
  size_t n;
  try {
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) {
      if (buf.size() + n > limit) { pn_set_error("PNW1: more than %zu bytes: not a PercepNet weight container", limit); return NULL; }
      buf.insert(buf.end(), tmp, tmp + n);
    }
  } catch (...) { pn_set_error("PNW1: out of host memory"); return NULL; }
  if (ferror(f)) { pn_set_error("PNW1: read error"); return NULL; }
  return pn_model_from_blob(buf.data(), buf.size());
}

extern "C" void pn_model_digest(const pn_model *m, unsigned char out[32]) { if (m && out) memcpy(out, m->sha256, 32); }
extern "C" void pn_model_free(pn_model *m) { if (m) { free(m->storage); free(m); } }

