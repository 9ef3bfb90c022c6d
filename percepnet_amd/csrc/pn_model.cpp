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
  // content hash (64-bit multiply-xorshift over 8-byte words; the arrays are 4-byte floats, n_floats is even or the last
  // word is taken alone): two models with the same weights and activations share one packed copy per device and mode
  uint64_t h = 0x9e3779b97f4a7c15ull;
  auto mix = [&](uint64_t v) { h ^= v; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; };
  for (size_t i = 0; i + 1 < total; i += 2) { uint64_t v; memcpy(&v, m->storage + i, 8); mix(v); }
  if (total & 1) { uint32_t v; memcpy(&v, m->storage + total - 1, 4); mix(v); }
  for (int li = 0; li < PN_NLAYERS; li++) mix(((uint64_t)(uint32_t)m->L[li].act << 32) | (uint32_t)m->L[li].reset_after);
  m->content_hash = h;
  uint64_t f = 0xcbf29ce484222325ull;                  // FNV-1a, byte-wise: shares no structure with the word-wise mix above
  const unsigned char *bytes = (const unsigned char *)m->storage;
  for (size_t i = 0; i < total * 4; i++) { f ^= bytes[i]; f *= 0x100000001b3ull; }
  m->content_hash2 = f;
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

extern "C" void pn_model_free(pn_model *m) { if (m) { free(m->storage); free(m); } }

