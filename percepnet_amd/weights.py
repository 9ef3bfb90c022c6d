"""Weight ingestion for the PercepNet gain network, in the reference's ``nnet_data.h`` layout.

The reference ships no pretrained model (README todo; ``src/nnet_data.cpp`` is absent), so the
benchmark/parity weights are the ones BASELINE.md names: a ``torch.manual_seed(seed)``
default-initialised network with the topology of ``rnn_train.py:105-121``, laid out exactly as
``dump_percepnet.py`` lays it out in C:

* Dense   ``input_weights[in*N + out]``            = ``weight.T``               (dump_percepnet.py:62)
* Conv1d  ``input_weights[(k*Cin + c)*N + out]``   = ``weight.permute(2,1,0)``  (dump_percepnet.py:113)
* GRU     ``input_weights[in*3N + gate*N + out]``, gates re-ordered torch (r,z,n) -> C (z,r,h)
          (dump_percepnet.py:68-76); ``bias[6N] = cat(b_ih,b_hh).reshape(2,3,N)[:,[1,0,2]]`` (78-80)

``printVector`` (dump_percepnet.py:32-49) prints each float32 through ``'{}'.format`` which
yields the exact (double) decimal expansion of the float32, so the C compiler reads back the
identical float32: the dump is value-preserving and no text round trip is needed here
(checked against the reference's own dumper in tests/golden/make_golden.py).

The binary container ("PNW1") is this repo's answer to the declared-but-never-defined
``rnnoise_model_from_file`` (rnnoise.h:62): ten layer records in ``RNNModel`` order
(nnet_data.h:6-26).
"""
import hashlib
import os
import struct

import numpy as np

# RNNModel order (nnet_data.h:6-26) with (kind, nb_inputs, nb_neurons, kernel_size, activation)
ACT_LINEAR, ACT_SIGMOID, ACT_TANH, ACT_RELU = 0, 1, 2, 3
KIND_DENSE, KIND_CONV1D, KIND_GRU = 0, 1, 2
LAYERS = [
    ("fc", KIND_DENSE, 70, 128, 1, ACT_RELU),
    ("conv1", KIND_CONV1D, 128, 512, 5, ACT_RELU),
    ("conv2", KIND_CONV1D, 512, 512, 3, ACT_TANH),
    ("gru1", KIND_GRU, 512, 512, 1, ACT_TANH),
    ("gru2", KIND_GRU, 512, 512, 1, ACT_TANH),
    ("gru3", KIND_GRU, 512, 512, 1, ACT_TANH),
    ("gru_gb", KIND_GRU, 512, 512, 1, ACT_TANH),
    ("gru_rb", KIND_GRU, 1024, 128, 1, ACT_TANH),
    ("fc_gb", KIND_DENSE, 2560, 34, 1, ACT_SIGMOID),
    ("fc_rb", KIND_DENSE, 128, 34, 1, ACT_SIGMOID),
]
MAGIC = b"PNW1"
N_PARAMS = 7962564


def _gru_kernel_to_c(kernel):
    """torch [3N, in] (r,z,n) -> C [in, 3N] (z,r,h)."""
    k_r, k_z, k_h = np.vsplit(kernel, 3)
    return np.ascontiguousarray(np.hstack([k_z.T, k_r.T, k_h.T]))


def _gru_bias_to_c(b_ih, b_hh):
    b = np.concatenate([b_ih, b_hh]).reshape(2, 3, -1)
    return np.ascontiguousarray(b[:, [1, 0, 2], :].reshape(-1))


def build_torch_modules(seed=1234):
    """Same construction order as rnn_train.PercepNet.__init__ so the RNG stream matches."""
    import torch
    from torch import nn

    torch.manual_seed(seed)
    mods = {}
    mods["fc"] = nn.Linear(70, 128)
    mods["conv1"] = nn.Conv1d(128, 512, 5, stride=1, padding=4)
    mods["conv2"] = nn.Conv1d(512, 512, 3, stride=1, padding=2)
    mods["gru1"] = nn.GRU(512, 512, 1, batch_first=True)
    mods["gru2"] = nn.GRU(512, 512, 1, batch_first=True)
    mods["gru3"] = nn.GRU(512, 512, 1, batch_first=True)
    mods["gru_gb"] = nn.GRU(512, 512, 1, batch_first=True)
    mods["gru_rb"] = nn.GRU(1024, 128, 1, batch_first=True)
    mods["fc_gb"] = nn.Linear(512 * 5, 34)
    mods["fc_rb"] = nn.Linear(128, 34)
    return mods


def modules_to_layers(mods, gain=1.0):
    """-> dict name -> dict(bias, input_weights[, recurrent_weights]) as float32 C-layout arrays.

    ``gain`` scales the *weights* (not biases) after layout; gain != 1 gives a "saturating"
    weight set that drives the tanh/sigmoid table into its clamp (|x| > 8)."""
    out = {}
    for name, kind, nin, nn_, ks, act in LAYERS:
        m = mods[name]
        if kind == KIND_DENSE:
            w = m.weight.detach().numpy().T
            rec = None
            b = m.bias.detach().numpy()
        elif kind == KIND_CONV1D:
            w = m.weight.detach().permute(2, 1, 0).numpy()
            rec = None
            b = m.bias.detach().numpy()
        else:
            w = _gru_kernel_to_c(m.weight_ih_l0.detach().numpy())
            rec = _gru_kernel_to_c(m.weight_hh_l0.detach().numpy())
            b = _gru_bias_to_c(m.bias_ih_l0.detach().numpy(), m.bias_hh_l0.detach().numpy())
        d = {
            "bias": np.ascontiguousarray(b, dtype=np.float32).reshape(-1),
            "input_weights": (np.ascontiguousarray(w, dtype=np.float32).reshape(-1) * np.float32(gain)),
        }
        if rec is not None:
            d["recurrent_weights"] = np.ascontiguousarray(rec, dtype=np.float32).reshape(-1) * np.float32(gain)
        out[name] = d
    return out


def random_layers(seed, scale=None):
    """Torch-free weights (numpy RNG), uniform(-1/sqrt(fan), 1/sqrt(fan)) like torch's default."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, kind, nin, nn_, ks, act in LAYERS:
        if kind == KIND_GRU:
            bound = 1.0 / np.sqrt(nn_)
            d = {
                "bias": rng.uniform(-bound, bound, 6 * nn_),
                "input_weights": rng.uniform(-bound, bound, nin * 3 * nn_),
                "recurrent_weights": rng.uniform(-bound, bound, nn_ * 3 * nn_),
            }
        else:
            bound = 1.0 / np.sqrt(nin * ks)
            d = {
                "bias": rng.uniform(-bound, bound, nn_),
                "input_weights": rng.uniform(-bound, bound, nin * ks * nn_),
            }
        if scale is not None:
            for k in d:
                if k != "bias":
                    d[k] = d[k] * scale
        out[name] = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in d.items()}
    return out


def pack_blob(layers, acts=None):
    """Serialise to the PNW1 container: magic, u32 n_layers, then per layer
    ``u32 kind, nb_inputs, nb_neurons, kernel_size, activation, reset_after`` followed by the
    float32 arrays bias, input_weights[, recurrent_weights].  acts: optional {layer name: activation code}
    overriding the PercepNet topology's activations (the known-answer tests embed the reference's toy layers)."""
    parts = [MAGIC, struct.pack("<I", len(LAYERS))]
    for name, kind, nin, nn_, ks, act in LAYERS:
        d = layers[name]
        if acts and name in acts:
            act = acts[name]
        parts.append(struct.pack("<6I", kind, nin, nn_, ks, act, 1 if kind == KIND_GRU else 0))
        nb = 6 * nn_ if kind == KIND_GRU else nn_
        assert d["bias"].size == nb, name
        assert d["input_weights"].size == nin * ks * nn_ * (3 if kind == KIND_GRU else 1), name
        parts.append(d["bias"].astype("<f4").tobytes())
        parts.append(d["input_weights"].astype("<f4").tobytes())
        if kind == KIND_GRU:
            assert d["recurrent_weights"].size == nn_ * 3 * nn_, name
            parts.append(d["recurrent_weights"].astype("<f4").tobytes())
    return b"".join(parts)


def unpack_blob(blob):
    assert blob[:4] == MAGIC
    (n,) = struct.unpack_from("<I", blob, 4)
    off = 8
    out = {}
    for li in range(n):
        kind, nin, nn_, ks, act, ra = struct.unpack_from("<6I", blob, off)
        off += 24
        name = LAYERS[li][0]
        sizes = [("bias", 6 * nn_ if kind == KIND_GRU else nn_),
                 ("input_weights", nin * ks * nn_ * (3 if kind == KIND_GRU else 1))]
        if kind == KIND_GRU:
            sizes.append(("recurrent_weights", nn_ * 3 * nn_))
        d = {}
        for key, sz in sizes:
            d[key] = np.frombuffer(blob, dtype="<f4", count=sz, offset=off).copy()
            off += 4 * sz
        out[name] = d
    assert off == len(blob)
    return out


def layer_digest(layers):
    """sha256 per array, the unit compared with tests/golden/weights_seed1234.json."""
    dg = {}
    for name, *_ in LAYERS:
        for k, v in layers[name].items():
            dg[f"{name}.{k}"] = hashlib.sha256(v.astype("<f4").tobytes()).hexdigest()
    return dg


_CACHE = {}


def default_blob(seed=1234, gain=1.0, cache_dir=None):
    """The BASELINE.md weight set (seeded default-init PercepNet), cached on disk."""
    key = (seed, gain)
    if key in _CACHE:
        return _CACHE[key]
    cache_dir = cache_dir or os.environ.get("PERCEPNET_CACHE", "/tmp/percepnet_amd_cache")
    os.makedirs(cache_dir, exist_ok=True)
    path = os.path.join(cache_dir, f"pnw1_seed{seed}_gain{gain:g}.bin")
    if os.path.exists(path):
        with open(path, "rb") as f:
            blob = f.read()
    else:
        blob = pack_blob(modules_to_layers(build_torch_modules(seed), gain))
        tmp = path + f".{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
    _CACHE[key] = blob
    return blob


# ---- exporter: trained checkpoints / generated nnet_data.cpp -> PNW1 ------------------------------------------------
# The reference's route from a trained model to the C engine is dump_percepnet.py:128-155: load the bare state_dict that
# rnn_train.py:245,322 saves, print every array as C text into src/nnet_data.cpp, recompile.  Here both ends of that
# route are accepted as inputs and the result is the binary container the engine loads at run time
# (rnnoise_model_from_file / pn_model_from_file / percepnet_run --model):
#     python -m percepnet_amd.weights --checkpoint model.pt -o model.pnw
#     python -m percepnet_amd.weights --nnet-data src/nnet_data.cpp -o model.pnw
_SD_PREFIX = {KIND_DENSE: "{}.0.", KIND_CONV1D: "{}.0.", KIND_GRU: "{}."}   # Sequential(Linear|Conv1d, act) vs bare GRU


def layers_from_state_dict(sd):
    """A rnn_train.PercepNet state_dict (name -> tensor / ndarray; a {"model": sd} / {"state_dict": sd} wrapper and a
    DataParallel "module." prefix are unwrapped) -> layers in nnet_data.h layout, by the dumper's own transforms
    (dump_percepnet.py:51-126)."""
    for key in ("model", "state_dict", "model_state_dict"):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}

    def arr(name):
        if name not in sd:
            raise KeyError(f"checkpoint has no '{name}' (not a rnn_train.PercepNet state_dict?)")
        v = sd[name]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        return np.asarray(v, dtype=np.float32)

    out = {}
    for name, kind, nin, nn_, ks, act in LAYERS:
        p = _SD_PREFIX[kind].format(name)
        if kind == KIND_DENSE:
            w, b, rec = arr(p + "weight").T, arr(p + "bias"), None
            want = (nin, nn_)
        elif kind == KIND_CONV1D:
            w, b, rec = arr(p + "weight").transpose(2, 1, 0), arr(p + "bias"), None
            want = (ks, nin, nn_)
        else:
            w = _gru_kernel_to_c(arr(p + "weight_ih_l0"))
            rec = _gru_kernel_to_c(arr(p + "weight_hh_l0"))
            b = _gru_bias_to_c(arr(p + "bias_ih_l0"), arr(p + "bias_hh_l0"))
            want = (nin, 3 * nn_)
        if tuple(w.shape) != want:
            raise ValueError(f"{name}: weight shape {tuple(w.shape)} is not the PercepNet topology's {want} (rnn_train.py:105-121)")
        d = {"bias": np.ascontiguousarray(b, dtype=np.float32).reshape(-1),
             "input_weights": np.ascontiguousarray(w, dtype=np.float32).reshape(-1)}
        if rec is not None:
            d["recurrent_weights"] = np.ascontiguousarray(rec, dtype=np.float32).reshape(-1)
        out[name] = d
    return out


_ACT_CODE = {"LINEAR": ACT_LINEAR, "SIGMOID": ACT_SIGMOID, "TANH": ACT_TANH, "RELU": ACT_RELU}


def layers_from_nnet_data(text):
    """The C text dump_percepnet.py writes (src/nnet_data.cpp) -> (layers, acts).  Every float literal is the exact
    decimal expansion of a float32 ('{}'.format of a numpy float32, dump_percepnet.py:38), so float(literal) cast to
    float32 is the identical value.  The layer records (`const DenseLayer fc = { fc_bias, fc_weights, 70, 128,
    ACTIVATION_RELU };` ...) are checked against the topology and supply the activations."""
    import re
    arrays = {}
    for m in re.finditer(r"static\s+const\s+float\s+(\w+)\s*\[\s*(\d+)\s*\]\s*=\s*\{(.*?)\}\s*;", text, re.S):
        vals = np.array([float(v) for v in m.group(3).replace("\n", " ").split(",") if v.strip()], dtype=np.float64)
        if vals.size != int(m.group(2)):
            raise ValueError(f"{m.group(1)}: declared {m.group(2)} values, found {vals.size}")
        arrays[m.group(1)] = vals.astype(np.float32)
    recs = {m.group(2): (m.group(1), [t.strip() for t in m.group(3).split(",") if t.strip()])
            for m in re.finditer(r"const\s+(DenseLayer|Conv1DLayer|GRULayer)\s+(\w+)\s*=\s*\{(.*?)\}\s*;", text, re.S)}
    out, acts = {}, {}
    ctype = {KIND_DENSE: "DenseLayer", KIND_CONV1D: "Conv1DLayer", KIND_GRU: "GRULayer"}
    for name, kind, nin, nn_, ks, act in LAYERS:
        if name not in recs or recs[name][0] != ctype[kind]:
            raise ValueError(f"nnet_data text has no `const {ctype[kind]} {name}` record")
        tok = recs[name][1]
        if kind == KIND_DENSE:       # bias, weights, nb_inputs, nb_neurons, activation            (nnet.h:44-50)
            b, w, rw, dims, a = tok[0], tok[1], None, (int(tok[2]), 1, int(tok[3])), tok[4]
        elif kind == KIND_CONV1D:    # bias, weights, nb_inputs, kernel_size, nb_neurons, activation (nnet.h:75-82)
            b, w, rw, dims, a = tok[0], tok[1], None, (int(tok[2]), int(tok[3]), int(tok[4])), tok[5]
        else:                        # bias, weights, recurrent, nb_inputs, nb_neurons, activation, reset_after (nnet.h:62-70)
            b, w, rw, dims, a = tok[0], tok[1], tok[2], (int(tok[3]), 1, int(tok[4])), tok[5]
            if int(tok[6]) != 1:
                raise ValueError(f"{name}: reset_after = {tok[6]}; only reset_after GRUs exist in this engine (dump_percepnet.py:94-98)")
        if dims != (nin, ks, nn_):
            raise ValueError(f"{name}: geometry {dims} is not the PercepNet topology's {(nin, ks, nn_)}")
        acts[name] = _ACT_CODE[a.replace("ACTIVATION_", "")]
        d = {"bias": arrays[b], "input_weights": arrays[w]}
        if rw:
            d["recurrent_weights"] = arrays[rw]
        out[name] = d
    return out, acts


def export_main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m percepnet_amd.weights",
                                 description="Convert a trained PercepNet (rnn_train.py checkpoint or the nnet_data.cpp text "
                                             "dump_percepnet.py writes) into the PNW1 container libpercepnet_hip loads.")
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--checkpoint", help="torch.save()d state_dict of rnn_train.PercepNet (rnn_train.py:245,322)")
    src.add_argument("--nnet-data", help="C text written by dump_percepnet.py (src/nnet_data.cpp)")
    src.add_argument("--seed", type=int, help="the seeded default-init network of BASELINE.md (no trained model exists upstream)")
    ap.add_argument("-o", "--output", required=True)
    a = ap.parse_args(argv)
    acts = None
    if a.checkpoint:
        import torch
        try:
            sd = torch.load(a.checkpoint, map_location="cpu", weights_only=True)
        except TypeError:
            sd = torch.load(a.checkpoint, map_location="cpu")
        layers = layers_from_state_dict(sd)
    elif a.nnet_data:
        layers, acts = layers_from_nnet_data(open(a.nnet_data).read())
    else:
        layers = modules_to_layers(build_torch_modules(a.seed))
    blob = pack_blob(layers, acts)
    with open(a.output, "wb") as f:
        f.write(blob)
    n = sum(v.size for d in layers.values() for v in d.values())
    print(f"{a.output}: PNW1, {len(LAYERS)} layers, {n} parameters, {len(blob)} bytes, "
          f"sha256 {hashlib.sha256(blob).hexdigest()[:16]}")
    return 0


if __name__ == "__main__":
    raise SystemExit(export_main())
