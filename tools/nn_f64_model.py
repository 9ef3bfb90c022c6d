"""float64 evaluation of ONE compute_rnn step (rnn.cpp:42-81, nnet.cpp) from a given RNN state: the yardstick for "how far
is each network mode from exact arithmetic".  Same formulas as the reference — table-interpolated tanh/sigmoid
(vec.h:53-75, evaluated in double on the float table), reset-after GRU (nnet.cpp:122-180), conv as dense over the FIFO
(nnet.cpp:182-200) — with every product and sum in double, so what remains between a mode's output and this one is that
mode's rounding (operand representation + accumulation order), not the model.

  step(layers, table, state, feat) -> gr [B, 68]
      layers: percepnet_amd.weights.unpack_blob(...) ; table: the 201 float tanh table ; state: Context.get_rnn_state() ;
      feat [B, 70]
"""
import numpy as np


def _tansig(x, tab):
    x = np.asarray(x, np.float64)
    sign = np.where(x < 0, -1.0, 1.0)
    ax = np.abs(x)
    i = np.clip(np.floor(.5 + 25 * ax), 0, 200).astype(np.int64)
    ax = ax - .04 * i
    y = tab[i]
    dy = 1 - y * y
    y = y + ax * dy * (1 - y * ax)
    return sign * y


def _act(x, act, tab):
    if act == 1:
        return .5 + .5 * _tansig(.5 * x, tab)
    if act == 2:
        return _tansig(x, tab)
    if act == 3:
        return np.maximum(x, 0)
    return x


def _dense(lay, x, nn_, act, tab):
    W = lay["input_weights"].astype(np.float64).reshape(-1, nn_)
    return _act(lay["bias"].astype(np.float64) + x @ W, act, tab)


def _gru(lay, x, h, nn_, act, tab):
    W = lay["input_weights"].astype(np.float64).reshape(-1, 3 * nn_)
    U = lay["recurrent_weights"].astype(np.float64).reshape(nn_, 3 * nn_)
    b = lay["bias"].astype(np.float64)
    z = _act(b[0:nn_] + b[3 * nn_:4 * nn_] + x @ W[:, 0:nn_] + h @ U[:, 0:nn_], 1, tab)
    r = _act(b[nn_:2 * nn_] + b[4 * nn_:5 * nn_] + x @ W[:, nn_:2 * nn_] + h @ U[:, nn_:2 * nn_], 1, tab)
    tmp = b[5 * nn_:6 * nn_] + h @ U[:, 2 * nn_:3 * nn_]
    hc = _act(b[2 * nn_:3 * nn_] + tmp * r + x @ W[:, 2 * nn_:3 * nn_], act, tab)
    return z * h + (1 - z) * hc


def step(layers, table, state, feat, acts=None):
    from percepnet_amd import weights
    acts = acts or {name: act for name, kind, nin, nn_, ks, act in weights.LAYERS}
    tab = np.asarray(table, np.float64)[:201]
    f = np.asarray(feat, np.float64)
    st = {k: np.asarray(v, np.float64) for k, v in state.items()}
    fc = _dense(layers["fc"], f, 128, acts["fc"], tab)
    c1 = _dense(layers["conv1"], np.concatenate([st["conv1"], fc], axis=1), 512, acts["conv1"], tab)      # FIFO oldest first, then the current input
    c2 = _dense(layers["conv2"], np.concatenate([st["conv2"], c1], axis=1), 512, acts["conv2"], tab)
    g1 = _gru(layers["gru1"], c2, st["gru1"], 512, acts["gru1"], tab)
    g2 = _gru(layers["gru2"], g1, st["gru2"], 512, acts["gru2"], tab)
    g3 = _gru(layers["gru3"], g2, st["gru3"], 512, acts["gru3"], tab)
    gb = _gru(layers["gru_gb"], g3, st["gru_gb"], 512, acts["gru_gb"], tab)
    rb = _gru(layers["gru_rb"], np.concatenate([g3, c2], axis=1), st["gru_rb"], 128, acts["gru_rb"], tab)
    g = _dense(layers["fc_gb"], np.concatenate([c2, g1, g2, g3, gb], axis=1), 34, acts["fc_gb"], tab)
    r = _dense(layers["fc_rb"], rb, 34, acts["fc_rb"], tab)
    return np.concatenate([g, r], axis=1)
