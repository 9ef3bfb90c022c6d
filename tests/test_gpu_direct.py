"""Direct-operand fp32 GRU kernels (percepnet_amd/csrc/pn_nn_d.hip) and fc_gb on 16x16x4 batch tiles (pn_nn_n48.hip); -m gpu.

From 24 576 streams the GRU steps of nn_mode NN_MFMA evaluate the same k-ascending fmaf chains as the batch-GEMM and small-batch
families (compute_gru, reference src/nnet.cpp:120-180; order of sgemv_accum, src/vec.h:102-135) with the activations taken from
fragment-order fp32 shadows instead of through LDS (conv2's batch kernel and the GRU steps write them), so every output must be
BIT-IDENTICAL to the batch family's — on ragged batches, with 32 and 64 rows per wave, as row-range chains, and through every
call that touches the network state (per-stream reset, the active set, state load / store): the shadows are a second copy of that
state and must follow it."""
import numpy as np
import pytest

from percepnet_amd import api, synth

pytestmark = pytest.mark.gpu

PCM_TOL_LSB = 1
GR_TOL = 2e-5


@pytest.fixture(scope="module")
def model(blob):
    m = api.Model(blob)
    yield m
    m.close()


def _ctx(monkeypatch, model, B, direct, rg=None, chains=None):
    monkeypatch.setenv("PERCEPNET_SMALL_ROWS", "0")                 # never the small-batch family
    monkeypatch.setenv("PERCEPNET_NN_DIRECT", "1" if direct else "0")
    if rg is not None:
        monkeypatch.setenv("PERCEPNET_NN_DIRECT_RG", str(rg))
    else:
        monkeypatch.delenv("PERCEPNET_NN_DIRECT_RG", raising=False)
    if chains is not None:
        monkeypatch.setenv("PN_NN_CHAINS", str(chains))
    else:
        monkeypatch.delenv("PN_NN_CHAINS", raising=False)
    ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
    d = ctx.describe()
    want = ("direct_rows64" if rg == 2 else "direct_rows32") if direct and rg else ("direct" if direct else "batch")
    assert d["dense"] == "batch" and d["gru"].startswith(want) and d["gru_rb"].startswith(want), d
    return ctx


@pytest.mark.parametrize("B,rg", [(300, 1), (700, 2), (129, 2)], ids=["300x32rows", "700x64rows", "129x64rows"])
def test_direct_family_is_bit_identical_to_the_batch_family(model, oracle, monkeypatch, B, rg):
    """Ragged batches (300 = 2.3 blocks of 128 rows; 700 = 2.7 blocks of 256; 129 = one row into the second 128-row tile of a
    256-row block), 14 frames (more than the 12-slot history ring, the 5- and 3-slot conv rings and both GRU halves): PCM,
    g,r and the network-only entry point equal bit for bit; the last rows still follow the oracle."""
    T = 14
    pcm = synth.synth_batch(B, T, first_stream=40)
    feat = np.random.default_rng(5).standard_normal((B, 70)).astype(np.float32)
    res = {}
    for direct in (False, True):
        ctx = _ctx(monkeypatch, model, B, direct, rg)
        o, g = ctx.run_pcm(pcm)
        r1 = ctx.compute_rnn(feat)
        r2 = ctx.compute_rnn(feat[::-1])
        st = ctx.get_rnn_state()
        res[direct] = (o, g, r1, r2, st)
        ctx.close()
    a, b = res[False], res[True]
    assert np.array_equal(a[0], b[0])
    for i in (1, 2, 3):
        assert np.array_equal(a[i].view(np.uint32), b[i].view(np.uint32)), i
    for k in a[4]:
        assert np.array_equal(a[4][k].view(np.uint32), b[4][k].view(np.uint32)), k
    for s in (0, B - 1):
        ro, rg_ = oracle.run_pcm(pcm[s])
        assert np.abs(b[0][s].astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
        assert np.abs(b[1][s] - rg_).max() <= GR_TOL


def test_direct_family_as_row_range_chains(model, monkeypatch):
    """8192 + 300 streams as one, two and three (capped to two) row-range chains, 64 rows per wave: the chains' shares are whole
    256-row blocks (4352 + 4140) and a block past a chain's last row stores nothing (neither rows nor shadow entries: they are the
    other chain's) — bit-identical to the batch family on one stream."""
    B, T = 8192 + 300, 5
    pool = synth.synth_batch(64, T)
    pcm = pool[np.arange(B) % 64].copy()
    pcm[8192:] = np.roll(pcm[8192:], 11, axis=1)
    feat = np.random.default_rng(3).standard_normal((B, 70)).astype(np.float32)
    ctx = _ctx(monkeypatch, model, B, False, chains=1)
    ref = (*ctx.run_pcm(pcm), ctx.compute_rnn(feat))
    ctx.close()
    for chains in (1, 2, 3):
        ctx = _ctx(monkeypatch, model, B, True, rg=2, chains=chains)
        assert ctx.describe()["nn_chains"].split(":")[0] == str(min(chains, 2)), ctx.describe()
        o, g = ctx.run_pcm(pcm)
        r = ctx.compute_rnn(feat)
        ctx.close()
        assert np.array_equal(ref[0], o), chains
        assert np.array_equal(ref[1].view(np.uint32), g.view(np.uint32)), chains
        assert np.array_equal(ref[2].view(np.uint32), r.view(np.uint32)), chains


def test_direct_family_shadows_follow_every_state_change(model, monkeypatch):
    """One scenario through both families, compared tick by tick: 300 streams x 40 ticks with a per-stream reset at tick 7
    (slots 5, 130, 299), streams 17 / 200 / 299 skipping ticks through the active set (every ring phase crossed), the network
    state replaced by random values at tick 20 (pn_ctx_set_rnn_state_host) and a whole-context reset at tick 30.  The fp32
    shadows are a second copy of the GRU states: any call that forgets them shows up as a difference."""
    import torch
    B, T = 300, 40
    skips = {17: {10, 11, 25}, 200: {11, 21}, 299: {0, 22, 39}}
    pcm = synth.synth_batch(B, T)
    rng = np.random.default_rng(11)
    state = {k: (0.5 * rng.standard_normal((B, n))).astype(np.float32) for k, n in api.Context.RNN_STATE_SHAPES}
    dev = torch.device("cuda:0")
    runs = {}
    for direct in (False, True):
        ctx = _ctx(monkeypatch, model, B, direct, rg=2 if direct else None)
        d_out = torch.full((B, 480), 12345, dtype=torch.int16, device=dev)
        d_gr = torch.full((B, 68), -7.0, dtype=torch.float32, device=dev)
        log = []
        for t in range(T):
            if t == 7:
                ctx.reset_streams([5, 130, 299])
            if t == 20:
                ctx.set_rnn_state(state)
            if t == 30:
                ctx.reset()
            active = [s for s in range(B) if not (s in skips and t in skips[s])]
            d_in = torch.from_numpy(np.ascontiguousarray(pcm[:, t * 480:(t + 1) * 480])).to(dev)
            torch.cuda.synchronize()
            ctx.process_i16_active_dev(d_in.data_ptr(), d_out.data_ptr(), d_gr.data_ptr(), active)
            ctx.synchronize()
            log.append((d_out.cpu().numpy().copy(), d_gr.cpu().numpy().copy()))
            if t in (19, 39):
                log.append(tuple(v.copy() for v in ctx.get_rnn_state().values()))
        ctx.close()
        runs[direct] = log
    for i, (a, b) in enumerate(zip(runs[False], runs[True])):
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), i


@pytest.mark.parametrize("B", [300, 128 * 5 + 1])
def test_fc_gb_on_16x16x4_batch_tiles_is_bit_identical(model, oracle, monkeypatch, B):
    """fc_gb (2560 -> 34, reference rnn.cpp:72-77) of large fp32 contexts runs on 48-column blocks of v_mfma_f32_16x16x4_f32 tiles
    (pn_nn_n48.hip) instead of two padded 32-column tiles: the same k-ascending chain, so g (and with it the PCM) must not
    change by a bit — ragged batches, five frames into every ring, and as two row-range chains."""
    monkeypatch.setenv("PERCEPNET_SMALL_ROWS", "0")
    monkeypatch.setenv("PERCEPNET_N16_ROWS", "0")                   # not the latency-regime kernel either
    monkeypatch.setenv("PERCEPNET_NN_DIRECT", "0")
    T = 8
    pcm = synth.synth_batch(B, T, first_stream=7)
    feat = np.random.default_rng(9).standard_normal((B, 70)).astype(np.float32)
    res = {}
    for n48 in ("0", "1"):
        monkeypatch.setenv("PERCEPNET_N48", n48)
        ctx = api.Context(model, B, nn_mode=api.NN_MFMA)
        assert ctx.describe()["narrow"] == ("fc_gb:n48+fc_rb:batch" if n48 == "1" else "batch"), ctx.describe()
        o, g = ctx.run_pcm(pcm)
        res[n48] = (o, g, ctx.compute_rnn(feat))
        ctx.close()
    assert np.array_equal(res["0"][0], res["1"][0])
    assert np.array_equal(res["0"][1].view(np.uint32), res["1"][1].view(np.uint32))
    assert np.array_equal(res["0"][2].view(np.uint32), res["1"][2].view(np.uint32))
    ro, rg = oracle.run_pcm(pcm[B - 1])
    assert np.abs(res["1"][0][B - 1].astype(np.int32) - ro.astype(np.int32)).max() <= PCM_TOL_LSB
    assert np.abs(res["1"][1][B - 1] - rg).max() <= GR_TOL
