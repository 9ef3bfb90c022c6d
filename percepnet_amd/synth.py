"""Synthetic 48 kHz test/bench audio (SURVEY.md §8(d) "Synthetic inputs").

The reference's two sample PCMs are absent (.MISSING_LARGE_BLOBS), so every parity and bench
input is generated: stream ``s`` is a gated harmonic source with vibrato plus white noise;
5 % of the streams are noise bursts with 1 s silences, 5 % are 200+300 Hz two-tone (pitch
ambiguity for remove_doubling, pitch.cpp:424) and 5 % are near-full-scale ("loud": the only
frames for which the reference's ``silence = sum(Ex) < 0.1`` test, denoise.cpp:433, is false, so
the only ones that exercise the pitch-filter branch, 536-538).  Values are int16 PCM, the CLI's format (main.cpp:30-34).
"""
import numpy as np

FS = 48000
FRAME = 480
BASE_SEED = 20260925


def stream_kind(s):
    m = s % 20
    if m == 7:
        return "bursts"
    if m == 13:
        return "twotone"
    if m == 3:
        return "loud"
    return "voiced"


def synth_stream(s, n_frames, base_seed=BASE_SEED):
    """-> int16[n_frames*480] for stream index ``s``."""
    rng = np.random.default_rng(base_seed + s)
    n = n_frames * FRAME
    t = np.arange(n) / FS
    kind = stream_kind(s)
    F = rng.uniform(80, 320)
    a = rng.uniform(0.05, 0.5)
    ns = rng.uniform(0.005, 0.1)
    ph = rng.uniform(0, 1)
    noise = rng.standard_normal(n)
    if kind == "voiced":
        f0 = F + 30.0 * np.sin(2 * np.pi * 0.7 * t)
        phi = 2 * np.pi * np.cumsum(f0) / FS
        v = ((t * 1.3 + ph) % 1.0 < 0.5).astype(np.float64)
        x = np.zeros(n)
        for k in range(1, 12):
            x += (0.5 / k) * np.sin(k * phi)
        x = a * v * x + ns * noise
    elif kind == "loud":
        # near-full-scale (clipping) voiced source: with the reference's 1/960 FFT scale and
        # int16/32768 input, only frames this loud make sum(Ex) >= 0.1, i.e. take the NON-silent
        # branch that applies the pitch filter (denoise.cpp:433,536-538)
        f0 = F + 30.0 * np.sin(2 * np.pi * 0.7 * t)
        phi = 2 * np.pi * np.cumsum(f0) / FS
        v = ((t * 1.3 + ph) % 1.0 < 0.6).astype(np.float64)
        x = np.zeros(n)
        for k in range(1, 12):
            x += (0.5 / k) * np.sin(k * phi)
        x = 2.5 * v * x + 0.25 * noise
    elif kind == "bursts":
        gate = (((t + ph) % 2.0) < 1.0).astype(np.float64)  # 1 s noise, 1 s digital silence
        x = (a * 0.5) * gate * noise
    else:
        x = a * (0.5 * np.sin(2 * np.pi * 200 * t) + 0.5 * np.sin(2 * np.pi * 300 * t + ph)) + ns * noise
    return np.clip(np.round(32768.0 * x), -32768, 32767).astype(np.int16)


def synth_batch(n_streams, n_frames, first_stream=0, base_seed=BASE_SEED):
    """-> int16[n_streams, n_frames*480]"""
    return np.stack([synth_stream(first_stream + s, n_frames, base_seed) for s in range(n_streams)])


def _synth_chunk(args):
    lo, hi, n_frames, base_seed = args
    return np.stack([synth_stream(s, n_frames, base_seed) for s in range(lo, hi)])


def synth_batch_parallel(n_streams, n_frames, first_stream=0, base_seed=BASE_SEED, workers=None):
    """synth_batch on worker PROCESSES (fresh interpreters fed over pipes: safe in a process that already
    initialised HIP and independent of what __main__ is): the long-horizon parity tests need 1024 distinct streams x
    1000 frames (~0.1 s each on one core)."""
    import os
    import subprocess
    import sys
    from concurrent.futures import ThreadPoolExecutor
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    workers = max(1, min(workers, 64, n_streams // 4))
    if workers == 1:
        return synth_batch(n_streams, n_frames, first_stream, base_seed)
    step = (n_streams + workers - 1) // workers
    jobs = [(first_stream + lo, first_stream + min(lo + step, n_streams)) for lo in range(0, n_streams, step)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from percepnet_amd import synth; "
            "sys.stdout.buffer.write(synth._synth_chunk((%d, %d, %d, %d)).tobytes())")
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")

    def run(job):
        out = subprocess.run([sys.executable, "-c", code % (root, job[0], job[1], n_frames, base_seed)],
                             stdout=subprocess.PIPE, check=True, env=env).stdout
        return np.frombuffer(out, np.int16).reshape(job[1] - job[0], n_frames * FRAME)

    with ThreadPoolExecutor(len(jobs)) as ex:
        return np.concatenate(list(ex.map(run, jobs)))


def synth_batch_fast(n_streams, n_frames, seed=0):
    """Cheap bench filler for very large batches: a small pool of distinct streams (all three
    kinds) tiled to n_streams with per-stream sample rotation.  Parity never uses this."""
    pool = synth_batch(min(n_streams, 64), n_frames, base_seed=BASE_SEED + 7919 * seed)
    reps = (n_streams + pool.shape[0] - 1) // pool.shape[0]
    out = np.tile(pool, (reps, 1))[:n_streams].copy()
    for r in range(1, reps):
        lo, hi = r * pool.shape[0], min((r + 1) * pool.shape[0], n_streams)
        out[lo:hi] = np.roll(out[lo:hi], 37 * r, axis=1)
    return out


def synth_pair(p, n_frames, base_seed=BASE_SEED):
    """-> (speech, noisy) int16[n_frames*480] for training-feature pair ``p``: the speech file is
    synth_stream(p); the noisy file is that speech plus white noise (SNR varies with p) plus a
    quieter interfering stream.  Every 4th pair has 12 frames of digital silence in both files (at
    NORM_RATIO 1 only exact zeros reach train()'s `silence` branch, denoise.cpp:433,744)."""
    speech = synth_stream(p, n_frames, base_seed).astype(np.int32)
    rng = np.random.default_rng(base_seed + 100003 + p)
    sigma = 10.0 ** rng.uniform(1.5, 3.3)
    noise = rng.standard_normal(n_frames * FRAME) * sigma
    interf = synth_stream(p + 1000, n_frames, base_seed).astype(np.int32) // 4
    noisy = speech + np.rint(noise).astype(np.int32) + interf
    if p % 4 == 3 and n_frames > 24:
        speech[8 * FRAME:20 * FRAME] = 0
        noisy[8 * FRAME:20 * FRAME] = 0
    return (np.clip(speech, -32768, 32767).astype(np.int16), np.clip(noisy, -32768, 32767).astype(np.int16))


def synth_pairs(n_pairs, n_frames, base_seed=BASE_SEED):
    sp = np.empty((n_pairs, n_frames * FRAME), np.int16)
    no = np.empty((n_pairs, n_frames * FRAME), np.int16)
    for p in range(n_pairs):
        sp[p], no[p] = synth_pair(p, n_frames, base_seed)
    return sp, no
