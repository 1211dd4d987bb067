"""Synthetic benchmark inputs (SURVEY.md §8d "Synthetic inputs").

16 kHz mono float32 utterances built from short "phone-like" segments (60-220 ms): each
segment is band-limited noise plus a few harmonics of a random pitch with its own
spectral tilt and level, with occasional short silences, then a per-utterance gain
U(0.25, 1).  Stationary noise alone makes every encoder frame look the same, which in
turn makes greedy RNN-T on random weights degenerate (all blank or max_symbols on every
frame); time-varying spectra give the decode loop a realistic emission pattern.
Deterministic in (n_utt, seconds, seed, ragged)."""
import numpy as np


def _segment(rng, n, samplerate):
    t = np.arange(n, dtype=np.float32) / samplerate
    kind = rng.random()
    if kind < 0.12:                                   # silence / breath
        return (0.002 * rng.standard_normal(n)).astype(np.float32)
    x = rng.standard_normal(n + 8).astype(np.float32)
    taps = 1 + int(rng.integers(0, 8))                # crude low-pass of random width
    noise = np.convolve(x, np.ones(taps, np.float32) / taps, mode="valid")[:n]
    if rng.random() < 0.5:                            # high-pass-ish variant
        noise = noise - np.concatenate([[0.0], noise[:-1]]).astype(np.float32) * 0.9
    f0 = rng.uniform(90.0, 320.0)
    voiced = np.zeros(n, np.float32)
    for h in range(1, int(rng.integers(2, 9))):
        voiced += (rng.uniform(0.2, 1.0) / h) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)).astype(np.float32)
    mix = rng.random()
    seg = mix * 0.1 * noise / (np.std(noise) + 1e-6) + (1 - mix) * 0.1 * voiced
    ramp = min(80, n // 4)
    if ramp > 0:
        w = np.linspace(0, 1, ramp, dtype=np.float32)
        seg[:ramp] *= w
        seg[-ramp:] *= w[::-1]
    return (rng.uniform(0.3, 1.0) * seg).astype(np.float32)


def synthetic_batch(n_utt: int, seconds: float = 10.0, seed: int = 1234, ragged: bool = False,
                    min_seconds: float = 2.0, samplerate: int = 16000):
    """-> (audio f32[n_utt, Lmax] zero padded at the end, lengths i64[n_utt])"""
    rng = np.random.default_rng(seed)
    Lmax = int(round(seconds * samplerate))
    if ragged:
        lens = rng.integers(int(min_seconds * samplerate), Lmax + 1, size=n_utt)
    else:
        lens = np.full(n_utt, Lmax)
    audio = np.zeros((n_utt, Lmax), dtype=np.float32)
    for b in range(n_utt):
        L = int(lens[b])
        pos = 0
        while pos < L:
            n = min(L - pos, int(rng.uniform(0.06, 0.22) * samplerate))
            audio[b, pos:pos + n] = _segment(rng, n, samplerate)
            pos += n
        audio[b, :L] *= rng.uniform(0.25, 1.0)
    return audio, lens.astype(np.int64)
