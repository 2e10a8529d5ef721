"""Deterministic synthetic FM-broadcast I/Q (numpy), the bench / smoke input.

A constant-envelope FM signal (75 kHz deviation) carrying a stereo multiplex:
m(t) = 0.45(L+R) + 0.1 cos(2pi 19k t) + 0.45 (L-R) cos(2pi 38k t), with
L = 0.5(cos 2pi 1k t + cos 2pi 3k t), R = cos 2pi 2k t, quantised to the RTL-SDR
wire format the reference reads on stdin (interleaved unsigned 8-bit I,Q,
src/iofunc.cpp:128-135).  Constant envelope => no discriminator blow-ups;
audio stays within +-0.5 => no s16 overflow; pilot present => the stereo PLL
locks.  The phase integral is written in closed form, so any window
of the stream is a pure function of (rf_Fs, seed, start) and vectorises.

This is the same signal as oracle/fm_oracle.c:fmo_synth_fm_u8 (tests check the
two agree up to rare 1-LSB ties from libm differences); it lives here so that
the product-side tools never reach into oracle/.
"""
from __future__ import annotations

import numpy as np

_COMPONENTS = (  # (amplitude, frequency) of the cosine terms of m(t)
    (0.45 * 0.5, 1e3), (0.45 * 0.5, 3e3), (0.45 * 1.0, 2e3), (0.1, 19e3),
    (0.45 * 0.25, 38e3 - 1e3), (0.45 * 0.25, 38e3 + 1e3), (0.45 * 0.25, 38e3 - 3e3), (0.45 * 0.25, 38e3 + 3e3),
    (-0.45 * 0.5, 38e3 - 2e3), (-0.45 * 0.5, 38e3 + 2e3),
)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def synth_fm_u8(n_samples: int, rf_Fs: float = 2.4e6, seed: int = 0x3D74, start: int = 0) -> np.ndarray:
    """-> uint8[2*n_samples], interleaved I,Q."""
    idx = np.arange(start, start + n_samples, dtype=np.uint64)
    per = int(rf_Fs / 1000.0)
    if per * 1000.0 == rf_Fs:  # the multiplex repeats every 1 ms: keep t small and exact
        t = (idx % np.uint64(per)).astype(np.float64) / rf_Fs
    else:
        t = idx.astype(np.float64) / rf_Fs
    w = 2.0 * np.pi
    integ = np.zeros(n_samples, np.float64)
    for amp, f in _COMPONENTS:
        integ += amp * np.sin(w * f * t) / (w * f)
    phi = w * 75e3 * integ
    with np.errstate(over="ignore"):
        r = _splitmix64(np.uint64(seed) ^ (idx * np.uint64(0xD1342543DE82EF95)))
    di = (r & np.uint64(0xFFFFFFFF)).astype(np.float64) / 4294967296.0 - 0.5
    dq = (r >> np.uint64(32)).astype(np.float64) / 4294967296.0 - 0.5
    out = np.empty(2 * n_samples, np.uint8)
    out[0::2] = np.clip(np.floor(128.0 + 127.0 * 0.8 * np.cos(phi) + di + 0.5), 0, 255).astype(np.uint8)
    out[1::2] = np.clip(np.floor(128.0 + 127.0 * 0.8 * np.sin(phi) + dq + 0.5), 0, 255).astype(np.uint8)
    return out
