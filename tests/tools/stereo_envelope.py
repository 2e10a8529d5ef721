#!/usr/bin/env python3
"""Stereo error vs stream time, cause by cause: the same 2.13 s mode-0 stereo stream
through the GPU pipeline in four configurations, each against the oracle (= the compiled reference, bit for
bit) for the whole stream, RMS error of the left channel per {win / au_Fs:.1f} s window, in absolute terms and in units of
ulp(trigArg(t)):
   fast / parallel   default: specialised kernels, parallel-in-time PLL, fast math (closed-form phase detector)
   fast / serial     pll_mode 1: same math, serial recurrence               -> isolates the segment merge
   fast / glibc      pll_mode 2: serial recurrence with glibc's functions   -> isolates the math library
   bit-exact         set_force_generic: reference evaluation order upstream too -> must be 0
Checker script (imports the oracle): lives under tests/.  Usage: python tests/tools/stereo_envelope.py [block_bytes]"""
import importlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
from _oracle import Oracle  # noqa: E402

fmrx = importlib.import_module("software-defined-radio_amd")


def ulp(t):
    ta = 2 * np.pi * 19e3 / 240e3 * np.maximum(240e3 * np.asarray(t, np.float64), 1.0)
    return 2.0 ** (np.floor(np.log2(ta)) - 23)


def main():
    bb_arg = sys.argv[1] if len(sys.argv) > 1 else str(2 * 1024000)      # bytes per call, or xN = N reference blocks
    nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    o = Oracle()
    p = o.mode_params(mode, 101, 101, 101)
    bb = p.block_bytes * int(bb_arg[1:]) if bb_arg.startswith('x') else int(bb_arg)
    iq = o.synth_fm_u8(p.block_bytes // 2 * nblk, rf_Fs=p.rf_Fs, seed=0x3D74)
    po = o.pipeline(mode, 2)
    Lo = np.concatenate([po.process(iq[k:k + p.block_bytes])["audio_l"] for k in range(0, len(iq), p.block_bytes)])
    if_Fs = p.rf_Fs / p.rf_decim
    au_Fs = len(Lo) / (len(iq) / 2 / p.rf_Fs)                     # audio samples per second
    win = int(round(au_Fs * (0.1 if nblk <= 200 else 1.0)))       # 0.1 s windows (1 s for long streams)
    t_end = (np.arange(len(Lo) // win) + 1) * (win / au_Fs)
    global ulp
    ulp = lambda t: 2.0 ** (np.floor(np.log2(2 * np.pi * 19e3 / if_Fs * np.maximum(if_Fs * np.asarray(t, np.float64), 1.0))) - 23)
    rows = {}
    diag = {}
    sets = {"default": (("fast/parallel", {}), ("fast/serial", {"pll_mode": 1}), ("fast/glibc", {"pll_mode": 2}),
                        ("par W512 L64", {"pll_warmup": 512, "pll_segment": 64}), ("par W384 L64", {"pll_warmup": 384, "pll_segment": 64}),
                        ("par W256 L64", {"pll_warmup": 256, "pll_segment": 64}), ("bit-exact", {"generic": 1})),
            # lanes started from the locked loop as a linear system of the input's signs (pll_start = 1), by warm-up length
            "lti": (("start 0 W512", {"pll_start": 0}), ("lti W0", {"pll_start": 1, "pll_warmup": 0}), ("lti W64", {"pll_start": 1, "pll_warmup": 64}),
                    ("lti W128", {"pll_start": 1, "pll_warmup": 128}), ("lti W256", {"pll_start": 1, "pll_warmup": 256}),
                    ("fast/serial", {"pll_mode": 1}))}
    for name, cfg in sets[sys.argv[3] if len(sys.argv) > 3 else "default"]:
        pl = fmrx.Pipeline(mode, 2, max_block_bytes=bb)
        for k, v in cfg.items():
            if k == "generic":
                pl.set_force_generic(True)
            else:
                pl.set_option(k, v)
        t0 = time.perf_counter()
        L = np.concatenate([pl.process(iq[k:k + bb])["audio_l"] for k in range(0, len(iq) // bb * bb, bb)])
        dt = time.perf_counter() - t0
        diag[name] = pl.pll_diagnostics() if hasattr(pl, "pll_diagnostics") else None
        n = min(len(L), len(Lo)) // win * win
        d = (L[:n].astype(np.float64) - Lo[:n]).reshape(-1, win)
        rows[name] = (np.sqrt(np.mean(d * d, axis=1)), dt)
    names = list(rows)
    print(f"# mode {mode} stereo, {nblk} reference blocks = {len(iq) / 2 / p.rf_Fs:.2f} s, fed as {bb}-byte blocks; "
          f"left-channel RMS error vs the oracle per {win / au_Fs:.1f} s window: absolute (in ulp(trigArg))")
    print("t_end[s]  ulp(trigArg)  " + "  ".join(f"{n:>22s}" for n in names))
    for i, t in enumerate(t_end[: len(rows[names[0]][0])]):
        print(f"{t:7.1f}  {ulp(t):11.2e}  " + "  ".join(f"{rows[n][0][i]:12.2e} ({rows[n][0][i] / ulp(t):5.3f})" for n in names))
    print("repaired segments / largest accepted |dphase|, |dinteg| of the last call: " + ", ".join(f"{n} {diag[n]}" for n in names))
    print("wall seconds incl. host copies: " + ", ".join(f"{n} {rows[n][1]:.2f}" for n in names))


if __name__ == "__main__":
    main()
