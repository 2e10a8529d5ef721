#!/usr/bin/env python3
"""tests/golden/rds.npz from the REFERENCE's own Python model (build container only).

Row f4 of SURVEY 8(f): the RDS path exists only in the reference's Python model (model/fmMonoBlock.py:238-296 on top of
model/fmSupportLib.py).  This script imports model/fmSupportLib.py where it lies (python -B: /root/reference is read-only),
replays the RDS lines of fmMonoBlock.py call for call (scipy.signal.lfilter for its band-pass / RRC filters, the reference's
allPass, fmPll, convolveBlockResampleFIR, CDR, diff_decoding, framesync) on the synthetic RDS-bearing discriminator signal of
tests/rds_signal.py, and stores inputs and outputs.  Only data is written: no reference source.

    python3 -B tests/golden/make_golden_rds.py
"""
import hashlib
import math
import os
import sys

import numpy as np
from scipy import signal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.environ.get("FMRX_REFERENCE", "/root/reference") + "/model")
import fmSupportLib as L  # noqa: E402  the reference's model
from rds_signal import rds_demod_signal  # noqa: E402

# mode 0 of model/fmMonoBlock.py:72-78, 114, 137-141
IF_FS, RDS_TAPS, U, D, SPS, RRC_TAPS = 240000, 151, 247, 960, 26, 101
NBLK, N_IF = 4, 9600           # block_size = 2*rf_decim*audio_decim*rds_decim*2 bytes = 9600 discriminator samples (:164)
CHIP_OFFSET = float(os.environ.get("RDS_CHIP_OFFSET", "66"))   # places the chip centres where the reference's CDR starts sampling (index 158, every 26th)


def ht(a, n=256):
    return a.copy() if len(a) <= 2 * n else np.concatenate([a[:n], a[-n:]])


def run(x_all):
    ch = L.bandPass(RDS_TAPS, IF_FS, 54e3, 60e3)
    car = L.bandPass(RDS_TAPS, IF_FS, 113.5e3, 114.5e3)
    rs = L.impResponse(101 * U, IF_FS * U, 3e3)
    rrc = L.impulseResponseRootRaisedCosine(2375 * SPS, RRC_TAPS)
    st_ch, st_car = np.zeros(RDS_TAPS - 1), np.zeros(RDS_TAPS - 1)
    st_ap = np.zeros(int((RDS_TAPS - 1) / 2))
    st_pll = [0.0, 0.0, 1.0, 0.0, 1.0, 0, 1.0]
    st_rs, st_rs2 = np.zeros(101 * U - 1), np.zeros(101 * U - 1)
    st_rrc, st_rrc2 = np.zeros(RRC_TAPS - 1), np.zeros(RRC_TAPS - 1)
    out = {"h_channel": ch, "h_carrier": car, "h_resampler_ht": ht(rs), "h_rrc": rrc,
           "h_resampler_sha256": np.frombuffer(hashlib.sha256(rs.tobytes()).digest(), np.uint8)}
    decoded = np.array([])
    for b in range(NBLK):
        fm_demod = x_all[b * N_IF:(b + 1) * N_IF].astype(np.float64)
        filt, st_ch = signal.lfilter(ch, 1.0, fm_demod, zi=st_ch)                       # :241
        ap, st_ap = L.allPass(filt, st_ap)                                             # :245
        sq = filt * filt                                                               # :248
        cf, st_car = signal.lfilter(car, 1.0, sq, zi=st_car)                            # :251
        pll, pllq, st_pll = L.fmPll(cf, 114e3, IF_FS, st_pll, ncoScale=0.5, phaseAdjust=(3 * math.pi / 8), normBandwidth=0.002)   # :254
        mix, mixq = pll[:-1] * ap * 2, pllq[:-1] * ap * 2                              # :259, :270
        r1, st_rs = L.convolveBlockResampleFIR(mix, rs, st_rs, D, U)                   # :262
        r2, st_rs2 = L.convolveBlockResampleFIR(mixq, rs, st_rs2, D, U)                # :271
        y1, st_rrc = signal.lfilter(rrc, 1.0, r1, zi=st_rrc)                            # :266
        y2, st_rrc2 = signal.lfilter(rrc, 1.0, r2, zi=st_rrc2)                          # :273
        samples, state = L.CDR(y1, SPS, [np.zeros(2), 158, 0], b)                       # :276-289 (the state is re-made every block there)
        dd = L.diff_decoding(samples)                                                  # :292
        decoded = np.concatenate((decoded, dd))
        off, idx = L.framesync(decoded)                                                # :296
        decoded = decoded[idx:]
        for k, v in (("channel", filt), ("carrier", cf), ("pll_i", pll), ("pll_q", pllq), ("resampled_i", r1)):
            out[f"b{b}_{k}_ht"] = ht(np.asarray(v))            # head and tail (256 each) of the long intermediates
        for k, v in (("rrc_i", y1), ("rrc_q", y2), ("cdr_bits", samples), ("diff_bits", dd)):
            out[f"b{b}_{k}"] = np.asarray(v)
        out[f"b{b}_cdr_state"] = np.array([state[0][0], state[0][1], state[1], state[2]], np.float64)
        out[f"b{b}_framesync"] = np.array([ord(off[0]), len(off), idx], np.int64)
        print(f"block {b}: offset_type {off!r} next index {idx}; {len(samples)} bits; |rrc_i| at the sampling points "
              f"{np.abs(y1[158::SPS]).mean():.3f}, rrc_i rms {np.sqrt(np.mean(y1 ** 2)):.3f}")
    out["pll_state"] = np.array(st_pll, np.float64)
    return out


if __name__ == "__main__":
    x, bits = rds_demod_signal(NBLK * N_IF, IF_FS, seed=7, chip_offset=CHIP_OFFSET)
    out = run(x)
    out["fm_demod"], out["tx_bits"], out["chip_offset"] = x, bits, np.array([CHIP_OFFSET])
    # bit-recovery corner cases through the reference's own CDR / framesync: an irregular pair mended by flipping the
    # point below the 0.3 limit, one that forces a re-start (with and without the carried pair), the "third of three
    # equal-signed points" flip, an odd carried-over count, noise
    rng = np.random.default_rng(5)
    cases = []
    base = np.zeros(26 * 12)
    base[2::26] = [1, -1, 1, -1, -1, 1, 0.2, 0.9, 1, -1, 1, -1]
    cases.append((base.copy(), [np.zeros(2), 2, 0], 1))
    base[2 + 26 * 6] = 0.8
    cases.append((base.copy(), [np.zeros(2), 2, 0], 1))
    cases.append((base.copy(), [np.zeros(2), 2, 0], 0))
    cases.append((base.copy(), [np.array([0.7, 0.0]), 2, 5], 3))
    noisy = out["b2_rrc_i"] + 0.35 * rng.standard_normal(len(out["b2_rrc_i"]))
    cases.append((noisy, [np.zeros(2), 158, 0], 2))
    cases.append((noisy[:1500], [np.array([-0.4, 0.0]), 7, 3], 5))
    for i, (xx, st, bc) in enumerate(cases):
        got, ns = L.CDR(xx.copy(), SPS, [st[0].copy(), st[1], st[2]], bc)
        out[f"cdr_case{i}_x"], out[f"cdr_case{i}_in"] = xx, np.array([st[0][0], st[0][1], st[1], st[2], bc], float)
        out[f"cdr_case{i}_bits"], out[f"cdr_case{i}_state"] = np.asarray(got, float), np.array([ns[0][0], ns[0][1], ns[1], ns[2]], float)
    for i, stream in enumerate((bits[:26 * 9 + 5], bits[:300] ^ 1, bits[7:7 + 26 * 5], np.concatenate([rng.integers(0, 2, 40), bits[:26 * 4]]))):
        off, idx = L.framesync(stream.astype(float))
        out[f"fs_case{i}_bits"], out[f"fs_case{i}_out"] = stream.astype(np.uint8), np.array([ord(off[0]), len(off), idx], np.int64)
    if len(sys.argv) > 1 and sys.argv[1] == "probe":
        y = out["b1_rrc_i"]
        best = max(range(SPS), key=lambda p: np.abs(y[p::SPS]).mean())
        print("best sampling phase mod 26:", best, "mean |y| there", np.abs(y[best::SPS]).mean(), "at 158 % 26 =", 158 % SPS)
    else:
        np.savez_compressed(os.path.join(HERE, "rds.npz"), **out)
        print("rds.npz:", os.path.getsize(os.path.join(HERE, "rds.npz")) // 1024, "KiB")
