"""oracle/rds_oracle.py -- CPU restatement (numpy / plain Python) of the reference's RDS path.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).  Parity status: PINNED against golden vectors produced by
the reference's own Python model (tests/golden/rds.npz, made by tests/golden/make_golden_rds.py, which imports
model/fmSupportLib.py in the build container and replays model/fmMonoBlock.py:238-296 call for call).  The reference has
this path only in Python (float64 numpy / math); it never reached its C++ (report p.8), so float64 is the arithmetic to match.

Written from the behaviour of the reference (file:line cited per function), not copied: raw-history block filters in place
of scipy's lfilter state vectors, the polyphase resampler in stream form, the bit recovery with explicit state.
"""
from __future__ import annotations

import math

import numpy as np


# ---- coefficient design (float64) ---------------------------------------------------------------------------
def imp_response(n_taps: int, Fs: float, Fc: float) -> np.ndarray:
    """model/fmSupportLib.py:376-385 impResponse: sinc low-pass times a sin^2 window over i*pi/N (not N-1)."""
    h = np.zeros(n_taps)
    norm = Fc / (Fs / 2)
    c = (n_taps - 1) / 2
    for i in range(n_taps):
        if i == c:
            v = norm
        else:
            a = np.pi * norm * (i - c)
            v = norm * (np.sin(a) / a)
        h[i] = v * (np.sin(i * np.pi / n_taps)) ** 2
    return h


def band_pass(n_taps: int, Fs: float, Fb: float, Fe: float) -> np.ndarray:
    """model/fmSupportLib.py:358-371 bandPass (Python argument order: taps, Fs, Fb, Fe)."""
    h = np.zeros(n_taps)
    center = ((Fe + Fb) / 2) / (Fs / 2)
    width = (Fe - Fb) / (Fs / 2)
    c = (n_taps - 1) / 2
    for i in range(n_taps):
        if i == c:
            v = width
        else:
            a = np.pi * width / 2 * (i - c)
            v = width * (np.sin(a) / a)
        v = v * np.cos(i * np.pi * center)
        h[i] = v * (np.sin(i * np.pi / n_taps)) ** 2
    return h


def rrc(Fs: float, n_taps: int) -> np.ndarray:
    """model/fmSupportLib.py:251-287 impulseResponseRootRaisedCosine: beta 0.90, symbol time 1/2375 s."""
    T, beta = 1 / 2375.0, 0.90
    h = np.empty(n_taps)
    for k in range(n_taps):
        t = float((k - n_taps / 2)) / Fs
        if t == 0.0:
            h[k] = 1.0 + beta * ((4 / math.pi) - 1)
        elif t == -T / (4 * beta) or t == T / (4 * beta):
            h[k] = (beta / np.sqrt(2)) * (((1 + 2 / math.pi) * (math.sin(math.pi / (4 * beta)))) + ((1 - 2 / math.pi) * (math.cos(math.pi / (4 * beta)))))
        else:
            h[k] = (math.sin(math.pi * t * (1 - beta) / T) + 4 * beta * (t / T) * math.cos(math.pi * t * (1 + beta) / T)) / \
                   (math.pi * t * (1 - (4 * beta * t / T) * (4 * beta * t / T)) / T)
    return h


# ---- block primitives -------------------------------------------------------------------------------------------
def fir_block(x: np.ndarray, h: np.ndarray, hist: np.ndarray):
    """scipy.signal.lfilter(h, 1.0, x, zi=...) of model/fmMonoBlock.py:241,251,266 as a plain FIR on [history | block]:
    y[n] = sum_k h[k] x[n-k]; hist = the previous len(h)-1 samples.  -> (y, new history)."""
    xx = np.concatenate([hist, x])
    y = np.convolve(xx, h)[len(h) - 1:len(h) - 1 + len(x)]
    return y, xx[len(xx) - (len(h) - 1):]


def all_pass(x: np.ndarray, hist: np.ndarray):
    """model/fmSupportLib.py:291-295 allPass: a delay by len(hist)."""
    xx = np.concatenate([hist, x])
    return xx[:len(x)], xx[len(x):]


def fm_pll(x: np.ndarray, freq: float, Fs: float, state, ncoScale=2.0, phaseAdjust=0.0, normBandwidth=0.01):
    """model/fmSupportLib.py:297-353 fmPll (float64; returns the in-phase AND quadrature NCO outputs, 7-element state)."""
    Kp = normBandwidth * 2.666
    Ki = normBandwidth * normBandwidth * 3.555
    integ, phase, fI, fQ, last_i, off, last_q = state
    out_i, out_q = np.empty(len(x) + 1), np.empty(len(x) + 1)
    out_i[0], out_q[0] = last_i, last_q
    for k in range(len(x)):
        eD = math.atan2(x[k] * (-fQ), x[k] * (+fI))
        integ = integ + Ki * eD
        phase = phase + Kp * eD + integ
        off += 1
        arg = 2 * math.pi * (freq / Fs) * off + phase
        fI, fQ = math.cos(arg), math.sin(arg)
        out_i[k + 1] = math.cos(arg * ncoScale + phaseAdjust)
        out_q[k + 1] = math.sin(arg * ncoScale + phaseAdjust)
    return out_i, out_q, [integ, phase, fI, fQ, out_i[-1], off, out_q[-1]]


def resample(x: np.ndarray, h: np.ndarray, hist: np.ndarray, decim: int, upsamp: int):
    """model/fmSupportLib.py:388-407 convolveBlockResampleFIR in stream form: y[k] = U * sum_j h[ph + jU] x[floor(kD/U) - j],
    ph = kD mod U, j ascending (the Python model's gain is U; the C++ one's 1+U).  hist = the previous (len(h)-1)//U inputs."""
    xx = np.concatenate([hist, x])
    H = len(hist)
    n_out = int(len(x) * upsamp / decim)
    y = np.zeros(n_out)
    for k in range(n_out):
        m = k * decim
        ph = m % upsamp
        taps = h[ph::upsamp]
        b = H + m // upsamp
        acc = 0.0
        for j in range(len(taps)):
            acc += taps[j] * xx[b - j]
        y[k] = acc * upsamp
    return y, xx[len(xx) - H:]


# ---- bit recovery (model/fmSupportLib.py:103-249) -------------------------------------------------------------------
def symbol_to_bit(pair) -> int:
    """:221-229"""
    return 1 if pair[0] > 0 else 0


def manchester(samples) -> np.ndarray:
    """:203-219 manchestering: (low, high) -> 0, (high, low) -> 1, anything else 0."""
    out = np.zeros(len(samples) // 2)
    for i in range(0, len(samples) - 1, 2):
        if samples[i] > 0 and samples[i + 1] < 0:
            out[i // 2] = 1
    return out


def cdr(x, sps: int, state, block_count: int):
    """:103-200 CDR: sample every sps-th point from `start`, flip the third of three equal-signed points, pair the points,
    re-start one symbol later when a pair is irregular and cannot be mended, Manchester-decode.  state = [pair[2], start, prev_size]."""
    pair = [float(state[0][0]), float(state[0][1])]
    start0 = start = int(state[1])
    prev_size = int(state[2])
    out = []
    while True:
        pts = {}
        size = 0
        i = start
        while i < len(x):
            if i == start and start == start0 and prev_size % 2 == 1:
                pair[1] = x[i]
                out.append(symbol_to_bit(pair))
                pair[0] = pair[1]
                start += sps                        # the loop variable keeps running from the old start (Python range semantics)
                i += sps
                continue
            a, b = pts.get(i - 2 * sps, 0.0), pts.get(i - sps, 0.0)
            if i >= start + 2 * sps and a > 0 and b > 0 and x[i] > 0:
                pts[i] = -x[i]
            elif i >= start + 2 * sps and a < 0 and b < 0 and x[i] < 0:
                pts[i] = -x[i]
            else:
                pts[i] = x[i]
            size += 1
            i += sps
        samples = np.zeros(size)
        for i in range(start, len(x), sps):
            samples[(i - start) // sps] = pts.get(i, 0.0)
        again = False
        for i in range(0, len(samples), 2):
            if i + 1 < len(samples):
                if (samples[i] < 0 and samples[i + 1] < 0) or (samples[i] > 0 and samples[i + 1] > 0):
                    if abs(samples[i]) < 0.3 or abs(samples[i + 1]) < 0.3:
                        if abs(samples[i]) < 0.3:
                            samples[i] = -samples[i]
                        elif abs(samples[i + 1]) < 0.3:
                            samples[i + 1] = -samples[i + 1]
                    else:
                        start += sps
                        if block_count != 0:
                            pair[1] = samples[0]
                            out.append(symbol_to_bit(pair))
                            pair[0] = pair[1]
                        again = True
                        break
        if not again:
            break
    pair[0] = samples[-1]
    last_index = (size - 1) * sps + start
    new_state = [np.array(pair), sps - (len(x) - last_index), size]
    return np.concatenate([np.array(out, float), manchester(samples)]), new_state


def diff_decode(bits) -> np.ndarray:
    """:241-249 diff_decoding"""
    out = np.empty(len(bits))
    out[0] = bits[0]
    out[1:] = (np.asarray(bits)[1:] != np.asarray(bits)[:-1]).astype(float)
    return out


_SYNDROMES = {0x3D8: "A", 0x3D4: "B", 0x25C: "C", 0x3CC: "C_apos", 0x258: "D"}
_PARITY = [0x200, 0x100, 0x080, 0x040, 0x020, 0x010, 0x008, 0x004, 0x002, 0x001, 0x2DC, 0x16E, 0x0B7, 0x287, 0x39F, 0x313, 0x355,
           0x376, 0x1BB, 0x201, 0x3DC, 0x1EE, 0x0F7, 0x2A7, 0x38F, 0x31B]   # rows of :33-58 as 10-bit words, MSB = column 0


def frame_sync(bits):
    """:30-100 framesync: slide over the bit stream, syndrome = 26 bits x parity matrix over GF(2); on a known syndrome jump
    26 bits (stop when fewer than 26 remain), otherwise 1.  -> (offset_type of the last hit or ' ', index to keep from)."""
    n, off = 0, " "
    bits = [int(b) for b in bits]
    while n < len(bits) - 26:
        s = 0
        for i in range(26):
            if bits[n + i] == 1:
                s ^= _PARITY[i]
        if s in _SYNDROMES:
            off = _SYNDROMES[s]
            if len(bits) - (n + 26) < 26:
                break
            n += 26
        else:
            n += 1
    return off, (n if off == " " else n + 26)


class RdsChain:
    """model/fmMonoBlock.py:238-296 per block, state carried as raw histories."""

    def __init__(self, if_Fs=240000, taps=151, upsamp=247, decim=960, sps=26, rrc_taps=101):
        self.Fs, self.U, self.D, self.sps = if_Fs, upsamp, decim, sps
        self.h_ch = band_pass(taps, if_Fs, 54e3, 60e3)
        self.h_car = band_pass(taps, if_Fs, 113.5e3, 114.5e3)
        self.h_rs = imp_response(101 * upsamp, if_Fs * upsamp, 3e3)
        self.h_rrc = rrc(2375 * sps, rrc_taps)
        self.hist_x = np.zeros(taps - 1)
        self.hist_sq = np.zeros(taps - 1)
        self.hist_ap = np.zeros((taps - 1) // 2)
        self.pll = [0.0, 0.0, 1.0, 0.0, 1.0, 0, 1.0]
        H = (101 * upsamp - 1) // upsamp
        self.hist_mi, self.hist_mq = np.zeros(H), np.zeros(H)
        self.hist_ri, self.hist_rq = np.zeros(rrc_taps - 1), np.zeros(rrc_taps - 1)
        self.decoded = np.array([])
        self.block = 0

    def process(self, fm_demod):
        x = np.asarray(fm_demod, np.float64)
        ch, self.hist_x = fir_block(x, self.h_ch, self.hist_x)
        ap, self.hist_ap = all_pass(ch, self.hist_ap)
        car, self.hist_sq = fir_block(ch * ch, self.h_car, self.hist_sq)
        pi_, pq, self.pll = fm_pll(car, 114e3, self.Fs, self.pll, ncoScale=0.5, phaseAdjust=3 * math.pi / 8, normBandwidth=0.002)
        ri, self.hist_mi = resample(pi_[:-1] * ap * 2, self.h_rs, self.hist_mi, self.D, self.U)
        rq, self.hist_mq = resample(pq[:-1] * ap * 2, self.h_rs, self.hist_mq, self.D, self.U)
        yi, self.hist_ri = fir_block(ri, self.h_rrc, self.hist_ri)
        yq, self.hist_rq = fir_block(rq, self.h_rrc, self.hist_rq)
        bits, st = cdr(yi, self.sps, [np.zeros(2), 158, 0], self.block)      # the model re-makes this state every block (:276-280)
        dd = diff_decode(bits)
        self.decoded = np.concatenate([self.decoded, dd])
        off, idx = frame_sync(self.decoded)
        self.decoded = self.decoded[idx:]
        self.block += 1
        return dict(channel=ch, carrier=car, pll_i=pi_, pll_q=pq, resampled_i=ri, rrc_i=yi, rrc_q=yq, cdr_bits=bits, diff_bits=dd,
                    cdr_state=np.array([st[0][0], st[0][1], st[1], st[2]], float), offset_type=off, next_index=idx)
