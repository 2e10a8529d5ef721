"""Synthetic RDS-bearing FM multiplex at the discriminator output (test infrastructure).

The reference's RDS path (model/fmMonoBlock.py:238-296) starts from fm_demod; its own test input is an off-air
recording that is not in the repository (../data/lab3_iq_samples/samples8.raw).  This generator stands in for it:
an IEC 62106 bit stream -- groups of four 26-bit blocks, 16 information bits + 10-bit checkword (CRC with
g(x) = x^10 + x^8 + x^7 + x^5 + x^4 + x^3 + 1, plus the offset word A / B / C / D) -- differentially encoded,
biphase (Manchester) coded at 1187.5 bit/s = 2375 chips/s, on a suppressed 57 kHz subcarrier locked to the third
harmonic of the 19 kHz pilot, added to a mono + pilot multiplex.  The syndromes the reference's framesync looks
for (model/fmSupportLib.py:65-89) are the standard ones of exactly this code.
"""
import numpy as np

OFFSETS = {"A": 0x0FC, "B": 0x198, "C": 0x168, "D": 0x1B4}
GEN = 0x5B9  # x^10 + x^8 + x^7 + x^5 + x^4 + x^3 + 1


def checkword(info16: int) -> int:
    reg = info16 << 10
    for bit in range(25, 9, -1):
        if reg & (1 << bit):
            reg ^= GEN << (bit - 10)
    return reg & 0x3FF


def rds_bits(n_groups: int, seed: int = 1) -> np.ndarray:
    """n_groups x 104 information+check bits (MSB first), blocks A B C D."""
    rng = np.random.default_rng(seed)
    bits = []
    for _ in range(n_groups):
        for name in "ABCD":
            info = int(rng.integers(0, 1 << 16))
            word = (info << 10) | (checkword(info) ^ OFFSETS[name])
            bits += [(word >> (25 - k)) & 1 for k in range(26)]
    return np.array(bits, np.uint8)


def rds_demod_signal(n_samples: int, if_Fs: float = 240e3, seed: int = 1, amplitude: float = 0.06, chip_offset: float = 0.0,
                     noise: float = 0.0) -> tuple[np.ndarray, np.ndarray]:
    """-> (fm_demod float32[n_samples], the transmitted bits).  chip_offset shifts the chip grid (IF samples): the
    reference's clock recovery starts sampling at a fixed index, so the test signal is placed where it looks."""
    t = np.arange(n_samples, dtype=np.float64) / if_Fs
    chip_rate = 2375.0
    n_chips = int(np.ceil(n_samples / if_Fs * chip_rate)) + 4
    bits = rds_bits(n_chips // 208 + 2, seed)
    d = np.zeros(len(bits), np.int8)
    prev = 0
    for i, b in enumerate(bits):                      # differential encoding (the receiver outputs d[i] != d[i-1])
        prev ^= int(b)
        d[i] = prev
    chips = np.empty(2 * len(d), np.float64)          # biphase: 1 -> (+, -), 0 -> (-, +)  (model/fmSupportLib.py:203-219)
    chips[0::2] = np.where(d == 1, 1.0, -1.0)
    chips[1::2] = -chips[0::2]
    pos = (np.arange(n_samples) - chip_offset) * (chip_rate / if_Fs)
    idx = np.clip(np.floor(pos).astype(np.int64), 0, len(chips) - 1)
    frac = pos - np.floor(pos)
    base = chips[idx] * np.sin(np.pi * frac)          # half-sine chips: smooth, zero at the chip edges, peak at the centre
    pilot_phase = 2 * np.pi * 19e3 * t + 0.3
    mono = 0.25 * np.cos(2 * np.pi * 1e3 * t) + 0.15 * np.cos(2 * np.pi * 2.5e3 * t)
    x = mono + 0.1 * np.cos(pilot_phase) + amplitude * base * np.cos(3 * pilot_phase)
    if noise:
        x = x + noise * np.random.default_rng(seed + 99).standard_normal(n_samples)
    return x.astype(np.float32), bits
