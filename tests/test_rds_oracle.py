"""oracle/rds_oracle.py (numpy / Python restatement of the reference's RDS path) against tests/golden/rds.npz, the output of
the reference's OWN Python model (model/fmSupportLib.py imported in the build container by tests/golden/make_golden_rds.py).
Coefficients and bit-level results exact; float64 signal stages to 1e-12 of full scale (scipy's lfilter and a plain
convolution sum in different orders)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rds_oracle as R  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "rds.npz"))


def ht(a, n=256):
    return a if len(a) <= 2 * n else np.concatenate([a[:n], a[-n:]])


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= tol * scale, (what, np.abs(a - b).max() / scale)


def test_rds_coefficients():
    close(R.band_pass(151, 240000, 54e3, 60e3), G["h_channel"], 1e-15, "channel band-pass")
    close(R.band_pass(151, 240000, 113.5e3, 114.5e3), G["h_carrier"], 1e-15, "carrier band-pass")
    close(R.rrc(2375 * 26, 101), G["h_rrc"], 1e-15, "root raised cosine")
    h = R.imp_response(101 * 247, 240000 * 247, 3e3)
    close(ht(h), G["h_resampler_ht"], 1e-15, "resampler low-pass")
    assert hashlib.sha256(h.tobytes()).digest() == G["h_resampler_sha256"].tobytes()


def test_rds_chain_and_bits():
    x = G["fm_demod"]
    chain = R.RdsChain()
    n = 9600
    for b in range(4):
        out = chain.process(x[b * n:(b + 1) * n])
        for k in ("channel", "carrier", "pll_i", "pll_q", "resampled_i"):
            close(ht(out[k]), G[f"b{b}_{k}_ht"], 1e-10 if "pll" in k else 1e-12, f"block {b} {k}")
        close(out["rrc_i"], G[f"b{b}_rrc_i"], 1e-10, f"block {b} rrc_i")
        close(out["rrc_q"], G[f"b{b}_rrc_q"], 1e-10, f"block {b} rrc_q")
        np.testing.assert_array_equal(out["cdr_bits"], G[f"b{b}_cdr_bits"])
        np.testing.assert_array_equal(out["diff_bits"], G[f"b{b}_diff_bits"])
        np.testing.assert_allclose(out["cdr_state"], G[f"b{b}_cdr_state"], rtol=1e-9)
        fs = G[f"b{b}_framesync"]
        assert (ord(out["offset_type"][0]), len(out["offset_type"]), out["next_index"]) == tuple(int(v) for v in fs)
    close(np.array(chain.pll, float), G["pll_state"], 1e-9, "PLL state")


def test_bit_recovery_corner_cases():
    """CDR and framesync corner cases against what the reference's own functions return (tests/golden/rds.npz): an irregular
    pair mended by flipping the point below the 0.3 limit, one that forces a re-start (with and without the carried pair),
    the flip of the third of three equal-signed points, an odd carried-over count, a noisy block; framesync on the transmitted
    bits, on their complement (no syndrome), from the middle of a block, behind random bits."""
    i = 0
    while f"cdr_case{i}_x" in G.files:
        st = G[f"cdr_case{i}_in"]
        bits, ns = R.cdr(G[f"cdr_case{i}_x"], 26, [np.array([st[0], st[1]]), int(st[2]), int(st[3])], int(st[4]))
        np.testing.assert_array_equal(bits, G[f"cdr_case{i}_bits"], err_msg=f"case {i}")
        np.testing.assert_allclose([ns[0][0], ns[0][1], ns[1], ns[2]], G[f"cdr_case{i}_state"], rtol=1e-12, err_msg=f"case {i}")
        i += 1
    assert i >= 6
    i = 0
    while f"fs_case{i}_bits" in G.files:
        off, idx = R.frame_sync(G[f"fs_case{i}_bits"])
        assert (ord(off[0]), len(off), idx) == tuple(int(v) for v in G[f"fs_case{i}_out"]), i
        i += 1
    assert i >= 4
    np.testing.assert_array_equal(R.diff_decode([1, 1, 0, 1, 1, 1, 0]), [1, 0, 1, 1, 0, 0, 1])
