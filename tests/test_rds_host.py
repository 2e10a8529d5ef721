"""The host-side part of the RDS path in libfmrx (coefficient design in float64, clock and data recovery, Manchester /
differential decoding, frame synchronisation: C++ in csrc/rds.hip behind fmrx_rds_*) against the golden vectors produced by
the reference's own Python functions (tests/golden/rds.npz).  No GPU involved: runs anywhere."""
import hashlib
import os

import numpy as np

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rds.npz"))


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def test_rds_coefficients_host(fmrx):
    assert rel(fmrx.rdsBandPass(151, 240000, 54e3, 60e3), G["h_channel"]) <= 1e-15
    assert rel(fmrx.rdsBandPass(151, 240000, 113.5e3, 114.5e3), G["h_carrier"]) <= 1e-15
    assert rel(fmrx.impulseResponseRootRaisedCosine(2375 * 26, 101), G["h_rrc"]) <= 1e-15
    h = fmrx.rdsImpResponse(101 * 247, 240000 * 247, 3e3)
    assert rel(np.concatenate([h[:256], h[-256:]]), G["h_resampler_ht"]) <= 1e-15
    print("resampler taps bit-identical to the model's:", hashlib.sha256(h.tobytes()).digest() == G["h_resampler_sha256"].tobytes())


def test_rds_bit_recovery_host(fmrx):
    i = 0
    while f"cdr_case{i}_x" in G.files:
        st = G[f"cdr_case{i}_in"]
        bits, ns = fmrx.CDR(G[f"cdr_case{i}_x"], 26, [np.array([st[0], st[1]]), int(st[2]), int(st[3])], int(st[4]))
        np.testing.assert_array_equal(bits, G[f"cdr_case{i}_bits"], err_msg=f"CDR case {i}")
        np.testing.assert_allclose([ns[0][0], ns[0][1], ns[1], ns[2]], G[f"cdr_case{i}_state"], rtol=1e-12, err_msg=f"CDR case {i}")
        i += 1
    assert i >= 6
    for b in range(4):                                   # the four blocks of the chain fixture: model's RRC output in, model's bits out
        bits, ns = fmrx.CDR(G[f"b{b}_rrc_i"], 26, [np.zeros(2), 158, 0], b)
        np.testing.assert_array_equal(bits, G[f"b{b}_cdr_bits"])
        np.testing.assert_allclose([ns[0][0], ns[0][1], ns[1], ns[2]], G[f"b{b}_cdr_state"], rtol=1e-12)
        np.testing.assert_array_equal(fmrx.diff_decoding(bits), G[f"b{b}_diff_bits"])
    i = 0
    while f"fs_case{i}_bits" in G.files:
        off, idx = fmrx.framesync(G[f"fs_case{i}_bits"])
        assert (ord(off[0]), len(off), idx) == tuple(int(v) for v in G[f"fs_case{i}_out"]), i
        i += 1
    assert i >= 4
