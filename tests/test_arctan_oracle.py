"""The oracle's restatement of the Python model's arctangent demodulator (oracle/arctan_oracle.py; fmDemodArctan,
model/fmSupportLib.py:502-531) against tests/golden/arctan.npz, which the reference's own fmSupportLib.py produced
(tests/golden/make_golden_arctan.py, build container): bit for bit, state included.  No GPU."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
from arctan_oracle import fm_demod_arctan  # noqa: E402


def test_arctan_oracle_equals_the_model():
    g = np.load(os.path.join(HERE, "golden", "arctan.npz"))
    phase = 0.0
    for b in range(int(g["nblk"][0])):
        d, phase = fm_demod_arctan(g[f"b{b}_if_i"].astype(np.float64), g[f"b{b}_if_q"].astype(np.float64), phase)
        np.testing.assert_array_equal(d, g[f"b{b}_demod"])
        assert phase == float(g[f"b{b}_phase"][0])
    d, ph = fm_demod_arctan(g["edge_i"], g["edge_q"], float(g["edge_prev"][0]))
    np.testing.assert_array_equal(d, g["edge_demod"])
    assert ph == float(g["edge_phase"][0])
