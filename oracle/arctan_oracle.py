"""CPU restatement of the arctangent demodulator of the reference's Python model -- TEST INFRASTRUCTURE ONLY.

fmDemodArctan, /root/reference/model/fmSupportLib.py:502-531 (the C++ receiver uses the discriminator fmDemod,
src/filter.cpp:248-266; BASELINE.json names "arctan/PLL demod", so the model's variant is built as an option).
Pinned to tests/golden/arctan.npz, which tests/golden/make_golden_arctan.py produced by importing the reference's own
fmSupportLib.py in the build container.  Only tests/ may import this module; the product never does.
"""
import math

import numpy as np

PI = math.pi


def unwrap_pair(prev, cur):
    """np.unwrap([prev, cur])[1] (numpy/lib/function_base.py, period 2 pi, discont pi), restated for one pair."""
    dd = cur - prev
    ddmod = math.fmod(dd + PI, 2 * PI)
    if ddmod < 0:
        ddmod += 2 * PI                      # np.mod: sign of the divisor
    ddmod -= PI
    if ddmod == -PI and dd > 0:
        ddmod = PI
    corr = 0.0 if abs(dd) < PI else ddmod - dd
    return cur + corr


def fm_demod_arctan(I, Q, prev_phase=0.0):
    """The model's loop, sample by sample (fmSupportLib.py:510-527): float64, the running phase stays unwrapped."""
    out = np.empty(len(I))
    for k in range(len(I)):
        cur = unwrap_pair(prev_phase, math.atan2(Q[k], I[k]))
        out[k] = cur - prev_phase
        prev_phase = cur
    return out, prev_phase
