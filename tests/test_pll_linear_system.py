"""The algorithm behind the parallel PLL's lane starts (csrc/kernels_pll.hip: pll_lti_chunks_kernel, lti_start_state), checked
in NumPy without a GPU: while the loop is locked, the reference's recurrence (src/filter.cpp:32-80) with the closed-form phase
detector -- atan2f(in * -sin t, in * cos t) = -t, turned by pi for in < 0 -- is a LINEAR time-invariant system driven by a
staircase that climbs half a turn at every sign change of the input.  The test walks the non-linear recurrence in float64
(no float32 grid: that is what the lanes' 64 true steps are for) and reproduces its state at every 64th sample from the signs
alone, with the kernels' own decomposition: per-chunk zero-state responses, prefix sums of the climbs, Horner over the 20 chunks
behind with A^64."""
import numpy as np

KP, KI = 0.01 * 2.666, 0.01 * 0.01 * 3.555          # src/filter.cpp:41-44 (normBandwidth 0.01)
F = 19e3 / 240e3                                     # revolutions per IF sample
CHUNK, TERMS = 64, 20


def walk_nonlinear(x, phi0, iota0, off0):
    """(phase, integrator) / 2 pi in front of every sample: the recurrence with the closed-form detector."""
    phi, iota, off = phi0, iota0, off0
    out = np.empty((len(x) + 1, 2))
    out[0] = phi, iota
    for k, v in enumerate(x):
        th = F * off + phi
        fr = th - np.rint(th)                        # trigArg / 2 pi reduced to [-0.5, 0.5]
        e = -fr if v > 0 else (0.5 - fr if fr >= 0 else -0.5 - fr)
        iota = iota + KI * e
        phi = phi + KP * e + iota
        off += 1
        out[k + 1] = phi, iota
    return out


def lti_starts(x, phi0, iota0, off0):
    """State in front of sample 64 i from the signs alone, as the kernels compute it."""
    A = np.array([[1 - KP - KI, 1.0], [-KI, 1.0]])
    B = np.array([KP + KI, KI])
    n = len(x)
    nchunk = n // CHUNK
    pos = x > 0
    chg = np.zeros(n + 1)
    chg[1:n] = (pos[1:] != pos[:-1]) * 0.5          # climb INTO sample k
    rel = np.zeros((nchunk, 2))
    climb = np.zeros(nchunk)                         # sign changes inside chunk i and into the next chunk's first sample
    for i in range(nchunk):
        s, dT = np.zeros(2), 0.0
        for j in range(CHUNK):
            s = A @ s + B * (dT - F * j)
            dT += chg[i * CHUNK + j + 1]
        rel[i], climb[i] = s, dT
    before = np.concatenate([[0.0], np.cumsum(climb)[:-1]])
    th0 = F * off0 + phi0
    T0 = np.rint(th0) if pos[0] else np.rint(th0 - 0.5) + 0.5
    G, Q = np.zeros(2), np.eye(2)
    for _ in range(CHUNK):
        G = A @ G + B
        Q = A @ Q
    R = rel + ((T0 + before) - F * (off0 + CHUNK * np.arange(nchunk)) - phi0)[:, None] * G
    out = np.empty((nchunk, 2))
    for i in range(nchunk):
        s = np.zeros(2)
        for m in range(min(i, TERMS), 0, -1):
            if m == i:
                s = np.array([0.0, iota0])           # the block's true initial state, in deviations from (phi0, 0)
            s = Q @ s + R[i - m]
        if i == 0:
            s = np.array([0.0, iota0])
        out[i] = phi0 + s[0], s[1]
    return out


def test_locked_loop_is_linear_in_the_signs_of_its_input():
    rng = np.random.default_rng(3)
    n = 64 * 400
    k = np.arange(n)
    # pilot on / off frequency, clean / with noise that moves its zero crossings but adds none (the band-passed pilot the PLL sees)
    for df, noise in ((0.0, 0.0), (2.3e-6, 0.0), (-1.1e-6, 0.01)):
        x = np.cos(2 * np.pi * ((F + df) * k + 0.137)) + noise * rng.standard_normal(n)
        # start locked: walk 4000 samples of the same pilot first
        pre = np.cos(2 * np.pi * ((F + df) * (np.arange(4000) - 4000) + 0.137))
        st = walk_nonlinear(pre, 0.0, 0.0, 0.0)[-1]
        off0 = 4000.0 + 1.0e6                                          # late in a stream: large T, large f * off
        phi0 = st[0] - F * 1.0e6 + np.rint(F * 1.0e6)                  # same loop state, offset moved by whole turns
        ref = walk_nonlinear(x, phi0, st[1], off0)
        got = lti_starts(x, phi0, st[1], off0)
        err = np.abs(got - ref[::CHUNK][: len(got)])
        assert err[:, 0].max() < 2e-7 and err[:, 1].max() < 2e-9, (df, noise, err.max(axis=0))


def test_a_glitch_moves_the_staircase_by_a_whole_turn():
    """Two extra sign changes within a sample or two (what heavy noise does at a zero crossing) put the staircase one turn
    ahead: the linear system's phase ends one revolution off -- the same loop state modulo 2 pi, which the merge accepts
    (pll_phase_dist) -- not somewhere else."""
    n = 64 * 200
    k = np.arange(n)
    x = np.cos(2 * np.pi * (F * k + 0.05))
    st = walk_nonlinear(np.cos(2 * np.pi * (F * (np.arange(4000) - 4000) + 0.05)), 0.0, 0.0, 0.0)[-1]
    z = int(np.flatnonzero((x[1:] > 0) != (x[:-1] > 0))[40]) + 1    # a zero crossing; force the sample behind it back
    y = x.copy()
    y[z + 1] = -x[z + 1]
    ref = walk_nonlinear(y, st[0], st[1], 4000.0)
    got = lti_starts(y, st[0], st[1], 4000.0)
    d = got[-1, 0] - ref[::CHUNK][len(got) - 1, 0]
    assert abs(d - np.rint(d)) < 1e-3 and abs(np.rint(d)) == 1.0, d
