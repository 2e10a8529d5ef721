"""Oracle (oracle/fm_oracle.c) vs the committed golden vectors in
tests/golden/ (generated from the compiled reference by make_golden.py).
Bit-exact: the oracle restates the reference's float32 evaluation order.
Runs anywhere (no GPU, no /root/reference)."""
import hashlib
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HT = 256


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def bits_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    np.testing.assert_array_equal(a, b)


def ht(a):
    return a if len(a) <= 2 * HT else np.concatenate([a[:HT], a[-HT:]])


def test_coefficients(oracle):
    g = load("coeffs.npz")
    assert len(g.files) == 34
    for k in g.files:
        parts = k.split("_")
        if parts[0] == "lpf":
            got = oracle.impulse_response_lpf(float(parts[1]), float(parts[2]), int(parts[3]))
        else:
            got = oracle.band_pass(float(parts[1]), float(parts[2]), float(parts[3]), int(parts[4]))
        bits_equal(got, g[k])


def test_real_signal_block(oracle):
    iq = np.fromfile(os.path.join(G, "pipe_iq_102400.u8"), np.uint8)
    g = load("pipe_mode0.npz")
    assert hashlib.sha256(iq.tobytes()).digest() == g["iq_sha256"].tobytes()
    for rf_t, au_t in [(101, 101), (151, 101), (13, 13)]:
        out = oracle.pipeline(0, 1, rf_t, au_t, 101).process(iq)
        tag = f"t{rf_t}_{au_t}"
        for k in ("if_i", "if_q", "demod", "audio"):
            bits_equal(out[k], g[f"{tag}_{k}"])
        # the capture is noisy: s16 overflows and WRAPS in the compiled reference
        bits_equal(oracle.pcm16(out["audio"], wrap=True), g[f"{tag}_s16"])
    assert (np.abs(g["t101_101_audio"]) * 16384 > 32767).sum() > 0  # the wrap case is exercised


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("ch", [1, 2])
def test_synth_streams(oracle, mode, ch):
    iq = load("synth_inputs.npz")[f"mode{mode}"]
    g = load(f"synth_mode{mode}_ch{ch}.npz")
    bb, nblk = int(g["block_bytes"][0]), int(g["nblk"][0])
    p = oracle.mode_params(mode)
    assert p.block_bytes == bb
    # the stored input is what the oracle's generator produces today
    np.testing.assert_array_equal(oracle.synth_fm_u8(bb // 2 * nblk, rf_Fs=p.rf_Fs, seed=0x3D74 + mode), iq)
    pl = oracle.pipeline(mode, ch)
    for b in range(nblk):
        out = pl.process(iq[b * bb:(b + 1) * bb])
        bits_equal(out["audio_l"], g[f"b{b}_audio_l"])
        bits_equal(ht(out["if_i"]), g[f"b{b}_if_i_ht"])
        bits_equal(ht(out["if_q"]), g[f"b{b}_if_q_ht"])
        gd = g[f"b{b}_demod"]
        bits_equal(out["demod"] if len(gd) == len(out["demod"]) else ht(out["demod"]), gd)
        if ch == 2:
            bits_equal(out["audio_r"], g[f"b{b}_audio_r"])
            for k in ("carrier_filt", "stereo_filt", "pll", "mixer", "allpass", "mono_filt", "stereo_final"):
                bits_equal(ht(pl.intermediate(k)), g[f"b{b}_{k}_ht"])


@pytest.mark.parametrize("taps", [(151, 101), (13, 13)])
def test_synth_other_taps(oracle, taps):
    iq = load("synth_inputs.npz")["mode0"]
    g = load(f"synth_mode0_t{taps[0]}_{taps[1]}.npz")
    pl = oracle.pipeline(0, 1, taps[0], taps[1], 101)
    for b in range(2):
        out = pl.process(iq[b * 102400:(b + 1) * 102400])
        bits_equal(out["audio"], g[f"b{b}_audio"])
        bits_equal(ht(out["demod"]), g[f"b{b}_demod_ht"])


def test_function_level(oracle):
    g = load("functions.npz")
    x = g["x"]
    for U, D, n in [(4, 3, 150), (24, 125, 2500), (4, 25, 5000), (147, 800, 5600), (441, 3200, 3200)]:
        h = oracle.impulse_response_lpf(240e3 * U, 16e3, 101 * U)
        st = np.zeros(101 * U - 1, np.float32)
        st[U - 1::U] = x[-100:]
        y, st2 = oracle.convolve_block_resample_fir(x[:n], h, st, D, U)
        bits_equal(y, g[f"rs_{U}_{D}_{n}_y"])
        bits_equal(st2[U - 1::U], g[f"rs_{U}_{D}_{n}_state_used"])
    for T, D in [(101, 10), (101, 5), (101, 6), (101, 3), (151, 10), (13, 10), (101, 1), (13, 1), (7, 2)]:
        h = oracle.impulse_response_lpf(2.4e6, 100e3, T)
        n = 6000 // D * D
        y, st2 = oracle.convolve_block_fast_fir(x[:n], h, x[-(T - 1):], D)
        bits_equal(y, g[f"ff_{T}_{D}_y"])
        bits_equal(st2, g[f"ff_{T}_{D}_state"])
        if D == 1:  # property 1: FastFIR(D=1) == BlockFIR
            y1, s1 = oracle.convolve_block_fir(x[:n], h, x[-(T - 1):])
            bits_equal(y1, y); bits_equal(s1, st2)
    h = oracle.impulse_response_lpf(240e3, 16e3, 101)
    bits_equal(oracle.convolve_fir(x[:700], h), g["cf_101_y"])
    bits_equal(oracle.convolve_fir(x[:20], h), g["cf_short_y"])
    y, st2 = oracle.convolve_block_fast_fir(x[:100], h, np.zeros(100, np.float32), 5)
    bits_equal(y, g["ff_minblock_y"]); bits_equal(st2, g["ff_minblock_state"])


def test_edges(oracle):
    g = load("edge.npz")
    out = oracle.pipeline(0, 1).process(np.full(102400, 128, np.uint8))
    bits_equal(out["demod"], g["zero_demod"]); bits_equal(out["audio"], g["zero_audio"])
    assert not out["demod"].any()
    d, pi, pq = oracle.fm_demod(g["demod_I"], g["demod_Q"], 0.25, -0.5)
    bits_equal(d, g["demod_out"]); bits_equal(np.array([pi, pq], np.float32), g["demod_prev"])
    bits_equal(oracle.pcm16(g["pcm_in"], wrap=True), g["pcm_s16_wrap"])
    sat = oracle.pcm16(g["pcm_in"], wrap=False)
    assert sat[0] == 0 and sat[1] == 32767 and sat[2] == -32768 and sat[3] == 32767 and sat[4] == -32768
    bits_equal(oracle.u8_to_f32(np.arange(256, dtype=np.uint8)), g["u8_all"])
    st = np.array([0, 0, 1, 0, 1, 0], np.float32)
    outs = []
    for blk in np.split(g["pll_in"], 3):
        y, st = oracle.fm_pll(blk, st, 19e3, 240e3)
        outs.append(y)
    bits_equal(np.concatenate(outs), g["pll_out"]); bits_equal(st, g["pll_state"])
    x = load("functions.npz")["x"]
    ap, aps = oracle.all_pass(x[:500], x[1000:1050])
    bits_equal(ap, g["allpass_out"]); bits_equal(aps, g["allpass_state"])


def test_process_contract_binary(oracle):
    """u8 stdin -> s16 stdout of the reference's threadMonoOnly binary
    (151/101 taps), prefix comparison (its EOF truncation is nondeterministic)."""
    g = load("tmo_mode0.npz")
    nb = int(g["nblk"][0])
    iq = oracle.synth_fm_u8(51200 * nb, rf_Fs=2.4e6, seed=int(g["seed"][0]))
    assert hashlib.sha256(iq.tobytes()).digest() == g["iq_sha256"].tobytes()
    pl = oracle.pipeline(0, 1, 151, 101, 101)
    s16 = np.concatenate([oracle.pcm16(pl.process(iq[b * 102400:(b + 1) * 102400])["audio"]) for b in range(nb)])
    n = len(g["s16"])
    assert n >= 1024
    np.testing.assert_array_equal(s16[:n], g["s16"])


def test_block_split_invariance(oracle):
    """SURVEY section 4 property 2: streaming in blocks == one long block (bit-exact on CPU)."""
    iq = load("synth_inputs.npz")["mode0"]
    whole = oracle.pipeline(0, 1).process(iq)["audio"]
    pl = oracle.pipeline(0, 1)
    parts = [pl.process(iq[o:o + 25600])["audio"] for o in range(0, len(iq), 25600)]
    bits_equal(np.concatenate(parts), whole)


def test_resampler_equals_up_fir_down(oracle):
    """SURVEY section 4 property 3: resampler == downsample(convolveFIR(upsample(x))) * (1+U)."""
    rng = np.random.default_rng(5)
    U, D, n = 4, 3, 300
    x = rng.standard_normal(n).astype(np.float32)
    h = oracle.impulse_response_lpf(240e3 * U, 16e3, 101 * U)
    y, _ = oracle.convolve_block_resample_fir(x, h, np.zeros(101 * U - 1, np.float32), D, U)
    full = oracle.convolve_fir(oracle.upsample(x, U), h)[: n * U]
    ref = oracle.downsample(full, D).astype(np.float64) * (1 + U)
    np.testing.assert_allclose(y, ref[: len(y)], rtol=2e-5, atol=2e-6)


def test_psd_golden(oracle):
    g = load("psd.npz")
    for k in ("audio", "tone"):
        f, p = oracle.estimate_psd(g[f"{k}_in"], 48e3)
        bits_equal(f, g[f"{k}_freq"]); bits_equal(p, g[f"{k}_psd"])
    assert int(np.argmax(g["tone_psd"])) == 32          # 3 kHz at 48 kHz / 512 per bin
    assert np.all(np.diff(g["tone_freq"]) == np.float32(93.75))


def test_long_stereo_stream(oracle):
    """100 reference blocks (2.13 s) of mode-0 stereo through the oracle, block by block, against the
    compiled reference's output: snippets at the checkpoints and SHA-256 of the WHOLE left / right / NCO
    streams -- the pilot PLL recurrence (glibc sinf/cosf/atan2f inside) included, bit for bit."""
    g = load("stereo_long_mode0.npz")
    nblk, every = int(g["nblk"][0]), int(g["every"][0])
    p = oracle.mode_params(0, 101, 101, 101)
    iq = oracle.synth_fm_u8(p.block_bytes // 2 * nblk, rf_Fs=p.rf_Fs, seed=int(g["seed"][0]))
    assert hashlib.sha256(iq.tobytes()).digest() == g["iq_sha256"].tobytes()
    po = oracle.pipeline(0, 2)
    L, R, P = [], [], []
    for b in range(nblk):
        out = po.process(iq[b * p.block_bytes:(b + 1) * p.block_bytes])
        L.append(out["audio_l"]); R.append(out["audio_r"]); P.append(po.intermediate("pll")[1:])
        if b % every == 0:
            bits_equal(out["audio_l"][:256], g[f"b{b}_audio_l"]); bits_equal(out["audio_r"][:256], g[f"b{b}_audio_r"])
            bits_equal(po.intermediate("pll")[:257], g[f"b{b}_pll"])
    for k, v in (("audio_l", L), ("audio_r", R), ("pll", P)):
        assert hashlib.sha256(np.concatenate(v).tobytes()).digest() == g[f"{k}_sha256"].tobytes(), k


def spec_mode_params(oracle, U, D):
    """BASELINE configs[2] (doc/3dy4-project-2022.pdf p.3): 2.5 MS/s -> 250 kS/s -> 48 kS/s (24/125) or 40 kS/s (4/25)."""
    p = oracle.mode_params(2, 101, 101, 101)
    p.rf_Fs, p.if_Fs, p.rf_decim = 2500000, 250000, 10
    p.audio_upsamp, p.audio_decim = U, D
    p.audio_Fs = 250000.0 * U / D
    p.audio_taps = 101 * U
    p.block_bytes = 2 * p.rf_decim * 5000
    return p


@pytest.mark.parametrize("U,D", [(4, 25), (24, 125)])
@pytest.mark.parametrize("ch", [1, 2])
def test_spec_mode_pipeline(oracle, U, D, ch):
    """configs[2] as a whole pipeline: the reference's graph at the course spec's 2.5 MS/s parameters."""
    g = load(f"spec_mode_{U}_{D}_ch{ch}.npz")
    sp = spec_mode_params(oracle, U, D)
    assert int(g["block_bytes"][0]) == sp.block_bytes
    iq = oracle.synth_fm_u8(sp.block_bytes // 2 * 3, rf_Fs=sp.rf_Fs, seed=int(g["seed"][0]))
    po = oracle.pipeline_params(sp, ch)
    for b in range(3):
        out = po.process(iq[b * sp.block_bytes:(b + 1) * sp.block_bytes])
        bits_equal(out["audio_l"], g[f"b{b}_audio_l"])
        assert len(out["audio_l"]) == 5000 * U // D
        bits_equal(ht(out["demod"]), g[f"b{b}_demod_ht"])
        if ch == 2:
            bits_equal(out["audio_r"], g[f"b{b}_audio_r"])
            bits_equal(ht(po.intermediate("pll")), g[f"b{b}_pll_ht"])
