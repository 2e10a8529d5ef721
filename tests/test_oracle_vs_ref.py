"""Pin oracle/fm_oracle.c bit-for-bit against the reference's own compiled
sources (oracle/_ref/libfmref.so = /root/reference/src/filter.cpp + iofunc.cpp,
built by oracle/Makefile).  Skipped where _ref is absent."""
import numpy as np
import pytest

pytestmark = pytest.mark.ref


def bits_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    np.testing.assert_array_equal(a, b)


LPF_CASES = [(2.4e6, 100e3, 13), (2.4e6, 100e3, 101), (2.4e6, 100e3, 151), (1.44e6, 100e3, 101),
             (960e3, 100e3, 101), (240e3, 16e3, 13), (240e3, 16e3, 101), (288e3, 16e3, 101), (320e3, 16e3, 101),
             (240e3 * 147, 16e3, 14847), (320e3 * 441, 16e3, 44541), (240e3 * 147, 16e3, 13 * 147)]


@pytest.mark.parametrize("Fs,Fc,T", LPF_CASES)
def test_lpf(oracle, ref, Fs, Fc, T):
    bits_equal(oracle.impulse_response_lpf(Fs, Fc, T), ref.impulse_response_lpf(Fs, Fc, T))


@pytest.mark.parametrize("T", [13, 101, 151])
@pytest.mark.parametrize("Fs", [240e3, 288e3, 320e3])
def test_bpf(oracle, ref, Fs, T):
    bits_equal(oracle.band_pass(Fs, 18.5e3, 19.5e3, T), ref.band_pass(Fs, 18.5e3, 19.5e3, T))
    bits_equal(oracle.band_pass(Fs, 22e3, 54e3, T), ref.band_pass(Fs, 22e3, 54e3, T))


@pytest.fixture(scope="module")
def sig():
    rng = np.random.default_rng(1)
    return rng.standard_normal(6000).astype(np.float32), rng.standard_normal(150).astype(np.float32)


def test_convolve_fir(oracle, ref, sig):
    x, _ = sig
    h = oracle.impulse_response_lpf(240e3, 16e3, 101)
    bits_equal(oracle.convolve_fir(x[:700], h), ref.convolve_fir(x[:700], h))
    bits_equal(oracle.convolve_fir(x[:20], h), ref.convolve_fir(x[:20], h))  # shorter than taps


@pytest.mark.parametrize("T", [13, 101, 151])
def test_block_fir(oracle, ref, sig, T):
    x, st = sig
    h = oracle.impulse_response_lpf(240e3, 16e3, T)
    a, b = oracle.convolve_block_fir(x, h, st[:T - 1]), ref.convolve_block_fir(x, h, st[:T - 1])
    bits_equal(a[0], b[0]); bits_equal(a[1], b[1])


@pytest.mark.parametrize("D", [1, 3, 5, 6, 10])
@pytest.mark.parametrize("T", [13, 101, 151])
def test_fast_fir(oracle, ref, sig, T, D):
    x, st = sig
    x = x[: len(x) // D * D]
    h = oracle.impulse_response_lpf(2.4e6, 100e3, T)
    a, b = oracle.convolve_block_fast_fir(x, h, st[:T - 1], D), ref.convolve_block_fast_fir(x, h, st[:T - 1], D)
    bits_equal(a[0], b[0]); bits_equal(a[1], b[1])


@pytest.mark.parametrize("U,D,n", [(4, 3, 150), (147, 800, 5600), (441, 3200, 3200), (24, 125, 2500), (4, 25, 5000)])
def test_resample_fir(oracle, ref, sig, U, D, n):
    x, _ = sig
    rng = np.random.default_rng(2)
    h = oracle.impulse_response_lpf(240e3 * U, 16e3, 101 * U)
    st = np.zeros(101 * U - 1, np.float32)
    st[U - 1::U] = rng.standard_normal(len(st[U - 1::U])).astype(np.float32)
    a = oracle.convolve_block_resample_fir(x[:n], h, st, D, U)
    b = ref.convolve_block_resample_fir(x[:n], h, st, D, U)
    bits_equal(a[0], b[0]); bits_equal(a[1], b[1])


def test_demod(oracle, ref):
    rng = np.random.default_rng(3)
    I, Q = rng.standard_normal(1000).astype(np.float32), rng.standard_normal(1000).astype(np.float32)
    I[5] = Q[5] = 0  # den == 0 branch
    I[0] = Q[0] = 0
    a, b = oracle.fm_demod(I, Q, 0.3, -0.2), ref.fm_demod(I, Q, 0.3, -0.2)
    bits_equal(a[0], b[0])
    assert a[1:] == b[1:]


def test_allpass(oracle, ref, sig):
    x, st = sig
    for d in (6, 50, 75):
        a, b = oracle.all_pass(x, st[:d]), ref.all_pass(x, st[:d])
        bits_equal(a[0], b[0]); bits_equal(a[1], b[1])


def test_pll(oracle, ref):
    t = np.arange(6000)
    pilot = (0.1 * np.cos(2 * np.pi * 19e3 * t / 240e3 + 0.7)).astype(np.float32)
    s = np.array([0, 0, 1, 0, 1, 0], np.float32)
    for blk in np.split(pilot, 3):  # state carry
        a, b = oracle.fm_pll(blk, s, 19e3, 240e3), ref.fm_pll(blk, s, 19e3, 240e3)
        bits_equal(a[0], b[0]); bits_equal(a[1], b[1])
        s = a[1]


def test_up_down(oracle, ref, sig):
    x, _ = sig
    bits_equal(oracle.upsample(x[:50], 7), ref.upsample(x[:50], 7))
    bits_equal(oracle.downsample(x[:503], 7), ref.downsample(x[:503], 7))


def test_u8_and_pcm(oracle, ref):
    rng = np.random.default_rng(4)
    raw = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, 4096, dtype=np.uint8)])
    bits_equal(oracle.u8_to_f32(raw), ref.u8_to_f32(raw))
    au = (rng.standard_normal(4096) * 3).astype(np.float32)
    au[3], au[4], au[5], au[6], au[7] = np.nan, 1e9, -1e12, np.inf, -np.inf
    bits_equal(oracle.pcm16(au, wrap=True), ref.pcm16(au))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("taps", [(101, 101, 101), (151, 101, 151), (13, 13, 13)])
def test_pipeline(oracle, ref, mode, channels, taps):
    if mode == 3 and taps[1] == 101 and channels == 2:
        pass
    p = oracle.mode_params(mode, *taps)
    nblk = 3
    iq = oracle.synth_fm_u8(p.block_bytes // 2 * nblk, rf_Fs=p.rf_Fs, seed=7 + mode)
    po, pr = oracle.pipeline(mode, channels, *taps), ref.pipeline(mode, channels, *taps)
    for b in range(nblk):
        blk = iq[b * p.block_bytes:(b + 1) * p.block_bytes]
        a, c = po.process(blk), pr.process(blk)
        for k in a:
            bits_equal(a[k], c[k])
        if channels == 2:
            for k in ("carrier_filt", "stereo_filt", "pll", "mixer", "allpass", "mono_filt", "stereo_final"):
                bits_equal(po.intermediate(k), pr.intermediate(k))


def test_estimate_psd(oracle, ref):
    """the Bartlett PSD the reference's authors validated their stages with."""
    rng = np.random.default_rng(9)
    for n in (512, 5000, 2048):
        x = (0.3 * np.cos(2 * np.pi * 5e3 * np.arange(n) / 48e3) + 0.05 * rng.standard_normal(n)).astype(np.float32)
        a, b = oracle.estimate_psd(x, 48e3), ref.estimate_psd(x, 48e3)
        bits_equal(a[0], b[0]); bits_equal(a[1], b[1])


def test_restatement_is_a_fair_cpu_baseline(oracle, ref, capsys):
    """the oracle is what bench.py times as `cpu_baseline` on the GPU box (the reference cannot travel), so
    it must not be SLOWER than the compiled reference (that would flatter the GPU): mode-0 mono chain, 40 reference
    blocks, best of 5, C entry points only.  Measured here: the reference's own src/filter.cpp + iofunc.cpp (-O3, its
    flags) takes 1.25 .. 1.45 x the oracle's time -- its std::vector clear/resize zero-fills, back_inserter growth and
    per-block copies (src/project.cpp:98-105, src/filter.cpp:163) are work the restatement does not do -- so the
    `cpu_baseline` figure bench.py reports is generous to the CPU by that factor."""
    import time
    p = oracle.mode_params(0, 101, 101, 101)
    nblk = 40
    iq = oracle.synth_fm_u8(p.block_bytes // 2 * nblk, rf_Fs=p.rf_Fs, seed=3)

    import ctypes as C
    blocks = [np.ascontiguousarray(iq[b * p.block_bytes:(b + 1) * p.block_bytes]) for b in range(nblk)]
    bufs = [np.zeros(5120, np.float32) for _ in range(3)] + [np.zeros(1024, np.float32)]
    ptr = [a.ctypes.data_as(C.c_void_p) for a in bufs]

    def best(make, call):            # the C entry points only: no Python-side copies of intermediates
        t = []
        for _ in range(5):
            pl = make()
            t0 = time.perf_counter()
            for blk in blocks:
                call(pl, blk)
            t.append(time.perf_counter() - t0)
        return min(t)

    t_o = best(lambda: oracle.pipeline(0, 1), lambda pl, blk: oracle.lib.fmo_pipeline_process(pl.h, blk, len(blk), ptr[0], ptr[1], ptr[2], ptr[3], None))
    t_r = best(lambda: ref.pipeline(0, 1), lambda pl, blk: ref.lib.ref_pipeline_process(pl.h, blk, len(blk)))
    with capsys.disabled():
        print(f"\nmode-0 mono, {nblk * 51200} samples: oracle {nblk * 51200 / t_o / 1e6:.1f} MS/s, compiled reference "
              f"{nblk * 51200 / t_r / 1e6:.1f} MS/s, ratio {t_r / t_o:.3f}")
    assert 0.9 <= t_r / t_o <= 1.8, (t_o, t_r)
