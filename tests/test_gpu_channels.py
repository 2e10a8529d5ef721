"""GPU parity tests of the receiver banks (fmrx_channels_create_ex: N independent receivers per device call,
src/project.cpp:455-468 has one STATES set per receiver), through the C ABI.

Exact banks (exact = 1) promise the compiled reference's audio BIT FOR BIT per channel: every comparison below is on
bit patterns, against the oracle (oracle/fm_oracle.c, itself pinned bit for bit to the compiled reference) and, for the
golden fixture's stream, against the compiled reference's own SHA-256 (tests/golden/stereo_long_mode0.npz).
"""
import hashlib
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits_equal(a, b, msg=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, (a.shape, b.shape, msg)
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    np.testing.assert_array_equal(a, b, err_msg=msg)


def channel_stream(oracle, c, n_samples, rf_Fs):
    """Channel c's synthetic stream: its own dither seed and its own place in the programme (the generator's seed only
    drives the dither; the start offset moves the tones and the pilot's phase)."""
    return oracle.synth_fm_u8(n_samples, rf_Fs=rf_Fs, seed=0x3D74 + c, start=0 if c == 0 else 7919 * c)


@pytest.mark.parametrize("mode,taps", [(0, (101, 101, 101)), (1, (101, 101, 101)), (0, (151, 101, 151)), (0, (13, 13, 13)),
                                       (2, (101, 101, 101)), (3, (101, 101, 101)), (2, (13, 13, 13))])
def test_stereo_bank_exact(fmrx, oracle, mode, taps):
    """Five stereo receivers, four reference-size blocks each, every intermediate of every channel against the oracle
    streaming that channel alone: discriminator output, both band-pass outputs, NCO, left, right -- all bit for bit;
    PCM equal.  Then one channel is reset (a new stream starts there) while the others carry on."""
    p = oracle.mode_params(mode, *taps)
    N, nblk, bb = 5, 4, p.block_bytes
    streams = [channel_stream(oracle, c, bb // 2 * (nblk + 2), p.rf_Fs) for c in range(N)]
    ch = fmrx.Channels(mode, N, rf_taps=taps[0], base_audio_taps=taps[1], stereo_taps=taps[2], audio_channels=2, exact=True)
    refs = [oracle.pipeline(mode, 2, *taps) for _ in range(N)]
    assert ch.n_audio == {0: 1024, 1: 1024, 2: 1029, 3: 3087}[mode]     # the reference's blocks: src/project.cpp:55-57

    def check(b, channels):
        iq = np.stack([st[b * bb:(b + 1) * bb] for st in streams])
        out = ch.process(iq)
        for c in channels:
            want = refs[c].process(streams[c][b * bb:(b + 1) * bb])
            tag = f"mode {mode} taps {taps} channel {c} block {b}"
            bits_equal(ch.read_tap(c, "demod"), want["demod"], "demod " + tag)
            bits_equal(ch.read_tap(c, "carrier_filt"), refs[c].intermediate("carrier_filt"), "pilot band-pass " + tag)
            bits_equal(ch.read_tap(c, "stereo_filt"), refs[c].intermediate("stereo_filt"), "22-54 kHz band-pass " + tag)
            bits_equal(ch.read_tap(c, "pll"), refs[c].intermediate("pll"), "NCO " + tag)
            bits_equal(out["audio_l"][c], want["audio_l"], "left " + tag)
            bits_equal(out["audio_r"][c], want["audio_r"], "right " + tag)
            bits_equal(out["pcm16"][c, :, 0], oracle.pcm16(want["audio_l"]), "pcm left " + tag)
            bits_equal(out["pcm16"][c, :, 1], oracle.pcm16(want["audio_r"]), "pcm right " + tag)

    for b in range(nblk):
        check(b, range(N))
    ch.reset(3)
    refs[3] = oracle.pipeline(mode, 2, *taps)
    for b in range(nblk, nblk + 2):
        check(b, range(N))


def test_mono_bank_exact(fmrx, oracle):
    """The mono bank in the reference's evaluation order: audio bit for bit (the specialised mono bank promises 2e-6)."""
    for mode in (0, 1, 2, 3):
        p = oracle.mode_params(mode, 101, 101, 101)
        N, nblk, bb = 3, 3, p.block_bytes
        streams = [channel_stream(oracle, c, bb // 2 * nblk, p.rf_Fs) for c in range(N)]
        ch = fmrx.Channels(mode, N, exact=True)
        refs = [oracle.pipeline(mode, 1) for _ in range(N)]
        for b in range(nblk):
            out = ch.process(np.stack([st[b * bb:(b + 1) * bb] for st in streams]))
            for c in range(N):
                want = refs[c].process(streams[c][b * bb:(b + 1) * bb])["audio"]
                bits_equal(out["audio"][c], want, f"mode {mode} channel {c} block {b}")
                bits_equal(out["pcm16"][c], oracle.pcm16(want))


def test_bank_resampler_channel_groups(fmrx, oracle):
    """The lane-per-channel resampler works on groups of 64 channels (8 lanes share a channel's loads; spare lanes repeat the last
    channel): a bank of 65 receivers -- one full group and one with a single member -- in mode 2, stereo and mono, bit for bit."""
    p = oracle.mode_params(2, 101, 101, 101)
    N, nblk, bb = 65, 2, p.block_bytes
    base = [channel_stream(oracle, c, bb // 2 * nblk, p.rf_Fs) for c in range(3)]
    streams = [base[c % 3] for c in range(N)]
    for st in (2, 1):
        ch = fmrx.Channels(2, N, audio_channels=st, exact=True)
        refs = [oracle.pipeline(2, st) for _ in range(3)]
        for b in range(nblk):
            out = ch.process(np.stack([s_[b * bb:(b + 1) * bb] for s_ in streams]), want_pcm=False)
            want = [refs[c].process(base[c][b * bb:(b + 1) * bb]) for c in range(3)]
            for c in (0, 1, 2, 62, 63, 64):
                if st == 2:
                    bits_equal(out["audio_l"][c], want[c % 3]["audio_l"], f"stereo channel {c} block {b}")
                    bits_equal(out["audio_r"][c], want[c % 3]["audio_r"], f"stereo channel {c} block {b}")
                else:
                    bits_equal(out["audio"][c], want[c % 3]["audio"], f"mono channel {c} block {b}")
        ch.close()


def test_bank_other_block_sizes(fmrx, oracle):
    """Blocks that are not the reference's size: ragged against the kernels' tiles (504 IF outputs per wave, 2048 per
    band-pass workgroup, 512 audio outputs per workgroup), several times the reference's, and the smallest the bank
    accepts; the oracle consumes the same stream in the same cuts."""
    p = oracle.mode_params(0, 101, 101, 101)
    for bb in (102400 * 3, 2 * 10 * 5 * 8 * 63, 8000):
        N = 3
        streams = [channel_stream(oracle, 10 + c, bb // 2 * 3, p.rf_Fs) for c in range(N)]
        ch = fmrx.Channels(0, N, audio_channels=2, exact=True, block_bytes=bb)
        refs = [oracle.pipeline(0, 2) for _ in range(N)]
        for b in range(3):
            out = ch.process(np.stack([st[b * bb:(b + 1) * bb] for st in streams]))
            for c in range(N):
                want = refs[c].process(streams[c][b * bb:(b + 1) * bb])
                bits_equal(out["audio_l"][c], want["audio_l"], f"block_bytes {bb} channel {c} block {b}")
                bits_equal(out["audio_r"][c], want["audio_r"])
    with pytest.raises(fmrx.FmrxError):
        fmrx.Channels(0, 4, audio_channels=2, exact=True, block_bytes=1600)      # shorter than the history a channel carries
    with pytest.raises(fmrx.FmrxError):
        fmrx.Channels(2, 4, audio_channels=2, exact=True, block_bytes=16160)     # 808 IF samples: a block must end on an output boundary (n_if * U % D == 0)


@pytest.mark.parametrize("fused", [0, 1], ids=["two kernels", "front end + band-pass pair in one kernel"])
def test_stereo_bank_fast_error_envelope(fmrx, oracle, fused):
    """The FAST stereo bank (exact = 0: matrix-core front end, one fma per tap in the band-pass pair and the audio FIRs, the
    PLL's fast recurrence walked by one lane per channel) promises what the default single-stream path promises
    (tests/test_gpu_parity.py: ENVELOPE_FACTOR): per channel and 0.1 s window, audio RMS error <= max(1e-4, 0.06
    ulp(trigArg(t))) against the oracle, <= 1e-4 in the first window, mono sum (L+R)/2 <= 2e-6 throughout -- checked for
    each of 24 receivers with distinct signals over 2.13 s, one call per reference block."""
    from test_gpu_parity import ENVELOPE_FACTOR, stereo_error_envelope, trig_arg_ulp
    p = oracle.mode_params(0, 101, 101, 101)
    N, nblk, bb = 24, 100, p.block_bytes
    with ProcessPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as ex:
        res = sorted(ex.map(_oracle_channel, [(c, nblk, bb, float(p.rf_Fs)) for c in range(N)], chunksize=2), key=lambda r: r[0])
    fmrx.set_option("bank_fused", fused)                    # a bank copies the process-wide options when it is created
    try:
        ch = fmrx.Channels(0, N, audio_channels=2, exact=False)
    finally:
        fmrx.set_option("bank_fused", 0)
    L = np.zeros((N, nblk * 1024), np.float32)
    R = np.zeros((N, nblk * 1024), np.float32)
    for b in range(nblk):
        out = ch.process(np.stack([r[1][b * bb:(b + 1) * bb] for r in res]), want_pcm=(b == 0))
        L[:, b * 1024:(b + 1) * 1024] = out["audio_l"]
        R[:, b * 1024:(b + 1) * 1024] = out["audio_r"]
        if b == 0:
            # the fast bank's NCO tap (it keeps trigArg; the tap runs the NCO pass on a copy of the row): n_if + 1 values, PLL[0] = the
            # incoming state's lastOut, a 38 kHz cosine after lock
            nco = ch.read_tap(3, "pll")
            assert len(nco) == bb // 2 // p.rf_decim + 1 and nco[0] == 1.0 and np.isfinite(nco).all() and np.abs(nco).max() <= 1.0
            assert 0.4 < float(np.sqrt(np.mean(nco[1000:].astype(np.float64) ** 2))) < 0.9
            # the s16 output is the pack of the float output (what is compared with the reference below)
            for c in range(N):
                bits_equal(out["pcm16"][c, :, 0], oracle.pcm16(out["audio_l"][c]), f"pcm left, channel {c}")
                bits_equal(out["pcm16"][c, :, 1], oracle.pcm16(out["audio_r"][c]), f"pcm right, channel {c}")
    win = 4800
    t_end = (np.arange(nblk * 1024 // win) + 1) * 0.1
    bound = np.maximum(1e-4, ENVELOPE_FACTOR * trig_arg_ulp(t_end))
    worst = 0.0
    for c in range(N):
        for got, want in ((L[c], res[c][2]), (R[c], res[c][3])):
            env = stereo_error_envelope(got, want, win)
            assert (env <= bound).all(), (c, env, bound)
            assert env[0] <= 1e-4
            worst = max(worst, float((env / trig_arg_ulp(t_end)).max()))
        mono = stereo_error_envelope((L[c].astype(np.float64) + R[c]) / 2, (res[c][2].astype(np.float64) + res[c][3]) / 2, win)
        assert mono.max() <= 2e-6, (c, mono.max())
    print(f"fast stereo bank: worst window error over {N} channels = {worst:.3f} ulp(trigArg)")


@pytest.mark.parametrize("mode", [2, 3])
def test_bank_fast_resampling_modes(fmrx, oracle, mode):
    """The fast banks in the resampling modes (44.1 kHz out; src/project.cpp:425-426): matrix-core front end, the band-pass pair
    and the PLL of the fast stereo bank, mixer rows and the batched reference-order resampler (src/filter.cpp:191-223).
    Mono (exact = 0 routes the resampling modes to this bank): audio RMS error <= 1e-4 (measured ~1e-7), s16 equal or +-1 LSB.
    Stereo: the default path's envelope per receiver and 0.1 s window, mono sum <= 2e-6."""
    from test_gpu_parity import ENVELOPE_FACTOR, stereo_error_envelope, trig_arg_ulp
    p = oracle.mode_params(mode, 101, 101, 101)
    bb = p.block_bytes
    # mono
    N, nblk = 5, 3
    streams = [channel_stream(oracle, c, bb // 2 * nblk, p.rf_Fs) for c in range(N)]
    ch = fmrx.Channels(mode, N)                                   # audio_channels = 1, exact = 0
    refs = [oracle.pipeline(mode, 1) for _ in range(N)]
    for b in range(nblk):
        out = ch.process(np.stack([st[b * bb:(b + 1) * bb] for st in streams]))
        for c in range(N):
            want = refs[c].process(streams[c][b * bb:(b + 1) * bb])["audio"]
            err = float(np.sqrt(np.mean((out["audio"][c].astype(np.float64) - want) ** 2)))
            assert err <= 1e-4 and err <= 2e-6, (mode, c, b, err)
            assert np.abs(out["pcm16"][c].astype(np.int32) - oracle.pcm16(want).astype(np.int32)).max() <= 1
    ch.close()
    # stereo: 0.5 s per receiver
    N, nblk = 6, int(0.5 * p.rf_Fs / (bb // 2)) + 1
    with ProcessPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        res = sorted(ex.map(_oracle_channel, [(c, nblk, bb, float(p.rf_Fs), mode) for c in range(N)]), key=lambda r: r[0])
    ch = fmrx.Channels(mode, N, audio_channels=2, exact=False)
    na = ch.n_audio
    L = np.zeros((N, nblk * na), np.float32)
    R = np.zeros((N, nblk * na), np.float32)
    for b in range(nblk):
        out = ch.process(np.stack([r[1][b * bb:(b + 1) * bb] for r in res]), want_pcm=False)
        L[:, b * na:(b + 1) * na] = out["audio_l"]
        R[:, b * na:(b + 1) * na] = out["audio_r"]
    win = 4410
    t_end = (np.arange(nblk * na // win) + 1) * 0.1
    ulp = trig_arg_ulp(t_end, if_Fs=float(p.if_Fs))
    bound = np.maximum(1e-4, ENVELOPE_FACTOR * ulp)
    worst = 0.0
    for c in range(N):
        for got, want in ((L[c], res[c][2]), (R[c], res[c][3])):
            env = stereo_error_envelope(got, want, win)
            assert (env <= bound).all(), (mode, c, env, bound)
            assert env[0] <= 1e-4
            worst = max(worst, float((env / ulp).max()))
        mono = stereo_error_envelope((L[c].astype(np.float64) + R[c]) / 2, (res[c][2].astype(np.float64) + res[c][3]) / 2, win)
        assert mono.max() <= 2e-6, (mode, c, mono.max())
    print(f"fast stereo bank, mode {mode}: worst window error over {N} receivers = {worst:.3f} ulp(trigArg)")


def _oracle_channel(args):
    """Worker: channel c's stream and the oracle's left / right over it (CPU, one process per channel)."""
    c, nblk, bb, rf_Fs = args[:4]
    mode = args[4] if len(args) > 4 else 0
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    from _oracle import Oracle
    o = Oracle()
    iq = channel_stream(o, c, bb // 2 * nblk, rf_Fs)
    pl = o.pipeline(mode, 2)
    L, R = [], []
    for b in range(nblk):
        out = pl.process(iq[b * bb:(b + 1) * bb])
        L.append(out["audio_l"]); R.append(out["audio_r"])
    return c, iq, np.concatenate(L), np.concatenate(R)


def test_stereo_bank_exact_256_channels_two_seconds(fmrx, oracle):
    """256 stereo receivers with distinct signals, 100 reference blocks = 2.13 s of stream each, one bank, one call per
    block period: left and right of EVERY channel equal the oracle's bit for bit over the whole stream -- far beyond
    the 0.13 s after which any re-ordered implementation has left the reference's PLL trajectory.
    Channel 0 carries the stream of tests/golden/stereo_long_mode0.npz: its output must also hash to the COMPILED
    REFERENCE's SHA-256 (nothing from the oracle involved)."""
    g = np.load(os.path.join(G, "stereo_long_mode0.npz"))
    p = oracle.mode_params(0, 101, 101, 101)
    N, nblk, bb = 256, int(g["nblk"][0]), p.block_bytes
    assert int(g["seed"][0]) == 0x3D74 and nblk * bb // 2 / p.rf_Fs > 2.0
    with ProcessPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as ex:
        res = sorted(ex.map(_oracle_channel, [(c, nblk, bb, float(p.rf_Fs)) for c in range(N)], chunksize=4), key=lambda r: r[0])
    assert hashlib.sha256(res[0][1].tobytes()).digest() == g["iq_sha256"].tobytes()
    ch = fmrx.Channels(0, N, audio_channels=2, exact=True)
    L = np.zeros((N, nblk * 1024), np.float32)
    R = np.zeros((N, nblk * 1024), np.float32)
    for b in range(nblk):
        out = ch.process(np.stack([r[1][b * bb:(b + 1) * bb] for r in res]), want_pcm=False)
        L[:, b * 1024:(b + 1) * 1024] = out["audio_l"]
        R[:, b * 1024:(b + 1) * 1024] = out["audio_r"]
    for c in range(N):
        bits_equal(L[c], res[c][2], f"left, channel {c}")
        bits_equal(R[c], res[c][3], f"right, channel {c}")
    assert hashlib.sha256(L[0].tobytes()).digest() == g["audio_l_sha256"].tobytes()
    assert hashlib.sha256(R[0].tobytes()).digest() == g["audio_r_sha256"].tobytes()
    # the signals really are different receivers' (not 256 copies)
    assert len({hashlib.sha256(L[c].tobytes()).digest() for c in range(N)}) == N


def test_eight_pipelines_on_eight_streams(fmrx, oracle):
    """BASELINE configs[4] on the hardware that exists: the eight independent 2.4 MS/s mono channels that an 8-GPU node runs
    one per GPU (seeds 0x3D74 + c, as bench.py's ranks use them), here as eight pipeline handles on eight HIP streams of ONE
    device, their calls interleaved block by block.  Every channel's PCM and float audio must be bit-identical to the same
    handle type running that channel alone on one stream (handles share nothing), and within the mono tolerance of the
    oracle streaming that channel."""
    import torch
    nch, nblk, bb = 8, 2, 2 * 1024000
    iqs = [oracle.synth_fm_u8(bb // 2 * nblk, seed=0x3D74 + c) for c in range(nch)]
    d_iq = [torch.from_numpy(x).cuda() for x in iqs]
    na = 1024000 // 50

    def run(c, stream, pl, d_pcm, d_f32):
        for b in range(nblk):
            pl.process_dev(d_iq[c].data_ptr() + b * bb, bb, d_f32[b].data_ptr(), d_pcm[b].data_ptr(), wrap=True, stream=stream.cuda_stream)

    # alone: one channel after the other, one stream
    alone = []
    s0 = torch.cuda.Stream()
    for c in range(nch):
        pl = fmrx.Pipeline(0, 1, max_block_bytes=bb)
        d_pcm, d_f32 = torch.zeros(nblk, na, dtype=torch.int16, device="cuda"), torch.zeros(nblk, na, dtype=torch.float32, device="cuda")
        run(c, s0, pl, d_pcm, d_f32)
        s0.synchronize()
        alone.append((d_pcm.cpu().numpy(), d_f32.cpu().numpy()))
        pl.close()
    # together: eight handles, eight streams, calls interleaved
    streams = [torch.cuda.Stream() for _ in range(nch)]
    pls = [fmrx.Pipeline(0, 1, max_block_bytes=bb) for _ in range(nch)]
    outs = [(torch.zeros(nblk, na, dtype=torch.int16, device="cuda"), torch.zeros(nblk, na, dtype=torch.float32, device="cuda")) for _ in range(nch)]
    torch.cuda.synchronize()
    for b in range(nblk):
        for c in range(nch):
            pls[c].process_dev(d_iq[c].data_ptr() + b * bb, bb, outs[c][1][b].data_ptr(), outs[c][0][b].data_ptr(), wrap=True,
                               stream=streams[c].cuda_stream)
    torch.cuda.synchronize()
    for c in range(nch):
        bits_equal(outs[c][0].cpu().numpy(), alone[c][0], f"pcm, channel {c}")
        bits_equal(outs[c][1].cpu().numpy(), alone[c][1], f"audio, channel {c}")
        ref = oracle.pipeline(0, 1)
        for b in range(nblk):
            want = ref.process(iqs[c][b * bb:(b + 1) * bb])["audio"]
            err = float(np.sqrt(np.mean((outs[c][1][b].cpu().numpy().astype(np.float64) - want) ** 2)))
            assert err <= 1e-4, (c, b, err)
            d = np.abs(outs[c][0][b].cpu().numpy().astype(np.int32) - oracle.pcm16(want).astype(np.int32))
            assert d.max() <= 1
    assert len({hashlib.sha256(a[0].tobytes()).digest() for a in alone}) == nch
    for pl in pls:
        pl.close()
