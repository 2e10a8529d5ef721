"""ctypes bindings to the CHECKER libraries (test infrastructure only).

* ``Oracle``  -> oracle/liboracle.so   (our C restatement, oracle/fm_oracle.c)
* ``Ref``     -> oracle/_ref/libfmref.so (the reference's own src/filter.cpp +
  src/iofunc.cpp behind oracle/ref_shim.cpp; exists only where it was built
  from /root/reference, i.e. in the build container: oracle/_ref/ is git-ignored
  and gpurun-ignored, it does not travel to the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product (software-defined-radio_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libfmref.so")
REF_TMO = os.path.join(ORACLE_DIR, "_ref", "threadMonoOnly")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")


def build_oracle() -> None:
    """(Re)build oracle/liboracle.so (and oracle/_ref when the reference is present)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class FmoParams(C.Structure):
    _fields_ = [
        ("mode", C.c_int), ("rf_Fs", C.c_int), ("if_Fs", C.c_int), ("audio_Fs", C.c_float),
        ("rf_decim", C.c_int), ("audio_decim", C.c_int), ("audio_upsamp", C.c_int),
        ("rf_taps", C.c_int), ("audio_taps", C.c_int), ("stereo_taps", C.c_int), ("block_bytes", C.c_int),
    ]


class _FirFamily:
    """Shared numpy-level API over either library (prefix 'fmo_' or 'ref_')."""

    prefix = ""

    def __init__(self, path: str):
        self.lib = C.CDLL(path)
        L, p = self.lib, self.prefix
        self._sig(p + "impulse_response_lpf", None, [C.c_float, C.c_float, C.c_ushort, f32p])
        self._sig(p + "band_pass", None, [C.c_float, C.c_float, C.c_float, C.c_ushort, f32p])
        self._sig(p + "convolve_fir", None, [f32p, f32p, C.c_size_t, f32p, C.c_size_t])
        self._sig(p + "convolve_block_fir", None, [f32p, f32p, C.c_size_t, f32p, C.c_size_t, f32p])
        self._sig(p + "convolve_block_fast_fir", None, [f32p, f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_uint])
        self._sig(p + "convolve_block_resample_fir", None,
                  [f32p, f32p, C.c_size_t, f32p, C.c_size_t, f32p, C.c_uint, C.c_uint])
        self._sig(p + "upsample", None, [f32p, C.c_size_t, f32p, C.c_int])
        self._sig(p + "downsample", C.c_size_t, [f32p, f32p, C.c_size_t, C.c_ushort])
        self._sig(p + "fm_demod", None, [f32p, f32p, f32p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)])
        self._sig(p + "all_pass", None, [f32p, C.c_size_t, f32p, C.c_size_t, f32p])
        self._sig(p + "fm_pll", None, [f32p, C.c_size_t, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float])

    def _sig(self, name, res, args):
        fn = getattr(self.lib, name)
        fn.restype = res
        fn.argtypes = args

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    # -- coefficient API: same argument order as include/filter.h ---------
    def impulse_response_lpf(self, Fs, Fc, taps) -> np.ndarray:
        h = np.zeros(taps, np.float32)
        self._fn("impulse_response_lpf")(Fs, Fc, taps, h)
        return h

    def band_pass(self, Fs, Fb, Fe, taps) -> np.ndarray:
        h = np.zeros(taps, np.float32)
        self._fn("band_pass")(Fs, Fb, Fe, taps, h)
        return h

    # -- FIR family: return (y, new_state) ---------------------------------
    def convolve_fir(self, x, h) -> np.ndarray:
        x, h = _f32(x), _f32(h)
        y = np.zeros(len(x) + len(h) - 1, np.float32)
        self._fn("convolve_fir")(y, x, len(x), h, len(h))
        return y

    def convolve_block_fir(self, x, h, state):
        x, h, st = _f32(x), _f32(h), _f32(state).copy()
        y = np.zeros(len(x), np.float32)
        self._fn("convolve_block_fir")(y, x, len(x), h, len(h), st)
        return y, st

    def convolve_block_fast_fir(self, x, h, state, decim):
        x, h, st = _f32(x), _f32(h), _f32(state).copy()
        y = np.zeros(len(x) // decim + 2, np.float32)
        self._fn("convolve_block_fast_fir")(y, x, len(x), h, len(h), st, decim)
        return y[: len(x) // decim].copy(), st

    def convolve_block_resample_fir(self, x, h, state, decim, upsamp):
        x, h, st = _f32(x), _f32(h), _f32(state).copy()
        ny = (len(x) * upsamp) // decim
        y = np.zeros(ny + 2, np.float32)
        self._fn("convolve_block_resample_fir")(y, x, len(x), h, len(h), st, decim, upsamp)
        return y[:ny].copy(), st

    def upsample(self, x, up):
        x = _f32(x)
        xu = np.zeros(len(x) * up, np.float32)
        self._fn("upsample")(x, len(x), xu, up)
        return xu

    def downsample(self, x, ds):
        x = _f32(x)
        out = np.zeros(len(x) + 1, np.float32)
        n = self._fn("downsample")(out, x, len(x), ds)
        return out[:n].copy()

    def fm_demod(self, I, Q, prev_i=0.0, prev_q=0.0):
        I, Q = _f32(I), _f32(Q)
        out = np.zeros(len(I), np.float32)
        pi, pq = C.c_float(prev_i), C.c_float(prev_q)
        self._fn("fm_demod")(out, I, Q, len(I), C.byref(pi), C.byref(pq))
        return out, pi.value, pq.value

    def all_pass(self, x, state):
        x, st = _f32(x), _f32(state).copy()
        out = np.zeros(len(x), np.float32)
        self._fn("all_pass")(x, len(x), st, len(st), out)
        return out, st

    def fm_pll(self, x, state, freq, Fs, ncoScale=2.0, phaseAdjust=0.0, normBandwidth=0.01):
        x, st = _f32(x), _f32(state).copy()
        out = np.zeros(len(x) + 1, np.float32)
        self._fn("fm_pll")(x, len(x), out, st, freq, Fs, ncoScale, phaseAdjust, normBandwidth)
        return out, st


class Oracle(_FirFamily):
    prefix = "fmo_"

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build_oracle()
        super().__init__(path)
        L = self.lib
        self._sig("fmo_u8_to_f32", None, [u8p, C.c_size_t, f32p])
        self._sig("fmo_pcm16", None, [f32p, C.c_size_t, i16p, C.c_int])
        self._sig("fmo_mode_params", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(FmoParams)])
        self._sig("fmo_pipeline_create", C.c_void_p, [C.POINTER(FmoParams), C.c_int])
        self._sig("fmo_pipeline_destroy", None, [C.c_void_p])
        self._sig("fmo_pipeline_process", C.c_size_t,
                  [C.c_void_p, u8p, C.c_size_t] + [C.c_void_p] * 5)
        self._sig("fmo_pipeline_n_if", C.c_size_t, [C.c_void_p, C.c_size_t])
        self._sig("fmo_pipeline_n_audio", C.c_size_t, [C.c_void_p, C.c_size_t])
        self._sig("fmo_pipeline_intermediate", C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_float))])
        self._sig("fmo_synth_fm_u8", None, [u8p, C.c_size_t, C.c_double, C.c_uint64, C.c_uint64])
        self._sig("fmo_estimate_psd", C.c_int, [f32p, f32p, f32p, C.c_size_t, C.c_float, C.c_int])
        self._sig("fmo_libm", None, [C.c_int, f32p, f32p, C.c_size_t, f32p])

    def u8_to_f32(self, raw):
        raw = np.ascontiguousarray(raw, np.uint8)
        out = np.zeros(len(raw), np.float32)
        self.lib.fmo_u8_to_f32(raw, len(raw), out)
        return out

    def pcm16(self, audio, wrap=True):
        a = _f32(audio)
        out = np.zeros(len(a), np.int16)
        self.lib.fmo_pcm16(a, len(a), out, 1 if wrap else 0)
        return out

    def estimate_psd(self, samples, Fs, nfft=512):
        x = _f32(samples)
        freq, psd = np.zeros(nfft // 2, np.float32), np.zeros(nfft // 2, np.float32)
        self.lib.fmo_estimate_psd(freq, psd, x, len(x), Fs, nfft)
        return freq, psd

    def mode_params(self, mode, rf_taps=101, base_audio_taps=101, stereo_taps=101) -> FmoParams:
        p = FmoParams()
        if self.lib.fmo_mode_params(mode, rf_taps, base_audio_taps, stereo_taps, C.byref(p)) != 0:
            raise ValueError(f"bad mode {mode}")
        return p

    def synth_fm_u8(self, n_samples, rf_Fs=2.4e6, seed=0x3D74, start=0) -> np.ndarray:
        iq = np.zeros(2 * n_samples, np.uint8)
        self.lib.fmo_synth_fm_u8(iq, n_samples, float(rf_Fs), seed, start)
        return iq

    def pipeline(self, mode=0, channels=1, rf_taps=101, base_audio_taps=101, stereo_taps=101):
        return OraclePipeline(self, self.mode_params(mode, rf_taps, base_audio_taps, stereo_taps), channels)

    def pipeline_params(self, params: FmoParams, channels=1):
        return OraclePipeline(self, params, channels)

    def libm(self, fn: str, a, b=None) -> np.ndarray:
        """sinf / cosf / atan2f of this host's C library (what the reference calls)."""
        a = _f32(a)
        b = _f32(b) if b is not None else a
        out = np.zeros(len(a), np.float32)
        self.lib.fmo_libm({"sinf": 0, "cosf": 1, "atan2f": 2}[fn], a, b, len(a), out)
        return out


class OraclePipeline:
    NAMES = {"carrier_filt": 0, "stereo_filt": 1, "pll": 2, "mixer": 3, "allpass": 4, "mono_filt": 5, "stereo_final": 6}

    def __init__(self, orc: Oracle, params: FmoParams, channels: int):
        self.o, self.p, self.channels = orc, params, channels
        self.h = orc.lib.fmo_pipeline_create(C.byref(params), channels)

    def __del__(self):
        if getattr(self, "h", None):
            self.o.lib.fmo_pipeline_destroy(self.h)
            self.h = None

    def process(self, iq_u8) -> dict:
        iq = np.ascontiguousarray(iq_u8, np.uint8)
        L = self.o.lib
        n_if, n_a = L.fmo_pipeline_n_if(self.h, len(iq)), L.fmo_pipeline_n_audio(self.h, len(iq))
        out = {k: np.zeros(n_if, np.float32) for k in ("if_i", "if_q", "demod")}
        out["audio_l"] = np.zeros(n_a, np.float32)
        out["audio_r"] = np.zeros(n_a, np.float32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        L.fmo_pipeline_process(self.h, iq, len(iq), ptr(out["if_i"]), ptr(out["if_q"]), ptr(out["demod"]),
                               ptr(out["audio_l"]), ptr(out["audio_r"]) if self.channels == 2 else None)
        if self.channels == 1:
            del out["audio_r"]
            out["audio"] = out["audio_l"]
        return out

    def intermediate(self, name) -> np.ndarray:
        pp = C.POINTER(C.c_float)()
        n = self.o.lib.fmo_pipeline_intermediate(self.h, self.NAMES[name], C.byref(pp))
        return np.ctypeslib.as_array(pp, shape=(n,)).copy()


class Ref(_FirFamily):
    prefix = "ref_"

    def __init__(self, path: str = REF_SO):
        super().__init__(path)
        self._sig("ref_read_block", None, [u8p, C.c_size_t, f32p])
        self._sig("ref_pcm16", None, [f32p, C.c_size_t, i16p])
        self._sig("ref_pipeline_create", C.c_void_p, [C.c_int] * 5)
        if hasattr(self.lib, "ref_estimate_psd"):
            self._sig("ref_estimate_psd", C.c_int, [f32p, f32p, f32p, C.c_size_t, C.c_float])
        if hasattr(self.lib, "ref_pipeline_create_params"):
            self._sig("ref_pipeline_create_params", C.c_void_p, [C.c_int] * 9)
        self._sig("ref_pipeline_destroy", None, [C.c_void_p])
        self._sig("ref_pipeline_process", C.c_size_t, [C.c_void_p, u8p, C.c_size_t])
        self._sig("ref_pipeline_get", C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.POINTER(C.c_float))])

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def u8_to_f32(self, raw):
        raw = np.ascontiguousarray(raw, np.uint8)
        out = np.zeros(len(raw), np.float32)
        self.lib.ref_read_block(raw, len(raw), out)
        return out

    def pcm16(self, audio, wrap=True):
        a = _f32(audio)
        out = np.zeros(len(a), np.int16)
        self.lib.ref_pcm16(a, len(a), out)
        return out

    def estimate_psd(self, samples, Fs):
        x = _f32(samples)
        freq, psd = np.zeros(256, np.float32), np.zeros(256, np.float32)
        self.lib.ref_estimate_psd(freq, psd, x, len(x), Fs)
        return freq, psd

    def pipeline(self, mode=0, channels=1, rf_taps=101, base_audio_taps=101, stereo_taps=101):
        return RefPipeline(self, mode, channels, rf_taps, base_audio_taps, stereo_taps)

    def pipeline_params(self, p, channels=1):
        """The reference's graph at explicit parameters (FmoParams-like: rf_Fs, if_Fs, rf_decim, ...)."""
        rp = RefPipeline.__new__(RefPipeline)
        rp.r, rp.channels = self, channels
        rp.h = self.lib.ref_pipeline_create_params(p.rf_Fs, p.if_Fs, p.rf_decim, p.audio_decim, p.audio_upsamp, p.rf_taps,
                                                   p.audio_taps, p.stereo_taps, channels)
        return rp


class RefPipeline:
    NAMES = {"carrier_filt": 0, "stereo_filt": 1, "pll": 2, "mixer": 3, "allpass": 4, "mono_filt": 5,
             "stereo_final": 6, "if_i": 7, "if_q": 8, "demod": 9, "audio_l": 10, "audio_r": 11}

    def __init__(self, ref: Ref, mode, channels, rf_taps, base_audio_taps, stereo_taps):
        self.r, self.channels = ref, channels
        self.h = ref.lib.ref_pipeline_create(mode, channels, rf_taps, base_audio_taps, stereo_taps)

    def __del__(self):
        if getattr(self, "h", None):
            self.r.lib.ref_pipeline_destroy(self.h)
            self.h = None

    def intermediate(self, name) -> np.ndarray:
        pp = C.POINTER(C.c_float)()
        n = self.r.lib.ref_pipeline_get(self.h, self.NAMES[name], C.byref(pp))
        return np.ctypeslib.as_array(pp, shape=(n,)).copy() if n else np.zeros(0, np.float32)

    def process(self, iq_u8) -> dict:
        iq = np.ascontiguousarray(iq_u8, np.uint8)
        self.r.lib.ref_pipeline_process(self.h, iq, len(iq))
        out = {k: self.intermediate(k) for k in ("if_i", "if_q", "demod", "audio_l")}
        if self.channels == 2:
            out["audio_r"] = self.intermediate("audio_r")
        else:
            out["audio"] = out["audio_l"]
        return out
