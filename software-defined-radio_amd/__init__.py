"""software-defined-radio_amd -- MI355X-native FM receiver DSP hot path.

Python host mirror of the reference's operator interface for this path: the
free functions of ``include/filter.h:18-43`` and ``include/iofunc.h:36`` of
mnigm2001/Software-Defined-Radio (same names, argument order and meaning; numpy
arrays in place of ``std::vector<float>&``, outputs returned instead of passed
by reference), plus the pipeline handle that replaces ``src/project.cpp``'s
thread bodies.  Everything here is a thin ctypes binding over the C ABI of
``lib/libfmrx.so`` (``include/fmrx.h``); all compute runs in HIP kernels on the
GPU.  There is no CPU fallback: without the built library this module raises on
import, and without a GPU every compute call raises ``FmrxError`` (ENODEV).

The directory name contains '-', so import it with::

    import importlib; fmrx = importlib.import_module("software-defined-radio_amd")
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMRX_LIB") or os.path.join(_HERE, "lib", "libfmrx.so")   # FMRX_LIB: A/B a development build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fmrx.h")

OK, EINVAL, ENODEV, EHIP, ENOMEM = 0, 1, 2, 3, 4
PCM_WRAP, PCM_SATURATE = 1, 0
TAPS = {"if_i": 0, "if_q": 1, "demod": 2, "mono_filt": 3, "carrier_filt": 4, "stereo_filt": 5, "pll": 6, "mixer": 7,
        "stereo_final": 8}


class FmrxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fmrx error {code}: {msg}")
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C software-defined-radio_amd/csrc`). There is no Python/CPU fallback.")



# libfmrx.so and PyTorch can be loaded in either order: the library records its HIP / HSA runtime dependencies
# by the unversioned names torch's own libraries use, so ld.so maps ONE runtime whichever comes first
# (csrc/Makefile; tests/test_load_order.py runs both orders on the GPU).
lib = C.CDLL(LIB_PATH)

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_vp, _sz, _int, _uint, _flt = C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_float


class Params(C.Structure):
    """struct fmrx_params == the reference's PARAMS + mode table (src/project.cpp:17-27, 424-427)."""
    _fields_ = [("mode", _int), ("rf_Fs", _int), ("if_Fs", _int), ("audio_Fs", _flt), ("rf_decim", _int),
                ("audio_decim", _int), ("audio_upsamp", _int), ("rf_taps", _int), ("audio_taps", _int),
                ("stereo_taps", _int), ("block_bytes", _int)]


def _sig(name, args, res=_int):
    fn = getattr(lib, name)
    fn.argtypes, fn.restype = args, res
    return fn


_sig("fmrx_version", [], C.c_char_p)
_sig("fmrx_last_error", [], C.c_char_p)
_sig("fmrx_device_count", [])
_sig("fmrx_set_device", [_int])
_sig("fmrx_set_option", [C.c_char_p, C.c_long])
_sig("fmrx_get_option", [C.c_char_p, C.POINTER(C.c_long)])
_sig("fmrx_host_alloc", [C.POINTER(_vp), _sz])
_sig("fmrx_host_free", [_vp])
_sig("fmrx_impulse_response_lpf", [_flt, _flt, C.c_ushort, _f32p])
_sig("fmrx_band_pass", [_flt, _flt, _flt, C.c_ushort, _f32p])
_sig("fmrx_u8_to_f32", [_u8p, _sz, _f32p])
_sig("fmrx_deinterleave", [_f32p, _sz, _f32p, _f32p])
_sig("fmrx_convolve_fir", [_f32p, _f32p, _sz, _f32p, _sz])
_sig("fmrx_convolve_block_fir", [_f32p, _f32p, _sz, _f32p, _sz, _f32p])
_sig("fmrx_convolve_block_fast_fir", [_f32p, _f32p, _sz, _f32p, _sz, _f32p, _uint])
_sig("fmrx_convolve_block_resample_fir", [_f32p, _f32p, _sz, _f32p, _sz, _f32p, _uint, _uint])
_sig("fmrx_upsample", [_f32p, _sz, _f32p, _int])
_sig("fmrx_downsample", [_f32p, C.POINTER(_sz), _f32p, _sz, C.c_ushort])
_sig("fmrx_fm_demod", [_f32p, _f32p, _f32p, _sz, C.POINTER(_flt), C.POINTER(_flt)])
_sig("fmrx_all_pass", [_f32p, _sz, _f32p, _sz, _f32p])
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_sig("fmrx_fm_demod_arctan", [_f64p, _f64p, _f64p, _sz, C.POINTER(C.c_double)])
_sig("fmrx_fm_pll", [_f32p, _sz, _f32p, _f32p, _flt, _flt, _flt, _flt, _flt])
_sig("fmrx_stereo_mix", [_f32p, _f32p, _sz, _f32p])
_sig("fmrx_stereo_combine", [_f32p, _f32p, _sz, _f32p, _f32p])
_sig("fmrx_pcm16", [_f32p, _sz, _i16p, _int])
_sig("fmrx_estimate_psd", [_f32p, _f32p, _f32p, _sz, _flt, _int])
_sig("fmrx_diag_libm", [_int, _f32p, _vp, _sz, _f32p])
_sig("fmrx_diag_stream_read_dev", [_vp, _sz, _int, _vp])
_sig("fmrx_mode_params", [_int, _int, _int, _int, C.POINTER(Params)])
_sig("fmrx_pipeline_create", [C.POINTER(_vp), C.POINTER(Params), _int, _sz, _int])
_sig("fmrx_pipeline_destroy", [_vp])
_sig("fmrx_pipeline_reset", [_vp])
_sig("fmrx_pipeline_n_if", [_vp, _sz], _sz)
_sig("fmrx_pipeline_n_audio", [_vp, _sz], _sz)
_sig("fmrx_pipeline_process", [_vp, _u8p, _sz, _vp, _vp, _int])
_sig("fmrx_pipeline_process_dev", [_vp, _vp, _sz, _vp, _vp, _int, _vp])
_sig("fmrx_pipeline_submit", [_vp, _vp, _sz, _vp, _vp, _int])
_sig("fmrx_pipeline_wait", [_vp])
_sig("fmrx_pipeline_read_tap", [_vp, _int, _vp, C.POINTER(_sz)])
_sig("fmrx_pipeline_state_size", [_vp], _sz)
_sig("fmrx_pipeline_get_state", [_vp, _f32p, _sz])
_sig("fmrx_pipeline_set_state", [_vp, _f32p, _sz])
_sig("fmrx_pipeline_last_timing", [_vp, _f32p])
_sig("fmrx_pipeline_timing_sum", [_vp, _f32p, C.POINTER(_int), _int])
_sig("fmrx_pipeline_set_profiling", [_vp, _int])
_sig("fmrx_pipeline_set_force_generic", [_vp, _int])
_sig("fmrx_pipeline_set_keep_intermediates", [_vp, _int])
_sig("fmrx_pipeline_set_option", [_vp, C.c_char_p, C.c_long])
_sig("fmrx_pipeline_pll_diagnostics", [_vp, C.POINTER(_uint), C.POINTER(_flt), C.POINTER(_flt)])
_sig("fmrx_channels_create", [C.POINTER(_vp), C.POINTER(Params), _int, _sz, _int])
_sig("fmrx_channels_create_ex", [C.POINTER(_vp), C.POINTER(Params), _int, _int, _int, _sz, _int])
_sig("fmrx_channels_read_tap", [_vp, _int, _int, _vp, C.POINTER(_sz)])
_sig("fmrx_channels_destroy", [_vp])
_sig("fmrx_channels_n_audio", [_vp], _sz)
_sig("fmrx_channels_input_layout", [_vp, C.POINTER(_vp), C.POINTER(_sz)])
_sig("fmrx_channels_reset", [_vp, _int])
_sig("fmrx_channels_load_dev", [_vp, _vp, _vp])
_sig("fmrx_channels_process", [_vp, _u8p, _vp, _vp, _int])
_sig("fmrx_channels_process_dev", [_vp, _vp, _vp, _int, _vp])
class RdsParams(C.Structure):
    _fields_ = [("if_Fs", _int), ("taps", _int), ("upsamp", _int), ("decim", _int), ("sps", _int), ("rrc_taps", _int)]


_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_sig("fmrx_rds_mode_params", [_int, C.POINTER(RdsParams)])
_sig("fmrx_rds_create", [C.POINTER(_vp), C.POINTER(RdsParams), _sz, _int])
_sig("fmrx_rds_destroy", [_vp])
_sig("fmrx_rds_reset", [_vp])
_sig("fmrx_rds_n_out", [_vp, _sz], _sz)
_sig("fmrx_rds_process", [_vp, _f32p, _sz, _vp, _vp, _vp, C.POINTER(_sz), C.c_char_p])
_sig("fmrx_rds_process_dev", [_vp, _vp, _sz, _vp])
_sig("fmrx_rds_read_tap", [_vp, _int, _vp, C.POINTER(_sz)])
_sig("fmrx_rds_band_pass", [_int, C.c_double, C.c_double, C.c_double, _f64p])
_sig("fmrx_rds_imp_response", [_int, C.c_double, C.c_double, _f64p])
_sig("fmrx_rds_rrc", [C.c_double, _int, _f64p])
_sig("fmrx_rds_cdr", [_f64p, _sz, _int, _int, _f64p, _u8p, C.POINTER(_sz)])
_sig("fmrx_rds_diff_decode", [_u8p, _sz, _u8p])
_sig("fmrx_rds_frame_sync", [_u8p, _sz, C.c_char_p, C.POINTER(_sz)])
_sig("fmrx_fe_fir_decim_u8", [_u8p, _sz, _f32p, _sz, _uint, _vp, _vp, _vp, _int])
_sig("fmrx_fe_plan_create", [C.POINTER(_vp), _f32p, _sz, _uint])
_sig("fmrx_fe_plan_destroy", [_vp])
_sig("fmrx_fe_plan_is_specialised", [_vp])
_sig("fmrx_fe_plan_history_bytes", [_vp], _sz)
_sig("fmrx_fe_run_dev", [_vp, _vp, _sz, _vp, _vp, _int, _vp])


def _check(rc: int):
    if rc != OK:
        raise FmrxError(rc, lib.fmrx_last_error().decode(errors="replace"))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


def version() -> str:
    return lib.fmrx_version().decode()


def device_count() -> int:
    return lib.fmrx_device_count()


def set_device(dev: int) -> None:
    _check(lib.fmrx_set_device(dev))


_FE_VARIANTS = {"mfma": 0, "valu": 1, "discriminator": 0, "arctan": 1}   # option values that have names


def set_option(name: str, value) -> None:
    """Process-wide default of a run-time option (include/fmrx.h: fmrx_set_option); pipelines created
    afterwards start from it.  fe_variant also takes "mfma" / "valu"."""
    _check(lib.fmrx_set_option(name.encode(), int(_FE_VARIANTS.get(value, value))))


def get_option(name: str) -> int:
    v = C.c_long(0)
    _check(lib.fmrx_get_option(name.encode(), C.byref(v)))
    return v.value


def diagStreamRead(d_ptr, n_bytes, method=0, stream=None) -> None:
    """One pure streaming read of a device buffer (fmrx_diag_stream_read_dev), async on `stream`."""
    _check(lib.fmrx_diag_stream_read_dev(d_ptr, n_bytes, method, stream))


def hostAlloc(nbytes: int, dtype=np.uint8) -> np.ndarray:
    """A page-locked host buffer (fmrx_host_alloc) as a numpy array of `dtype`; free it with hostFree(arr)."""
    p = _vp()
    _check(lib.fmrx_host_alloc(C.byref(p), int(nbytes)))
    buf = (C.c_uint8 * int(nbytes)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype)
    arr.flags.writeable = True
    _PINNED[arr.ctypes.data] = (p.value, buf)
    return arr


def hostFree(arr: np.ndarray) -> None:
    p, _ = _PINNED.pop(arr.ctypes.data)
    _check(lib.fmrx_host_free(p))


_PINNED: dict = {}


def deviceLibm(fn: str, a, b=None, flat=False) -> np.ndarray:
    """sinf / cosf / atan2f as the device evaluates csrc/glibc_libm.hpp (fmrx_diag_libm): test hook.
    flat: through the branch-free forms the receiver banks' PLL lanes run."""
    a = _f32(a)
    out = np.zeros(len(a), np.float32)
    bb = _f32(b) if b is not None else None
    _check(lib.fmrx_diag_libm({"sinf": 0, "cosf": 1, "atan2f": 2}[fn] + (3 if flat else 0), a,
                              bb.ctypes.data if bb is not None else None, len(a), out))
    return out


# --------------------------------------------------------------------------
# filter.h mirror (reference names and argument order)
# --------------------------------------------------------------------------
def impulseResponseLPF(Fs, Fc, num_taps) -> np.ndarray:
    """filter.h:24 / filter.cpp:103-114 -> h[num_taps]."""
    h = np.zeros(num_taps, np.float32)
    _check(lib.fmrx_impulse_response_lpf(Fs, Fc, num_taps, h))
    return h


def bandPass(Fs, Fb, Fe, N_taps) -> np.ndarray:
    """filter.h:20 / filter.cpp:83-99 -> coeff[N_taps] (C++ argument order: Fs, Fb, Fe, taps)."""
    h = np.zeros(N_taps, np.float32)
    _check(lib.fmrx_band_pass(Fs, Fb, Fe, N_taps, h))
    return h


def convolveFIR(x, h) -> np.ndarray:
    """filter.h:26 / filter.cpp:118-130 -> y[len(x)+len(h)-1]."""
    x, h = _f32(x), _f32(h)
    y = np.zeros(len(x) + len(h) - 1, np.float32)
    _check(lib.fmrx_convolve_fir(y, x, len(x), h, len(h)))
    return y


def convolveBlockFIR(x, h, state):
    """filter.h:28 / filter.cpp:133-154 -> (y[len(x)], new_state)."""
    x, h, st = _f32(x), _f32(h), _f32(state).copy()
    if len(st) != len(h) - 1:
        raise FmrxError(EINVAL, "state must have len(h)-1 elements")
    y = np.zeros(len(x), np.float32)
    _check(lib.fmrx_convolve_block_fir(y, x, len(x), h, len(h), st))
    return y, st


def convolveBlockFastFIR(x, h, state, audio_decim):
    """filter.h:31 / filter.cpp:158-188 -> (y[len(x)//decim], new_state)."""
    x, h, st = _f32(x), _f32(h), _f32(state).copy()
    if len(st) != len(h) - 1:
        raise FmrxError(EINVAL, "state must have len(h)-1 elements")
    y = np.zeros(len(x) // max(int(audio_decim), 1), np.float32)
    _check(lib.fmrx_convolve_block_fast_fir(y, x, len(x), h, len(h), st, audio_decim))
    return y, st


def convolveBlockResampleFIR(x, h, state, audio_decim, audio_upsamp):
    """filter.h:34 / filter.cpp:191-223 -> (y[len(x)*U//D], new_state); state in the reference's layout."""
    x, h, st = _f32(x), _f32(h), _f32(state).copy()
    if len(st) != len(h) - 1:
        raise FmrxError(EINVAL, "state must have len(h)-1 elements")
    ny = (len(x) * int(audio_upsamp)) // max(int(audio_decim), 1)
    y = np.zeros(ny, np.float32)
    _check(lib.fmrx_convolve_block_resample_fir(y, x, len(x), h, len(h), st, audio_decim, audio_upsamp))
    return y, st


def upsample(x, up_rate) -> np.ndarray:
    x = _f32(x)
    xu = np.zeros(len(x) * up_rate, np.float32)
    _check(lib.fmrx_upsample(x, len(x), xu, up_rate))
    return xu


def downsample(x, ds_coeff) -> np.ndarray:
    x = _f32(x)
    out = np.zeros(len(x) + 1, np.float32)
    n = _sz(0)
    _check(lib.fmrx_downsample(out, C.byref(n), x, len(x), ds_coeff))
    return out[: n.value].copy()


def fmDemod(I, Q, prev_i=0.0, prev_q=0.0):
    """filter.h:41 / filter.cpp:248-266 -> (fm_demod, prev_i, prev_q)."""
    I, Q = _f32(I), _f32(Q)
    out = np.zeros(len(I), np.float32)
    pi, pq = _flt(prev_i), _flt(prev_q)
    _check(lib.fmrx_fm_demod(out, I, Q, len(I), C.byref(pi), C.byref(pq)))
    return out, pi.value, pq.value


def fmDemodArctan(I, Q, prev_phase=0.0):
    """model/fmSupportLib.py:502-531 -> (fm_demod, prev_phase): the Python model's arctangent demodulator, float64."""
    I, Q = np.ascontiguousarray(I, np.float64), np.ascontiguousarray(Q, np.float64)
    out = np.zeros(len(I), np.float64)
    ph = C.c_double(prev_phase)
    _check(lib.fmrx_fm_demod_arctan(out, I, Q, len(I), C.byref(ph)))
    return out, ph.value


def allPass(input_block, state_block):
    """filter.h:18 / filter.cpp:14-29 -> (output_block, new_state)."""
    x, st = _f32(input_block), _f32(state_block).copy()
    out = np.zeros(len(x), np.float32)
    _check(lib.fmrx_all_pass(x, len(x), st, len(st), out))
    return out, st


def fmPLL(PLLIn, state, freq, Fs, ncoScale=2.0, phaseAdjust=0.0, normBandwidth=0.01):
    """filter.h:22 / filter.cpp:32-80 -> (ncoOut[len+1], new_state[6])."""
    x, st = _f32(PLLIn), _f32(state).copy()
    if len(st) != 6:
        raise FmrxError(EINVAL, "PLL state has 6 elements")
    out = np.zeros(len(x) + 1, np.float32)
    _check(lib.fmrx_fm_pll(x, len(x), out, st, freq, Fs, ncoScale, phaseAdjust, normBandwidth))
    return out, st


def stereoMix(stereo_filt, pll) -> np.ndarray:
    a, b = _f32(stereo_filt), _f32(pll)
    out = np.zeros(len(a), np.float32)
    _check(lib.fmrx_stereo_mix(a, b, len(a), out))
    return out


def stereoCombine(stereo_final, mono):
    a, b = _f32(stereo_final), _f32(mono)
    l, r = np.zeros(len(a), np.float32), np.zeros(len(a), np.float32)
    _check(lib.fmrx_stereo_combine(a, b, len(a), l, r))
    return l, r


def estimatePSD(samples, Fs, nfft=512):
    """fourier.h / fourier.cpp:44-128 -> (freq[nfft/2], psd_est[nfft/2] in dB); NFFT = 512 in the reference."""
    x = _f32(samples)
    freq, psd = np.zeros(nfft // 2, np.float32), np.zeros(nfft // 2, np.float32)
    _check(lib.fmrx_estimate_psd(freq, psd, x, len(x), Fs, nfft))
    return freq, psd


def readBlockData(raw_u8) -> np.ndarray:
    """iofunc.h:36 / iofunc.cpp:128-135, the conversion part: (u8-128)/128."""
    raw = _u8(raw_u8)
    out = np.zeros(len(raw), np.float32)
    _check(lib.fmrx_u8_to_f32(raw, len(raw), out))
    return out


def deinterleave(iq):
    iq = _f32(iq)
    n = len(iq) // 2
    I, Q = np.zeros(n, np.float32), np.zeros(n, np.float32)
    _check(lib.fmrx_deinterleave(iq, n, I, Q))
    return I, Q


def pcm16(audio, wrap=True) -> np.ndarray:
    a = _f32(audio)
    out = np.zeros(len(a), np.int16)
    _check(lib.fmrx_pcm16(a, len(a), out, PCM_WRAP if wrap else PCM_SATURATE))
    return out


def frontEndFIR(iq_u8, h, decim, hist=None, force_generic=False):
    """Fused front end on host buffers -> (if_i, if_q, new_hist).  hist: u8[2*(taps-1)] or None."""
    iq, h = _u8(iq_u8), _f32(h)
    n = len(iq) // 2
    fi, fq = np.zeros(n // decim, np.float32), np.zeros(n // decim, np.float32)
    hb = None
    if hist is not None:
        hb = _u8(hist).copy()
        if len(hb) != 2 * (len(h) - 1):
            raise FmrxError(EINVAL, "hist must have 2*(taps-1) bytes")
    _check(lib.fmrx_fe_fir_decim_u8(iq, n, h, len(h), decim, hb.ctypes.data if hb is not None else None,
                                    fi.ctypes.data, fq.ctypes.data, 1 if force_generic else 0))
    return fi, fq, hb


def modeParams(mode, rf_taps=101, base_audio_taps=101, stereo_taps=101) -> Params:
    p = Params()
    _check(lib.fmrx_mode_params(mode, rf_taps, base_audio_taps, stereo_taps, C.byref(p)))
    return p


# --------------------------------------------------------------------------
# pipeline handle
# --------------------------------------------------------------------------
class Pipeline:
    """RF_FrontEnd + RF_MONO / RF_STEREO of src/project.cpp as one device-resident pipeline."""

    def __init__(self, mode=0, channels=1, rf_taps=101, base_audio_taps=101, stereo_taps=101, max_block_bytes=None,
                 device=0, params: Params | None = None):
        self.params = params if params is not None else modeParams(mode, rf_taps, base_audio_taps, stereo_taps)
        self.channels = channels
        self.max_block_bytes = int(max_block_bytes or self.params.block_bytes)
        self._h = _vp()
        _check(lib.fmrx_pipeline_create(C.byref(self._h), C.byref(self.params), channels, self.max_block_bytes, device))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # lib is None during interpreter shutdown
            lib.fmrx_pipeline_destroy(self._h)
            self._h = None

    __del__ = close

    def n_if(self, n_bytes):
        return lib.fmrx_pipeline_n_if(self._h, n_bytes)

    def n_audio(self, n_bytes):
        return lib.fmrx_pipeline_n_audio(self._h, n_bytes)

    def reset(self):
        _check(lib.fmrx_pipeline_reset(self._h))

    def set_profiling(self, on=True):
        _check(lib.fmrx_pipeline_set_profiling(self._h, int(on)))

    def set_keep_intermediates(self, on=True):
        """Also store the IF I/Q stream (read_tap('if_i'/'if_q')); the fused front end skips it by default."""
        _check(lib.fmrx_pipeline_set_keep_intermediates(self._h, int(on)))

    def pll_diagnostics(self):
        """(segments repaired serially, max accepted |dphase|, max accepted |dinteg|) of the parallel PLL."""
        r, dp, di = _uint(0), _flt(0), _flt(0)
        _check(lib.fmrx_pipeline_pll_diagnostics(self._h, C.byref(r), C.byref(dp), C.byref(di)))
        return r.value, dp.value, di.value

    def set_force_generic(self, on=True):
        """on: the bit-exact mode (reference evaluation order everywhere, serial PLL with glibc's functions)."""
        _check(lib.fmrx_pipeline_set_force_generic(self._h, int(on)))

    def set_option(self, name: str, value):
        """Per-handle run-time option (fmrx_pipeline_set_option); fe_variant also takes "mfma" / "valu"."""
        _check(lib.fmrx_pipeline_set_option(self._h, name.encode(), int(_FE_VARIANTS.get(value, value))))

    def process(self, iq_u8, want_pcm=True, wrap=True):
        """One block of interleaved u8 I/Q (host) -> dict(audio=..., [audio_l, audio_r], pcm16=...)."""
        iq = _u8(iq_u8)
        na = self.n_audio(len(iq))
        f = np.zeros(self.channels * na, np.float32)
        s = np.zeros(self.channels * na, np.int16) if want_pcm else None
        _check(lib.fmrx_pipeline_process(self._h, iq, len(iq), f.ctypes.data, s.ctypes.data if want_pcm else None,
                                         PCM_WRAP if wrap else PCM_SATURATE))
        out = {"pcm16": s}
        if self.channels == 1:
            out["audio"] = out["audio_l"] = f
        else:
            out["audio_l"], out["audio_r"] = f[:na], f[na:]
        return out

    def submit(self, iq_ptr, n_bytes, audio_ptr=None, pcm_ptr=None, wrap=True):
        """fmrx_pipeline_submit on raw HOST addresses (e.g. page-locked buffers from hostAlloc): enqueue and return."""
        _check(lib.fmrx_pipeline_submit(self._h, iq_ptr, n_bytes, audio_ptr, pcm_ptr, PCM_WRAP if wrap else PCM_SATURATE))

    def wait(self):
        """fmrx_pipeline_wait: the oldest submitted block's outputs are complete."""
        _check(lib.fmrx_pipeline_wait(self._h))

    def process_dev(self, d_iq_ptr, n_bytes, d_audio_ptr=None, d_pcm_ptr=None, wrap=True, stream=None):
        """Device-resident block: raw device addresses (e.g. torch.Tensor.data_ptr()); async on `stream`."""
        _check(lib.fmrx_pipeline_process_dev(self._h, d_iq_ptr, n_bytes, d_audio_ptr, d_pcm_ptr,
                                             PCM_WRAP if wrap else PCM_SATURATE, stream))

    def read_tap(self, name) -> np.ndarray:
        n = _sz(0)
        _check(lib.fmrx_pipeline_read_tap(self._h, TAPS[name], None, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        _check(lib.fmrx_pipeline_read_tap(self._h, TAPS[name], out.ctypes.data, C.byref(n)))
        return out

    def get_state(self) -> np.ndarray:
        st = np.zeros(lib.fmrx_pipeline_state_size(self._h), np.float32)
        _check(lib.fmrx_pipeline_get_state(self._h, st, len(st)))
        return st

    def set_state(self, st):
        st = _f32(st)
        _check(lib.fmrx_pipeline_set_state(self._h, st, len(st)))

    def last_timing(self):
        t = np.zeros(4, np.float32)
        _check(lib.fmrx_pipeline_last_timing(self._h, t))
        return dict(front_end_ms=float(t[0]), audio_ms=float(t[1]), rest_ms=float(t[2]), total_ms=float(t[3]))

    def timing_sum(self, max_calls=0):
        """Sum of per-stage device times (ms) over the most recent profiled calls -> (dict, count)."""
        t, n = np.zeros(4, np.float32), _int(0)
        _check(lib.fmrx_pipeline_timing_sum(self._h, t, C.byref(n), max_calls))
        return dict(front_end_ms=float(t[0]), audio_ms=float(t[1]), rest_ms=float(t[2]), total_ms=float(t[3])), n.value


class Channels:
    """N independent receivers (all four modes), the current block of all of them in one device call (fmrx_channels_*).

    audio_channels: 1 mono, 2 stereo (the reference's second command-line argument).  exact=True: every stage in the
    reference's float32 evaluation order, fmPLL as the serial recurrence with glibc's functions, one lane per channel --
    audio equals the compiled reference's bit for bit.  exact=False with audio_channels=2: the fast stereo bank (matrix-core
    front end, fma FIRs, the PLL's fast recurrence one lane per channel; the default stereo path's error envelope)."""

    def __init__(self, mode=0, n_channels=1, rf_taps=101, base_audio_taps=101, block_bytes=None, device=0, params: Params | None = None,
                 audio_channels=1, exact=False, stereo_taps=101):
        self.params = params if params is not None else modeParams(mode, rf_taps, base_audio_taps, stereo_taps)
        self.n_channels = int(n_channels)
        self.audio_channels = int(audio_channels)
        self.exact = bool(exact)
        self.block_bytes = int(block_bytes or self.params.block_bytes)
        self._h = _vp()
        _check(lib.fmrx_channels_create_ex(C.byref(self._h), C.byref(self.params), self.n_channels, self.audio_channels,
                                           int(self.exact), self.block_bytes, device))
        self.n_audio = lib.fmrx_channels_n_audio(self._h)

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.fmrx_channels_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self, channel=-1):
        _check(lib.fmrx_channels_reset(self._h, channel))

    def input_layout(self):
        """(device address of channel 0's block, pitch in bytes between channels)."""
        ptr, pitch = _vp(), _sz(0)
        _check(lib.fmrx_channels_input_layout(self._h, C.byref(ptr), C.byref(pitch)))
        return ptr.value, pitch.value

    def process(self, iq_u8, want_pcm=True, wrap=True):
        """iq_u8: [n_channels, block_bytes] uint8 (host) -> dict(audio=[n_channels, n_audio] f32, pcm16=... s16)."""
        iq = _u8(iq_u8).reshape(self.n_channels, self.block_bytes)
        st = self.audio_channels == 2
        f = np.zeros((self.n_channels, 2, self.n_audio) if st else (self.n_channels, self.n_audio), np.float32)
        s = None
        if want_pcm:
            s = np.zeros((self.n_channels, self.n_audio, 2) if st else (self.n_channels, self.n_audio), np.int16)
        _check(lib.fmrx_channels_process(self._h, iq.reshape(-1), f.ctypes.data, s.ctypes.data if want_pcm else None,
                                         PCM_WRAP if wrap else PCM_SATURATE))
        if st:   # stereo: audio [n_channels, 2, n_audio] (left, right), pcm16 [n_channels, n_audio, 2] interleaved L,R
            return {"audio": f, "audio_l": f[:, 0], "audio_r": f[:, 1], "pcm16": s}
        return {"audio": f, "pcm16": s}

    def read_tap(self, channel, name) -> np.ndarray:
        """Exact banks: one channel's intermediate of the last call ('demod', 'carrier_filt', 'stereo_filt', 'pll')."""
        n = _sz(0)
        _check(lib.fmrx_channels_read_tap(self._h, channel, TAPS[name], None, C.byref(n)))
        out = np.zeros(n.value, np.float32)
        _check(lib.fmrx_channels_read_tap(self._h, channel, TAPS[name], out.ctypes.data, C.byref(n)))
        return out

    def load_dev(self, d_iq_ptr, stream=None):
        """Device-resident [n_channels, block_bytes] blocks -> the channels' slots (async on `stream`)."""
        _check(lib.fmrx_channels_load_dev(self._h, d_iq_ptr, stream))

    def process_dev(self, d_audio_ptr=None, d_pcm_ptr=None, wrap=True, stream=None):
        _check(lib.fmrx_channels_process_dev(self._h, d_audio_ptr, d_pcm_ptr, PCM_WRAP if wrap else PCM_SATURATE, stream))


# --------------------------------------------------------------------------
# RDS path: the model's names (model/fmSupportLib.py), float64
# --------------------------------------------------------------------------
RDS_TAPS = {"channel": 0, "carrier": 1, "pll_i": 2, "pll_q": 3, "resampled_i": 4, "rrc_i": 5, "rrc_q": 6, "pll_state": 7}


def rdsBandPass(N_taps, Fs, Fb, Fe) -> np.ndarray:
    """fmSupportLib.py:358 bandPass (Python argument order), float64."""
    h = np.zeros(N_taps)
    _check(lib.fmrx_rds_band_pass(N_taps, Fs, Fb, Fe, h))
    return h


def rdsImpResponse(N_taps, Fs, Fc) -> np.ndarray:
    h = np.zeros(N_taps)
    _check(lib.fmrx_rds_imp_response(N_taps, Fs, Fc, h))
    return h


def impulseResponseRootRaisedCosine(Fs, N_taps) -> np.ndarray:
    h = np.zeros(N_taps)
    _check(lib.fmrx_rds_rrc(Fs, N_taps, h))
    return h


def CDR(input1, rds_SPS, to_pass_on_state, block_count):
    """fmSupportLib.py:103 CDR -> (Manchester-decoded bits, [pair, next_start, size])."""
    x = np.ascontiguousarray(input1, np.float64)
    st = np.array([to_pass_on_state[0][0], to_pass_on_state[0][1], to_pass_on_state[1], to_pass_on_state[2]], np.float64)
    bits = np.zeros(len(x) // max(int(rds_SPS), 1) + 4, np.uint8)
    n = _sz(0)
    _check(lib.fmrx_rds_cdr(x, len(x), rds_SPS, block_count, st, bits, C.byref(n)))
    return bits[:n.value].astype(np.float64), [st[:2].copy(), int(st[2]), int(st[3])]


def diff_decoding(manch_data) -> np.ndarray:
    b = np.ascontiguousarray(manch_data, np.uint8)
    out = np.zeros(len(b), np.uint8)
    _check(lib.fmrx_rds_diff_decode(b, len(b), out))
    return out.astype(np.float64)


def framesync(diff_data):
    b = np.ascontiguousarray(diff_data, np.uint8)
    off, idx = C.create_string_buffer(8), _sz(0)
    _check(lib.fmrx_rds_frame_sync(b, len(b), off, C.byref(idx)))
    return off.value.decode(), idx.value


class Rds:
    """The RDS chain of model/fmMonoBlock.py:238-296 on fm_demod blocks (fmrx_rds_*)."""

    def __init__(self, mode=0, max_block=9600, device=0, params: RdsParams | None = None):
        self.params = params if params is not None else RdsParams()
        if params is None:
            _check(lib.fmrx_rds_mode_params(mode, C.byref(self.params)))
        self._h = _vp()
        _check(lib.fmrx_rds_create(C.byref(self._h), C.byref(self.params), max_block, device))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.fmrx_rds_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _check(lib.fmrx_rds_reset(self._h))

    def process(self, fm_demod):
        x = _f32(fm_demod)
        no = lib.fmrx_rds_n_out(self._h, len(x))
        yi, yq = np.zeros(no), np.zeros(no)
        bits, nb, off = np.zeros(no // max(self.params.sps, 1) + 4, np.uint8), _sz(0), C.create_string_buffer(8)
        _check(lib.fmrx_rds_process(self._h, x, len(x), yi.ctypes.data, yq.ctypes.data, bits.ctypes.data, C.byref(nb), off))
        return {"rrc_i": yi, "rrc_q": yq, "diff_bits": bits[:nb.value].copy(), "offset_type": off.value.decode()}

    def process_dev(self, d_demod_ptr, n, stream=None):
        _check(lib.fmrx_rds_process_dev(self._h, d_demod_ptr, n, stream))

    def read_tap(self, name) -> np.ndarray:
        n = _sz(0)
        _check(lib.fmrx_rds_read_tap(self._h, RDS_TAPS[name], None, C.byref(n)))
        out = np.zeros(n.value)
        _check(lib.fmrx_rds_read_tap(self._h, RDS_TAPS[name], out.ctypes.data, C.byref(n)))
        return out


class FrontEndPlan:
    """Reusable device-side tap tables for the fused front-end kernel (fmrx_fe_plan)."""

    def __init__(self, h, decim):
        h = _f32(h)
        self.taps, self.decim = len(h), int(decim)
        self._h = _vp()
        _check(lib.fmrx_fe_plan_create(C.byref(self._h), h, len(h), decim))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.fmrx_fe_plan_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def specialised(self) -> bool:
        return bool(lib.fmrx_fe_plan_is_specialised(self._h))

    @property
    def history_bytes(self) -> int:
        return lib.fmrx_fe_plan_history_bytes(self._h)

    def run_dev(self, d_iq_ptr, n_samples, d_hist_ptr, d_if_ptr, force_generic=False, stream=None):
        _check(lib.fmrx_fe_run_dev(self._h, d_iq_ptr, n_samples, d_hist_ptr, d_if_ptr, int(force_generic), stream))
