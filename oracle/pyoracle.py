"""ctypes front ends for the CPU oracle and (build container only) the compiled reference.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; nothing under qpsk_amd/ does.

  Oracle()          -> oracle/libqpsk_oracle.so   (oracle/qpsk_oracle.c, the restatement)
  Reference(name)   -> oracle/_ref/libqpsk_ref_<name>.so (oracle/ref_harness.c around the
                       untouched /root/reference sources; one library per FS/RS/FRAME_SIZE
                       variant because the reference fixes them with #defines, qpsk.h:16-23)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libqpsk_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

TAU = 2.0 * 3.14159265358979323846
TIMING_HIST, TIMING_FIXED, TIMING_FFT = 0, 1, 2

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build_oracle():
    """(Re)build oracle/libqpsk_oracle.so with gcc; also oracle/_ref when /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _opt(a, dtype):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Costas(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("phase", "freq", "max_freq", "min_freq", "damping", "loop_bw", "alpha", "beta")]


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build_oracle()
        L = self.lib = C.CDLL(path)
        L.qo_rrc_make.argtypes = [C.c_float, C.c_float, C.c_float, _f32p]
        L.qo_rrc_fir.argtypes = [_f32p, _f32p, _f32p, C.c_int]
        L.qo_costas_create.argtypes = [C.POINTER(Costas), C.c_float, C.c_float, C.c_float]
        L.qo_phase_detector.argtypes = [C.c_float, C.c_float]
        L.qo_phase_detector.restype = C.c_float
        for n in ("qo_update_gains", "qo_phase_wrap", "qo_frequency_limit"):
            getattr(L, n).argtypes = [C.POINTER(Costas)]
        for n in ("qo_advance_loop", "qo_set_loop_bandwidth", "qo_set_damping_factor", "qo_set_alpha",
                  "qo_set_beta", "qo_set_frequency", "qo_set_phase"):
            getattr(L, n).argtypes = [C.POINTER(Costas), C.c_float]
        L.qo_costas_step.argtypes = [C.POINTER(Costas), C.c_float, C.c_float,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.qo_demod.argtypes = [C.c_float, C.c_float]
        L.qo_timing_index.argtypes = [_f32p, C.c_int, C.c_int]
        L.qo_timing_hist.argtypes = [_f32p, C.c_int, C.c_int, _i32p]
        L.qo_timing_fft_index.argtypes = [_f32p, _f32p, C.c_int, C.c_int]
        L.qo_modem_new.argtypes = [C.c_double, C.c_double, C.c_int, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int, C.c_int]
        L.qo_modem_new.restype = C.c_void_p
        L.qo_modem_free.argtypes = [C.c_void_p]
        L.qo_modem_reset.argtypes = [C.c_void_p]
        L.qo_modem_set_mixer.argtypes = [C.c_void_p, _f32p]
        L.qo_mixer_from_hz.argtypes = [C.c_double, C.c_double, _f32p]
        L.qo_rx_frame_pcm.argtypes = [C.c_void_p, _i16p]
        L.qo_rx_frame_cplx.argtypes = [C.c_void_p, _f32p]
        L.qo_rx_batch.argtypes = [C.c_double, C.c_double, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int, C.c_int, _f32p, C.c_int] + [C.c_void_p] * 6 + [C.c_int]
        L.qo_rx_batch_bw.argtypes = [C.c_double, C.c_double, C.c_int, C.c_float, _f32p, C.c_int, C.c_float,
                                     C.c_float, C.c_int, C.c_int, _f32p, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
        L.qo_fftn.argtypes = [_f64p, _f64p, C.c_int]
        L.qo_ifftn.argtypes = [_f64p, _f64p, C.c_int]
        L.qo_crc16.argtypes = [_u8p, C.c_int]
        L.qo_crc16.restype = C.c_uint16
        L.qo_interleave.argtypes = [_u8p, C.c_int, C.c_int]
        L.qo_scramble_init.argtypes = [C.POINTER(C.c_uint16)]
        L.qo_scramble.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)]
        L.qo_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.qo_cabsf.argtypes = [C.c_float, C.c_float]
        L.qo_cabsf.restype = C.c_float
        L.qo_tx_init.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_float, C.c_double]
        L.qo_tx_symbols.argtypes = [C.c_void_p, _i16p, _i32p, C.c_int]

    # ---- free functions
    def rrc_make(self, fs, rs, alpha):
        t = np.zeros(127, np.float32)
        self.lib.qo_rrc_make(fs, rs, alpha, t)
        return t

    def rrc_fir(self, taps, memory, sample):
        """in place on memory (127,2) and sample (n,2) float32, like rrc_fir()"""
        self.lib.qo_rrc_fir(taps, memory.reshape(-1), sample.reshape(-1), sample.size // 2)

    def timing_index(self, filtered, cycles):
        return self.lib.qo_timing_index(filtered.reshape(-1), filtered.size // 2, cycles)

    def timing_hist(self, filtered, cycles):
        """-> (index, hist_i + hist_q as int32[8])"""
        h = np.zeros(8, np.int32)
        idx = self.lib.qo_timing_hist(filtered.reshape(-1), filtered.size // 2, cycles, h)
        return idx, h

    def timing_fft_index(self, taps, frame, cycles):
        return self.lib.qo_timing_fft_index(np.ascontiguousarray(taps, np.float32), np.ascontiguousarray(frame, np.float32).reshape(-1),
                                            frame.size // 2, cycles)

    def sincosf(self, x):
        s, c = C.c_float(), C.c_float()
        self.lib.qo_sincosf(x, C.byref(s), C.byref(c))
        return np.float32(s.value), np.float32(c.value)

    def fftn(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.empty_like(x)
        self.lib.qo_fftn(x.view(np.float64), out.view(np.float64), x.size)
        return out

    def ifftn(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.empty_like(x)
        self.lib.qo_ifftn(x.view(np.float64), out.view(np.float64), x.size)
        return out

    def crc16(self, data):
        d = np.frombuffer(bytes(data), np.uint8).copy()
        return int(self.lib.qo_crc16(d, d.size))

    def interleave(self, data, direction):
        d = np.array(data, np.uint8)
        self.lib.qo_interleave(d, d.size, direction)
        return d

    def scramble_stream(self, syms):
        mem = C.c_uint16()
        self.lib.qo_scramble_init(C.byref(mem))
        out = np.array(syms, np.uint8)
        for i in range(out.size):
            v = C.c_uint8(int(out[i]))
            self.lib.qo_scramble(C.byref(v), C.byref(mem))
            out[i] = v.value
        return out

    def rx_batch(self, frames, fs, rs, rrc_alpha=0.35, loop_bw=np.float32(TAU / 100.0), min_freq=-1.0,
                 max_freq=1.0, timing_mode=TIMING_FIXED, fixed_index=0, want_costas=False, threads=0):
        """frames: (F, L, 2) float32.  Returns dict(sym, freq, phase, index, hz[, costas])."""
        frames = np.ascontiguousarray(frames, np.float32)
        F, L = frames.shape[0], frames.shape[1]
        N = L // int(fs / rs)
        out = dict(sym=np.zeros((F, N), np.uint8), freq=np.zeros(F, np.float32), phase=np.zeros(F, np.float32),
                   index=np.zeros(F, np.int32), hz=np.zeros(F, np.float32))
        if want_costas:
            out["costas"] = np.zeros((F, N, 2), np.float32)
        self.lib.qo_rx_batch(fs, rs, L, rrc_alpha, loop_bw, min_freq, max_freq, timing_mode, fixed_index,
                             frames.reshape(-1), F, _opt(out["sym"], np.uint8), _opt(out["freq"], np.float32),
                             _opt(out["phase"], np.float32), _opt(out.get("costas"), np.float32),
                             _opt(out["index"], np.int32), _opt(out["hz"], np.float32), threads)
        return out

    def rx_batch_bw(self, frames, fs, rs, loop_bws, rrc_alpha=0.35, min_freq=-1.0, max_freq=1.0,
                    timing_mode=TIMING_FIXED, fixed_index=0, threads=0):
        frames = np.ascontiguousarray(frames, np.float32)
        bws = np.ascontiguousarray(loop_bws, np.float32)
        F, L, B = frames.shape[0], frames.shape[1], bws.size
        N = L // int(fs / rs)
        out = dict(sym=np.zeros((F, B, N), np.uint8), freq=np.zeros((F, B), np.float32),
                   phase=np.zeros((F, B), np.float32), index=np.zeros(F, np.int32))
        self.lib.qo_rx_batch_bw(fs, rs, L, rrc_alpha, bws, B, min_freq, max_freq, timing_mode, fixed_index,
                                frames.reshape(-1), F, _opt(out["sym"], np.uint8), _opt(out["freq"], np.float32),
                                _opt(out["phase"], np.float32), _opt(out["index"], np.int32), threads)
        return out

    def modem(self, fs, rs, frame_size, **kw):
        return OracleModem(self, fs, rs, frame_size, **kw)

    def tx(self, fs, rs, rrc_alpha, tx_hz):
        return OracleTx(self, fs, rs, rrc_alpha, tx_hz)


class _ModemStruct(C.Structure):
    _fields_ = [("fs", C.c_double), ("rs", C.c_double), ("cycles", C.c_int), ("frame_size", C.c_int),
                ("nsym", C.c_int), ("timing_mode", C.c_int), ("fixed_index", C.c_int),
                ("taps", C.c_float * 127), ("loop", Costas), ("rx_filter", C.c_float * 254),
                ("mix_phase", C.c_float * 2), ("mix_rect", C.c_float * 2), ("offset_hz", C.c_float),
                ("last_index", C.c_int), ("input_frame", C.POINTER(C.c_float)),
                ("decimated", C.POINTER(C.c_float)), ("costas_frame", C.POINTER(C.c_float)),
                ("symbols", C.POINTER(C.c_uint8))]


class OracleModem:
    """One streaming modem instance == the reference's globals (qpsk.c:36-53, costas_loop.c:13-23)."""

    def __init__(self, orc, fs, rs, frame_size, rrc_alpha=0.35, loop_bw=np.float32(TAU / 100.0), min_freq=-1.0,
                 max_freq=1.0, timing_mode=TIMING_HIST, fixed_index=0):
        self.o = orc
        self.p = orc.lib.qo_modem_new(fs, rs, frame_size, rrc_alpha, loop_bw, min_freq, max_freq, timing_mode,
                                      fixed_index)
        self.s = C.cast(self.p, C.POINTER(_ModemStruct)).contents

    def __del__(self):
        if getattr(self, "p", None):
            self.o.lib.qo_modem_free(self.p)
            self.p = None

    def reset(self):
        self.o.lib.qo_modem_reset(self.p)

    def set_mixer(self, phase_rect4):
        self.o.lib.qo_modem_set_mixer(self.p, np.ascontiguousarray(phase_rect4, np.float32))

    def set_mixer_hz(self, hz):
        r = np.zeros(2, np.float32)
        self.o.lib.qo_mixer_from_hz(hz, self.s.fs, r)
        self.set_mixer(np.array([1.0, 0.0, r[0], r[1]], np.float32))

    def rx_pcm(self, pcm):
        self.o.lib.qo_rx_frame_pcm(self.p, np.ascontiguousarray(pcm, np.int16))

    def rx_cplx(self, x):
        self.o.lib.qo_rx_frame_cplx(self.p, np.ascontiguousarray(x, np.float32).reshape(-1))

    def _arr(self, ptr, n, dt=np.float32):
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy().astype(dt, copy=False)

    @property
    def nsym(self): return self.s.nsym
    @property
    def input_frame(self): return self._arr(self.s.input_frame, 2 * self.s.frame_size).reshape(-1, 2)
    @property
    def decimated(self): return self._arr(self.s.decimated, 4 * self.s.nsym).reshape(-1, 2)
    @property
    def costas_frame(self): return self._arr(self.s.costas_frame, 2 * self.s.nsym).reshape(-1, 2)
    @property
    def symbols(self): return self._arr(self.s.symbols, self.s.nsym, np.uint8)
    @property
    def rx_filter(self): return np.array(self.s.rx_filter, np.float32).reshape(-1, 2)
    @property
    def mixer(self): return np.array(list(self.s.mix_phase) + list(self.s.mix_rect), np.float32)
    @property
    def phase(self): return np.float32(self.s.loop.phase)
    @property
    def freq(self): return np.float32(self.s.loop.freq)
    @property
    def offset_hz(self): return np.float32(self.s.offset_hz)
    @property
    def index(self): return self.s.last_index
    @property
    def taps(self): return np.array(self.s.taps, np.float32)


class _TxStruct(C.Structure):
    _fields_ = [("taps", C.c_float * 127), ("tx_filter", C.c_float * 254), ("phase", C.c_float * 2),
                ("rect", C.c_float * 2), ("cycles", C.c_int)]


class OracleTx:
    def __init__(self, orc, fs, rs, rrc_alpha, tx_hz):
        self.o = orc
        self.s = _TxStruct()
        orc.lib.qo_tx_init(C.byref(self.s), fs, rs, rrc_alpha, tx_hz)

    def symbols(self, bits):
        bits = np.ascontiguousarray(bits, np.int32)
        nsym = bits.size // 2
        out = np.zeros(nsym * self.s.cycles, np.int16)
        self.o.lib.qo_tx_symbols(C.byref(self.s), out, bits, nsym)
        return out


# --------------------------------------------------------------------------- reference
def ref_available(name):
    return os.path.exists(os.path.join(REF_DIR, "libqpsk_ref_%s.so" % name))


class Reference:
    """The reference itself (one variant).  Holds the reference's process-global state."""

    def __init__(self, name):
        path = os.path.join(REF_DIR, "libqpsk_ref_%s.so" % name)
        L = self.lib = C.CDLL(path)
        fs, rs, fsz, cyc, nt = C.c_double(), C.c_double(), C.c_int(), C.c_int(), C.c_int()
        L.ref_params(C.byref(fs), C.byref(rs), C.byref(fsz), C.byref(cyc), C.byref(nt))
        self.fs, self.rs, self.frame_size, self.cycles, self.ntaps = fs.value, rs.value, fsz.value, cyc.value, nt.value
        self.nsym = self.frame_size // self.cycles
        L.ref_reset.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_double, C.c_double]
        L.ref_rrc_make.argtypes = [C.c_float, C.c_float, C.c_float]
        L.ref_get_taps.argtypes = [_f32p]
        L.ref_rrc_fir.argtypes = [_f32p, _f32p, C.c_int]
        L.ref_rx_frame_pcm.argtypes = [_i16p]
        L.ref_rx_frame_cplx.argtypes = [_f32p]
        for n in ("ref_get_input_frame", "ref_get_rx_filter", "ref_get_decimated", "ref_set_decimated",
                  "ref_get_costas", "ref_get_mixer", "ref_set_mixer", "ref_costas_state"):
            getattr(L, n).argtypes = [_f32p]
        for n in ("ref_get_phase", "ref_get_freq", "ref_get_alpha", "ref_get_beta", "ref_get_offset_hz"):
            getattr(L, n).restype = C.c_float
        L.ref_demod.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_int)]
        L.ref_get_symbols.argtypes = [_u8p]
        L.ref_phase_detector.argtypes = [C.c_float, C.c_float]
        L.ref_phase_detector.restype = C.c_float
        L.ref_tx_symbols.argtypes = [_i16p, _i32p, C.c_int]
        L.ref_tx_baseband.argtypes = [_f32p, _i32p, C.c_int]
        L.ref_fftn.argtypes = [_f64p, _f64p, C.c_int]
        L.ref_ifftn.argtypes = [_f64p, _f64p, C.c_int]
        L.ref_fft.argtypes = [_f64p, _f64p]
        L.ref_ifft.argtypes = [_f64p, _f64p]
        L.ref_crc16.argtypes = [_u8p, C.c_int]
        L.ref_crc16.restype = C.c_uint16
        L.ref_interleave.argtypes = [_u8p, C.c_int, C.c_int]
        L.ref_scramble.argtypes = [C.POINTER(C.c_uint8), C.c_int]
        # the costas_loop.h API is exported by name from the same library
        for n in ("advance_loop", "set_loop_bandwidth", "set_damping_factor", "set_alpha", "set_beta",
                  "set_frequency", "set_phase", "set_max_freq", "set_min_freq"):
            getattr(L, n).argtypes = [C.c_float]
        L.create_control_loop.argtypes = [C.c_float, C.c_float, C.c_float]

    def reset(self, loop_bw=np.float32(TAU / 100.0), min_freq=-1.0, max_freq=1.0, rrc_alpha=0.35, tx_hz=1550.0,
              rx_hz=1500.0):
        self.lib.ref_reset(loop_bw, min_freq, max_freq, rrc_alpha, tx_hz, rx_hz)

    def taps(self, fs=None, rs=None, alpha=None):
        if fs is not None:
            self.lib.ref_rrc_make(fs, rs, alpha)
        t = np.zeros(127, np.float32)
        self.lib.ref_get_taps(t)
        return t

    def rrc_fir(self, memory, sample):
        self.lib.ref_rrc_fir(memory.reshape(-1), sample.reshape(-1), sample.size // 2)

    def rx_pcm(self, pcm):
        self.lib.ref_rx_frame_pcm(np.ascontiguousarray(pcm, np.int16))

    def rx_cplx(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        assert x.size == 2 * self.frame_size
        self.lib.ref_rx_frame_cplx(x)

    def _get(self, fn, n):
        a = np.zeros(n, np.float32)
        getattr(self.lib, fn)(a)
        return a

    @property
    def input_frame(self): return self._get("ref_get_input_frame", 2 * self.frame_size).reshape(-1, 2)
    @property
    def rx_filter(self): return self._get("ref_get_rx_filter", 254).reshape(-1, 2)
    @property
    def decimated(self): return self._get("ref_get_decimated", 4 * self.nsym).reshape(-1, 2)
    @property
    def costas_frame(self): return self._get("ref_get_costas", 2 * self.nsym).reshape(-1, 2)
    @property
    def mixer(self): return self._get("ref_get_mixer", 4)
    @property
    def costas_state(self): return self._get("ref_costas_state", 8)
    @property
    def phase(self): return np.float32(self.lib.ref_get_phase())
    @property
    def freq(self): return np.float32(self.lib.ref_get_freq())
    @property
    def offset_hz(self): return np.float32(self.lib.ref_get_offset_hz())

    @property
    def symbols(self):
        s = np.zeros(self.nsym, np.uint8)
        self.lib.ref_get_symbols(s)
        return s

    def set_decimated(self, d):
        self.lib.ref_set_decimated(np.ascontiguousarray(d, np.float32).reshape(-1))

    def set_mixer(self, m):
        self.lib.ref_set_mixer(np.ascontiguousarray(m, np.float32))

    def demod(self, re, im):
        b = (C.c_int * 2)()
        self.lib.ref_demod(re, im, b)
        return (b[1] << 1) | b[0]

    def phase_detector(self, re, im):
        return np.float32(self.lib.ref_phase_detector(re, im))

    def tx_symbols(self, bits):
        bits = np.ascontiguousarray(bits, np.int32)
        nsym = bits.size // 2
        out = np.zeros(nsym * self.cycles, np.int16)
        n = self.lib.ref_tx_symbols(out, bits, nsym)
        assert n == out.size
        return out

    def tx_baseband(self, bits):
        bits = np.ascontiguousarray(bits, np.int32)
        nsym = bits.size // 2
        out = np.zeros(nsym * self.cycles * 2, np.float32)
        self.lib.ref_tx_baseband(out, bits, nsym)
        return out.reshape(-1, 2)

    def fftn(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.empty_like(x)
        self.lib.ref_fftn(x.view(np.float64), out.view(np.float64), x.size)
        return out

    def ifftn(self, x):
        x = np.ascontiguousarray(x, np.complex128)
        out = np.empty_like(x)
        self.lib.ref_ifftn(x.view(np.float64), out.view(np.float64), x.size)
        return out

    def crc16(self, data):
        d = np.frombuffer(bytes(data), np.uint8).copy()
        return int(self.lib.ref_crc16(d, d.size))

    def interleave(self, data, direction):
        d = np.array(data, np.uint8)
        self.lib.ref_interleave(d, d.size, direction)
        return d

    def scramble_stream(self, syms, reg=0):
        self.lib.ref_scramble_init(reg)
        out = np.array(syms, np.uint8)
        for i in range(out.size):
            v = C.c_uint8(int(out[i]))
            self.lib.ref_scramble(C.byref(v), reg)
            out[i] = v.value
        return out

    def independent_frame(self, x, **reset_kw):
        """SURVEY 8(c) independent-frame pin: fresh state, rx_frame(frame), rx_frame(zeros)."""
        self.reset(**reset_kw)
        self.rx_cplx(x)
        self.rx_cplx(np.zeros(2 * self.frame_size, np.float32))
        return dict(sym=self.symbols, costas=self.costas_frame, phase=self.phase, freq=self.freq,
                    hz=self.offset_hz)
