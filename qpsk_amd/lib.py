"""ctypes binding of libqpsk_hip.so (include/qpsk_hip.h) for tests and bench.py.

Device buffers are torch tensors on the context's GPU; only their data_ptr() crosses the C ABI.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TAU = 2.0 * 3.14159265358979323846
TIMING_HIST, TIMING_FIXED, TIMING_FFT = 0, 1, 2
NTAPS = 127


class QpskError(RuntimeError):
    pass


class Params(C.Structure):
    """qpsk_params of include/qpsk_hip.h (the reference's #defines and main() literals)."""
    _fields_ = [("fs", C.c_double), ("rs", C.c_double), ("frame_size", C.c_int), ("rrc_alpha", C.c_float),
                ("loop_bw", C.c_float), ("min_freq", C.c_float), ("max_freq", C.c_float),
                ("timing_mode", C.c_int), ("fixed_index", C.c_int)]


def lib_path():
    """The built library; QPSK_HIP_LIB names another build of it (A/B timing of two source versions on one box)."""
    return os.environ.get("QPSK_HIP_LIB") or os.path.join(HERE, "libqpsk_hip.so")


def build(verbose=False):
    """Compile qpsk_amd/csrc for gfx950 into qpsk_amd/libqpsk_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return lib_path()


_LIB = None

# every symbol include/qpsk_hip.h declares (tests check that the library exports all of them)
API_SYMBOLS = [
    "qpsk_last_error", "qpsk_version", "qpsk_device_count", "qpsk_params_default", "qpsk_ctx_create",
    "qpsk_ctx_destroy", "qpsk_ctx_sync", "qpsk_ctx_set_stream", "qpsk_ctx_set_tuning", "qpsk_ctx_cycles", "qpsk_ctx_nsym",
    "qpsk_ctx_last_kernel",
    "qpsk_ctx_get_taps", "qpsk_ctx_get_gains", "qpsk_ctx_set_taps", "qpsk_ctx_set_loop", "qpsk_rx_batch", "qpsk_rx_batch_pitched",
    "qpsk_rx_batch_bw", "qpsk_rrc_fir_batch", "qpsk_rrc_fir_batch_fast", "qpsk_timing_hist_batch", "qpsk_timing_scan_batch", "qpsk_timing_fft_batch", "qpsk_timing_fft_bin_batch", "qpsk_costas_batch", "qpsk_fft_batch",
    "qpsk_streams_reset", "qpsk_streams_set_loop_state", "qpsk_streams_get_loop_state", "qpsk_streams_rx_cplx",
    "qpsk_streams_rx_pcm", "qpsk_streams_rx_pcm_host", "qpsk_dev_alloc", "qpsk_dev_free", "qpsk_dev_upload", "qpsk_dev_download",
    "qpsk_selftest_sincos_hash", "qpsk_crc16_batch", "qpsk_interleave_batch", "qpsk_scramble_batch",
    "qpsk_tx_reset", "qpsk_tx_symbols", "qpsk_test_inject_status", "qpsk_test_hist_state", "qpsk_ctx_check",
    "qpsk_multi_create", "qpsk_multi_destroy", "qpsk_multi_shards", "qpsk_multi_load", "qpsk_multi_shard", "qpsk_multi_use_device_input",
    "qpsk_multi_rx_begin", "qpsk_multi_rx_end", "qpsk_multi_set_direct_output", "qpsk_host_alloc", "qpsk_host_free",
    "qpsk_multi_set_packed", "qpsk_pack_symbols", "qpsk_unpack_symbols_host",
]
# every symbol include/qpsk_dropin.h declares
DROPIN_SYMBOLS = [
    "qpsk_dropin_configure", "qpsk_dropin_set_device", "qpsk_dropin_shutdown", "rrc_fir", "rrc_make",
    "create_control_loop", "phase_detector", "update_gains", "advance_loop", "phase_wrap", "frequency_limit",
    "set_loop_bandwidth", "set_damping_factor", "set_alpha", "set_beta", "set_frequency", "set_phase",
    "set_max_freq", "set_min_freq", "get_loop_bandwidth", "get_damping_factor", "get_alpha", "get_beta",
    "get_frequency", "get_phase", "get_max_freq", "get_min_freq", "fft", "fftn", "ifft", "ifftn", "qpsk_demod",
    "rx_frame", "qpsk_dropin_costas_frame", "qpsk_dropin_symbols", "qpsk_dropin_offset_freq",
    "qpsk_dropin_timing_index",
]


def load():
    """Load the library; raises QpskError if it has not been built (no fallback of any kind)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise QpskError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)" % p)
    # PyTorch ships its own libamdhip64; whichever HIP runtime is loaded first serves the whole process, and
    # torch.cuda stops working if the system one got in before it.  This plumbing always shares the process with
    # torch (device buffers are torch tensors), so let torch load its runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(p)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.qpsk_last_error.restype = C.c_char_p
    L.qpsk_version.restype = C.c_char_p
    L.qpsk_params_default.argtypes = [C.POINTER(Params)]
    L.qpsk_ctx_create.argtypes = [C.POINTER(vp), i32, C.POINTER(Params), vp]
    L.qpsk_ctx_destroy.argtypes = [vp]
    L.qpsk_ctx_destroy.restype = None
    L.qpsk_ctx_sync.argtypes = [vp]
    L.qpsk_ctx_set_stream.argtypes = [vp, vp]
    L.qpsk_ctx_set_tuning.argtypes = [vp, C.c_char_p, i32]
    L.qpsk_ctx_cycles.argtypes = [vp]
    L.qpsk_ctx_nsym.argtypes = [vp]
    L.qpsk_ctx_last_kernel.argtypes = [vp]
    L.qpsk_ctx_last_kernel.restype = C.c_char_p
    L.qpsk_ctx_get_taps.argtypes = [vp, C.POINTER(f32)]
    L.qpsk_ctx_get_gains.argtypes = [vp, C.POINTER(f32), C.POINTER(f32)]
    L.qpsk_ctx_set_taps.argtypes = [vp, C.POINTER(f32)]
    L.qpsk_ctx_set_loop.argtypes = [vp, f32, f32, f32, f32]
    L.qpsk_rx_batch.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.qpsk_rx_batch_pitched.argtypes = [vp, vp, C.c_longlong, i32, vp, vp, vp, vp, vp, vp]
    L.qpsk_rx_batch_bw.argtypes = [vp, vp, i32, C.POINTER(f32), i32, vp, vp, vp, vp]
    L.qpsk_rrc_fir_batch.argtypes = [vp, vp, vp, vp, i32, i32]
    L.qpsk_rrc_fir_batch_fast.argtypes = [vp, vp, vp, vp, i32, i32]
    L.qpsk_timing_hist_batch.argtypes = [vp, vp, i32, vp, vp]
    L.qpsk_timing_fft_batch.argtypes = [vp, vp, i32, vp, vp, vp]
    L.qpsk_timing_fft_bin_batch.argtypes = [vp, vp, i32, vp, vp, vp]
    L.qpsk_timing_scan_batch.argtypes = [vp, vp, i32, vp, vp]
    L.qpsk_costas_batch.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.qpsk_fft_batch.argtypes = [vp, vp, vp, i32, i32, i32]
    L.qpsk_streams_reset.argtypes = [vp, i32, C.c_double]
    L.qpsk_streams_set_loop_state.argtypes = [vp, C.POINTER(f32)]
    L.qpsk_streams_get_loop_state.argtypes = [vp, C.POINTER(f32)]
    L.qpsk_streams_rx_cplx.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.qpsk_streams_rx_pcm.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.qpsk_streams_rx_pcm_host.argtypes = [vp, vp, vp, vp, vp, vp]
    L.qpsk_selftest_sincos_hash.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong)]
    L.qpsk_test_inject_status.argtypes = [vp, i32]
    L.qpsk_ctx_check.argtypes = [vp]
    L.qpsk_test_hist_state.argtypes = [vp, C.POINTER(i32)]
    L.qpsk_multi_create.argtypes = [C.POINTER(vp), C.POINTER(i32), i32, C.POINTER(Params)]
    L.qpsk_multi_destroy.argtypes = [vp]
    L.qpsk_multi_destroy.restype = None
    L.qpsk_multi_shards.argtypes = [vp]
    L.qpsk_multi_load.argtypes = [vp, C.c_longlong, vp]
    L.qpsk_multi_shard.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(vp), C.POINTER(vp)]
    L.qpsk_multi_use_device_input.argtypes = [vp, i32, vp]
    L.qpsk_multi_set_direct_output.argtypes = [vp, i32, vp, vp, vp]
    L.qpsk_multi_set_packed.argtypes = [vp, i32]
    L.qpsk_pack_symbols.argtypes = [vp, vp, C.c_longlong, i32, vp]
    L.qpsk_unpack_symbols_host.argtypes = [vp, C.c_longlong, i32, vp]
    L.qpsk_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.qpsk_host_free.argtypes = [vp]
    L.qpsk_multi_rx_begin.argtypes = [vp, i32]
    L.qpsk_multi_rx_end.argtypes = [vp, i32, vp, vp, vp]
    L.qpsk_crc16_batch.argtypes = [vp, vp, i32, i32, vp]
    L.qpsk_interleave_batch.argtypes = [vp, vp, i32, i32, i32]
    L.qpsk_scramble_batch.argtypes = [vp, vp, i32, i32]
    L.qpsk_tx_reset.argtypes = [vp, i32, C.c_double]
    L.qpsk_tx_symbols.argtypes = [vp, vp, i32, vp, vp]
    _LIB = L
    return L


def version():
    """qpsk_version() of the loaded library"""
    return load().qpsk_version().decode()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Modem:
    """One qpsk_ctx: a configuration (taps, loop gains) bound to one GPU and one HIP stream."""

    def __init__(self, fs=9600.0, rs=2400.0, frame_size=512, rrc_alpha=0.35, loop_bw=np.float32(TAU / 100.0),
                 min_freq=-1.0, max_freq=1.0, timing_mode=TIMING_HIST, fixed_index=0, device=None, stream="torch"):
        import torch
        self.torch = torch
        self.L = load()
        if not torch.cuda.is_available():
            raise QpskError("no GPU visible to torch; libqpsk_hip has no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.dev = torch.device("cuda", self.device)
        self.params = Params(fs, rs, frame_size, rrc_alpha, loop_bw, min_freq, max_freq, timing_mode, fixed_index)
        if stream == "torch":
            s = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        elif stream is None:
            s = None
        else:
            s = C.c_void_p(int(stream))
        h = C.c_void_p()
        self._check(self.L.qpsk_ctx_create(C.byref(h), self.device, C.byref(self.params), s))
        self.h = h
        self.cycles = self.L.qpsk_ctx_cycles(h)
        self.nsym = self.L.qpsk_ctx_nsym(h)
        self.frame_size = frame_size
        self.nstreams = 0

    def _check(self, rc):
        if rc != 0:
            raise QpskError("libqpsk_hip error %d: %s" % (rc, self.L.qpsk_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.qpsk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.L.qpsk_ctx_sync(self.h))

    def last_kernel(self):
        """Name of the receive kernel the last rx_batch*() call launched (which geometry served that shape)."""
        return self.L.qpsk_ctx_last_kernel(self.h).decode()

    def tune(self, **kw):
        """Kernel-geometry selection (tests, measurements): m.tune(pipe_nf=3, pipe_wide=0); None = library's choice.
        Keys are the QPSK_* names of qpsk_ctx_set_tuning() in lower case without the prefix."""
        for k, v in kw.items():
            self._check(self.L.qpsk_ctx_set_tuning(self.h, ("QPSK_" + k.upper()).encode(), -1 if v is None else int(v)))

    def set_stream(self, stream):
        """Enqueue this context's work on another HIP stream (a torch.cuda.Stream or a raw handle; None = default)."""
        s = None if stream is None else C.c_void_p(getattr(stream, "cuda_stream", stream))
        self._check(self.L.qpsk_ctx_set_stream(self.h, s))

    # ---- configuration
    @property
    def taps(self):
        t = (C.c_float * NTAPS)()
        self._check(self.L.qpsk_ctx_get_taps(self.h, t))
        return np.array(t, np.float32)

    @property
    def gains(self):
        a, b = C.c_float(), C.c_float()
        self._check(self.L.qpsk_ctx_get_gains(self.h, C.byref(a), C.byref(b)))
        return np.float32(a.value), np.float32(b.value)

    def set_taps(self, taps):
        t = (C.c_float * NTAPS)(*[float(x) for x in taps])
        self._check(self.L.qpsk_ctx_set_taps(self.h, t))

    def set_loop(self, alpha, beta, min_freq, max_freq):
        self._check(self.L.qpsk_ctx_set_loop(self.h, alpha, beta, min_freq, max_freq))

    # ---- helpers
    def _dev(self, a, dtype):
        t = self.torch
        if isinstance(a, np.ndarray):
            a = t.from_numpy(np.ascontiguousarray(a))
        a = a.to(self.dev)
        assert a.dtype == dtype, (a.dtype, dtype)
        return a.contiguous()

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.dev)

    # ---- the hot path
    def rx_batch(self, frames, want_costas=False, out=None):
        """frames: (F, frame_size, 2) float32 (numpy or torch).  Returns dict of torch tensors."""
        t = self.torch
        x = self._dev(frames, t.float32)
        F = x.shape[0]
        assert x.shape[1] == self.frame_size and x.shape[2] == 2
        o = out or dict(sym=self.empty((F, self.nsym), t.uint8), freq=self.empty((F,), t.float32),
                        phase=self.empty((F,), t.float32), index=self.empty((F,), t.int32),
                        hz=self.empty((F,), t.float32),
                        costas=self.empty((F, self.nsym, 2), t.float32) if want_costas else None)
        self._check(self.L.qpsk_rx_batch(self.h, _ptr(x), F, _ptr(o["sym"]), _ptr(o["freq"]), _ptr(o["phase"]),
                                         _ptr(o.get("costas")), _ptr(o.get("index")), _ptr(o.get("hz"))))
        return o

    def rx_batch_raw(self, x, F, sym, freq, phase, pitch=0, index=None):
        """No allocation, no checks: the call bench.py times.  pitch: complex samples between frame starts (0 = packed);
        index: optional (F,) int32 tensor receiving the timing indices."""
        if pitch:
            rc = self.L.qpsk_rx_batch_pitched(self.h, _ptr(x), pitch, F, _ptr(sym), _ptr(freq), _ptr(phase), None, _ptr(index), None)
        else:
            rc = self.L.qpsk_rx_batch(self.h, _ptr(x), F, _ptr(sym), _ptr(freq), _ptr(phase), None, _ptr(index), None)
        if rc:
            self._check(rc)

    def rx_batch_bw(self, frames, loop_bws):
        t = self.torch
        x = self._dev(frames, t.float32)
        F, B = x.shape[0], len(loop_bws)
        bws = (C.c_float * B)(*[float(b) for b in loop_bws])
        o = dict(sym=self.empty((F, B, self.nsym), t.uint8), freq=self.empty((F, B), t.float32),
                 phase=self.empty((F, B), t.float32), index=self.empty((F,), t.int32))
        self._check(self.L.qpsk_rx_batch_bw(self.h, _ptr(x), F, bws, B, _ptr(o["sym"]), _ptr(o["freq"]),
                                            _ptr(o["phase"]), _ptr(o["index"])))
        return o

    # ---- stages
    def rrc_fir(self, x, memory=None, fast=False):
        """x: (F, n, 2); memory: (F, 127, 2) updated in place (torch tensor) or None.  fast: the overlap-save filter
        (qpsk_rrc_fir_batch_fast: ~1e-6 of the peak from the exact one, not bit for bit)."""
        t = self.torch
        x = self._dev(x, t.float32)
        y = t.empty_like(x)
        fn = self.L.qpsk_rrc_fir_batch_fast if fast else self.L.qpsk_rrc_fir_batch
        self._check(fn(self.h, _ptr(memory), _ptr(x), _ptr(y), x.shape[0], x.shape[1]))
        return y

    def timing_hist(self, filtered, want_hist=False):
        t = self.torch
        y = self._dev(filtered, t.float32)
        idx = self.empty((y.shape[0],), t.int32)
        hist = self.empty((y.shape[0], 8), t.int32) if want_hist else None
        self._check(self.L.qpsk_timing_hist_batch(self.h, _ptr(y), y.shape[0], _ptr(idx), _ptr(hist)))
        return (idx, hist) if want_hist else idx

    def timing_scan(self, frames):
        """histogram timing estimate straight from unfiltered (F, frame_size, 2) frames -> (index (F,), hist (F, 8))"""
        t = self.torch
        x = self._dev(frames, t.float32)
        idx = self.empty((x.shape[0],), t.int32)
        hist = self.empty((x.shape[0], 8), t.int32)
        self._check(self.L.qpsk_timing_scan_batch(self.h, _ptr(x), x.shape[0], _ptr(idx), _ptr(hist)))
        return idx, hist

    def timing_fft(self, frames, want_internals=False):
        """FFT timing estimate of (F, frame_size, 2) frames -> index (F,) [, filtered (F, 512, 2), spectrum (F, 512) complex128]"""
        t = self.torch
        x = self._dev(frames, t.float32)
        F = x.shape[0]
        idx = self.empty((F,), t.int32)
        y = self.empty((F, 512, 2), t.float32) if want_internals else None
        X = self.empty((F, 512), t.complex128) if want_internals else None
        self._check(self.L.qpsk_timing_fft_batch(self.h, _ptr(x), F, _ptr(idx), _ptr(y), _ptr(X)))
        return (idx, y, X) if want_internals else idx

    def timing_fft_bin(self, frames):
        """the estimate as qpsk_rx_batch runs it (pruned transform) -> index (F,), filtered (F, 512, 2), bin (F,) complex128"""
        t = self.torch
        x = self._dev(frames, t.float32)
        F = x.shape[0]
        idx = self.empty((F,), t.int32)
        y = self.empty((F, 512, 2), t.float32)
        xk = self.empty((F,), t.complex128)
        self._check(self.L.qpsk_timing_fft_bin_batch(self.h, _ptr(x), F, _ptr(idx), _ptr(y), _ptr(xk)))
        return idx, y, xk

    def costas(self, d, state=None, want_costas=True):
        t = self.torch
        d = self._dev(d, t.float32)
        F, N = d.shape[0], d.shape[1]
        sym = self.empty((F, N), t.uint8)
        z = self.empty((F, N, 2), t.float32) if want_costas else None
        self._check(self.L.qpsk_costas_batch(self.h, _ptr(d), F, N, _ptr(state), _ptr(sym), _ptr(z)))
        return sym, z

    def fft(self, x, inverse=False):
        """x: (B, n) complex128 numpy/torch -> torch complex128"""
        t = self.torch
        if isinstance(x, np.ndarray):
            x = t.from_numpy(np.ascontiguousarray(x, dtype=np.complex128))
        x = x.to(self.dev).contiguous()
        out = t.empty_like(x)
        self._check(self.L.qpsk_fft_batch(self.h, _ptr(x), _ptr(out), x.shape[0], x.shape[1], int(inverse)))
        return out

    # ---- streams
    def streams_reset(self, nstreams, mixer_hz=1500.0):
        self._check(self.L.qpsk_streams_reset(self.h, nstreams, mixer_hz))
        self.nstreams = nstreams

    def _stream_out(self, want_costas):
        t, n = self.torch, self.nstreams
        return dict(sym=self.empty((n, self.nsym), t.uint8), freq=self.empty((n,), t.float32),
                    phase=self.empty((n,), t.float32), index=self.empty((n,), t.int32),
                    costas=self.empty((n, self.nsym, 2), t.float32) if want_costas else None)

    def streams_rx_cplx(self, blocks, want_costas=True):
        x = self._dev(blocks, self.torch.float32)
        o = self._stream_out(want_costas)
        self._check(self.L.qpsk_streams_rx_cplx(self.h, _ptr(x), _ptr(o["sym"]), _ptr(o["freq"]), _ptr(o["phase"]),
                                                _ptr(o["costas"]), _ptr(o["index"])))
        return o

    def streams_rx_pcm(self, pcm, want_costas=True):
        x = self._dev(pcm, self.torch.int16)
        o = self._stream_out(want_costas)
        self._check(self.L.qpsk_streams_rx_pcm(self.h, _ptr(x), _ptr(o["sym"]), _ptr(o["freq"]), _ptr(o["phase"]),
                                               _ptr(o["costas"]), _ptr(o["index"])))
        return o

    def streams_loop_state(self):
        a = (C.c_float * (2 * self.nstreams))()
        self._check(self.L.qpsk_streams_get_loop_state(self.h, a))
        return np.array(a, np.float32).reshape(-1, 2)

    # ---- bit-level stages (algorithms/ of the reference)
    # ---- transmitters (qpsk.c:225-285)
    def tx_reset(self, nstreams, tx_hz=1550.0):
        self._check(self.L.qpsk_tx_reset(self.h, nstreams, tx_hz))
        self.ntx = nstreams

    def tx_symbols(self, symbols, want_pcm=True, want_baseband=False):
        """symbols: (ntx, nsym) uint8 dibits -> dict(pcm (ntx, nsym*CYCLES) int16, baseband (ntx, nsym*CYCLES, 2))"""
        t = self.torch
        d = self._dev(symbols, t.uint8)
        if d.dim() != 2 or d.shape[0] != getattr(self, "ntx", 0):
            raise ValueError("symbols must be (ntx, nsym) after tx_reset(ntx)")
        n = d.shape[1] * self.cycles
        pcm = self.empty((d.shape[0], n), t.int16) if want_pcm else None
        bb = self.empty((d.shape[0], n, 2), t.float32) if want_baseband else None
        self._check(self.L.qpsk_tx_symbols(self.h, _ptr(d), d.shape[1], _ptr(pcm), _ptr(bb)))
        return dict(pcm=pcm, baseband=bb)

    def crc16(self, packets):
        """packets: (P, nbytes) uint8 -> (P,) uint16 (torch int16 view returned as numpy uint16)"""
        t = self.torch
        d = self._dev(packets, t.uint8)
        out = self.empty((d.shape[0],), t.int16)
        self._check(self.L.qpsk_crc16_batch(self.h, _ptr(d), d.shape[0], d.shape[1], _ptr(out)))
        return out.cpu().numpy().view(np.uint16)

    def interleave(self, packets, direction):
        d = self._dev(packets, self.torch.uint8).clone()
        self._check(self.L.qpsk_interleave_batch(self.h, _ptr(d), d.shape[0], d.shape[1], int(direction)))
        return d

    def scramble(self, symbols):
        d = self._dev(symbols, self.torch.uint8).clone()
        self._check(self.L.qpsk_scramble_batch(self.h, _ptr(d), d.shape[0], d.shape[1]))
        return d

    def sincos_hash(self, first, count):
        h = C.c_ulonglong()
        self._check(self.L.qpsk_selftest_sincos_hash(self.h, first, count, C.byref(h)))
        return h.value


class MultiJob:
    """qpsk_multi of include/qpsk_hip.h: a batch sharded over several devices (repeats allowed), one context + one host thread + two
    streams per shard, results gathered into host arrays.  Plumbing for the tests and bench.py's `gather` key."""

    def __init__(self, devices, **params):
        kw = dict(fs=9600.0, rs=2400.0, frame_size=512, rrc_alpha=0.35, loop_bw=np.float32(TAU / 100.0), min_freq=-1.0, max_freq=1.0,
                  timing_mode=TIMING_HIST, fixed_index=0)
        kw.update(params)
        self.L = load()
        self.params = Params(kw["fs"], kw["rs"], kw["frame_size"], kw["rrc_alpha"], kw["loop_bw"], kw["min_freq"], kw["max_freq"],
                             kw["timing_mode"], kw["fixed_index"])
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        self._check(self.L.qpsk_multi_create(C.byref(h), devs, len(devices), C.byref(self.params)))
        self.h = h
        self.frame_size = kw["frame_size"]
        self.total = 0
        self.nsym = None

    def _check(self, rc):
        if rc != 0:
            raise QpskError("libqpsk_hip error %d: %s" % (rc, self.L.qpsk_last_error().decode()))

    def load(self, frames_host=None, total=None):
        """frames_host: (F, frame_size, 2) float32 numpy array uploaded shard by shard, or None with `total` (device buffers left empty)"""
        if frames_host is not None:
            x = np.ascontiguousarray(frames_host, np.float32)
            total = x.shape[0]
            self._check(self.L.qpsk_multi_load(self.h, total, x.ctypes.data_as(C.c_void_p)))
        else:
            self._check(self.L.qpsk_multi_load(self.h, int(total), None))
        self.total = int(total)
        ctx = C.c_void_p()
        self._check(self.L.qpsk_multi_shard(self.h, 0, None, None, None, C.byref(ctx), None))
        self.nsym = self.L.qpsk_ctx_nsym(ctx)

    def shard(self, r):
        dev, first, count, ctx, d_in = C.c_int32(), C.c_longlong(), C.c_longlong(), C.c_void_p(), C.c_void_p()
        self._check(self.L.qpsk_multi_shard(self.h, r, C.byref(dev), C.byref(first), C.byref(count), C.byref(ctx), C.byref(d_in)))
        return dict(device=dev.value, first=first.value, count=count.value, ctx=ctx, d_in=d_in.value)

    def use_device_input(self, r, tensor):
        self._check(self.L.qpsk_multi_use_device_input(self.h, r, C.c_void_p(tensor.data_ptr())))

    def begin(self, slot):
        self._check(self.L.qpsk_multi_rx_begin(self.h, slot))

    def end(self, slot, sym=None, freq=None, phase=None):
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)      # noqa: E731
        self._check(self.L.qpsk_multi_rx_end(self.h, slot, p(sym), p(freq), p(phase)))

    def set_packed(self, on):
        self._check(self.L.qpsk_multi_set_packed(self.h, int(bool(on))))
        self.packed = bool(on)

    def unpack(self, packed):
        out = np.empty((packed.shape[0], self.nsym), np.uint8)
        self._check(self.L.qpsk_unpack_symbols_host(packed.ctypes.data_as(C.c_void_p), packed.shape[0], self.nsym, out.ctypes.data_as(C.c_void_p)))
        return out

    def pinned_outputs(self):
        """(sym, freq, phase) numpy views of page-locked memory (qpsk_host_alloc) for direct mode; kept alive by this object"""
        out = []
        cols = (self.nsym + 3) // 4 if getattr(self, "packed", False) else self.nsym
        for shape, dt in (((self.total, cols), np.uint8), ((self.total,), np.float32), ((self.total,), np.float32)):
            nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
            p = C.c_void_p()
            self._check(self.L.qpsk_host_alloc(C.byref(p), nbytes))
            self._pinned = getattr(self, "_pinned", []) + [p]
            buf = (C.c_uint8 * nbytes).from_address(p.value)
            out.append(np.frombuffer(buf, dtype=dt).reshape(shape))
        return tuple(out)

    def set_direct(self, slot, sym=None, freq=None, phase=None):
        p = lambda a: None if a is None else C.c_void_p(a.ctypes.data)      # noqa: E731
        self._check(self.L.qpsk_multi_set_direct_output(self.h, slot, p(sym), p(freq), p(phase)))

    def outputs(self):
        cols = (self.nsym + 3) // 4 if getattr(self, "packed", False) else self.nsym
        return (np.empty((self.total, cols), np.uint8), np.empty(self.total, np.float32), np.empty(self.total, np.float32))

    def close(self):
        if getattr(self, "h", None):
            self.L.qpsk_multi_destroy(self.h)
            self.h = None
            for p in getattr(self, "_pinned", []):
                self.L.qpsk_host_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

