"""ctypes binding of the C ABI (include/amps_recc.h) -- the only way Python reaches the HIP path.

There is deliberately no pure-Python / torch / numpy implementation of any entry point here: if
libamps_recc.so is missing or no HIP device is usable, calls raise (the library itself returns
-ENODEV from amps_recc_create).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMPS_RECC_LIB") or os.path.join(_HERE, "libamps_recc.so")  # env override: A/B builds only

CAPTURE = 3374
MAX_WORK_ITEMS = 61439
MEM_HOST, MEM_DEVICE = 0, 1
FLAG_TIME_KERNELS = 1
FLAG_MAJORITY = 2
FLAG_UNFUSED_WIDEBAND = 4
FLAG_SLICER_PRODUCT = 8
FLAG_SLICER_SINE = 16
FLAG_KEEP_BURSTS = 32
FLAG_SLICER_ATAN = 64
FLAG_SLICER_EXACT = 128
FLAG_FIXED_TIMING = 256

# slicer names accepted by Recc(slicer=...): numeric spec of include/amps_recc_numerics.h -> cfg flag
_SLICER_FLAGS = {None: 0, "default": 0, "atan": FLAG_SLICER_ATAN, 0: FLAG_SLICER_ATAN, "A": FLAG_SLICER_ATAN,
                 "product": FLAG_SLICER_PRODUCT, 1: FLAG_SLICER_PRODUCT, "B": FLAG_SLICER_PRODUCT,
                 "sine": FLAG_SLICER_SINE, 2: FLAG_SLICER_SINE, "C": FLAG_SLICER_SINE,
                 "exact": FLAG_SLICER_EXACT, 3: FLAG_SLICER_EXACT, "D": FLAG_SLICER_EXACT}
SLICER_NAMES = ("atan", "product", "sine", "exact")      # indexed by AMPS_SLICER_*

MSG_CLASSES = ("invalid_word_a", "e_zero", "page_response", "registration", "origination", "bad_nawc", "unknown")

# mirrors amps_recc_burst_t; checked against amps_recc_burst_size() at load
BURST_DTYPE = np.dtype([
    ("channel", "<u4"), ("flags", "<u4"), ("position", "<u8"),
    ("dcc", "u1", (7,)), ("dcc_bad", "u1"),
    ("manch_bad", "<u2", (7,)), ("valid", "u1", (7,)), ("first_valid_rep", "u1", (7,)),
    ("word_raw", "u1", (7, 48)), ("word_dec", "u1", (7, 36)),
    ("a_F", "u1"), ("a_NAWC", "u1"), ("a_T", "u1"), ("a_S", "u1"), ("a_E", "u1"), ("a_ER", "u1"),
    ("a_SCM", "u1"), ("_pad0", "u1"), ("a_MIN1", "<u4"),
    ("b_F", "u1"), ("b_NAWC", "u1"), ("b_MSG_TYPE", "u1"), ("b_ORDQ", "u1"), ("b_ORDER", "u1"),
    ("b_LT", "u1"), ("b_EP", "u1"), ("b_SCM4", "u1"), ("b_MPCI", "u1"), ("b_SDCC1", "u1"),
    ("b_SDCC2", "u1"), ("_pad1", "u1"), ("b_MIN2", "<u2"), ("_pad2", "<u2"),
    ("esn", "<u4"), ("has_esn", "u1"), ("msg_class", "u1"), ("n_called_words", "u1"), ("_pad3", "u1"),
    ("min", "S12"), ("dialed", "S36"), ("_pad4", "<u4"),
], align=False)


class Cfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_channels", C.c_uint32), ("samples_per_symbol", C.c_uint32),
        ("max_samples_per_push", C.c_uint32), ("max_bursts", C.c_uint32), ("device", C.c_int32),
        ("flags", C.c_uint32), ("wideband_channels", C.c_uint32), ("wideband_decim", C.c_uint32),
        ("wideband_taps_per_branch", C.c_uint32), ("wideband_first_channel", C.c_uint32),
        ("sync_tolerance", C.c_uint32), ("wideband_groups", C.c_uint32), ("wideband_group", C.c_uint32), ("stream", C.c_void_p),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("launches_front", C.c_uint32), ("ms_front", C.c_double),
        ("ms_resolve", C.c_double), ("ms_decode", C.c_double), ("ms_carry", C.c_double),
        ("ms_symbols", C.c_double), ("samples_front", C.c_uint64),
        ("ms_channelizer", C.c_double), ("launches_channelizer", C.c_uint32), ("_pad", C.c_uint32),
        ("ms_xlate", C.c_double),
    ]


class XlateCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("decim", C.c_uint32), ("rate_hz", C.c_double), ("center_hz", C.c_double),
        ("gain", C.c_double), ("cutoff_hz", C.c_double), ("width_hz", C.c_double),
    ]


class RcclInfo(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("alive", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32),
        ("comm_nranks", C.c_int32), ("comm_rank", C.c_int32), ("device", C.c_int32),
        ("pci_domain", C.c_int32), ("pci_bus", C.c_int32), ("pci_device", C.c_int32),
        ("device_uuid", C.c_uint8 * 16), ("max_samples_per_push", C.c_uint64), ("max_bursts_per_gather", C.c_uint32),
        ("timeout_ms", C.c_uint32), ("collectives_timed", C.c_uint64), ("collective_ms", C.c_double),
        ("collective_bytes", C.c_uint64), ("last_mode", C.c_int32), ("_pad", C.c_int32), ("library", C.c_char * 96),
    ]


class Reply(C.Structure):
    _fields_ = [
        ("has_focc", C.c_uint8), ("focc_stream", C.c_int32), ("focc_nwords", C.c_int32),
        ("focc_word1", C.c_uint8 * 28), ("focc_word2", C.c_uint8 * 28),
        ("has_fvc", C.c_uint8), ("fvc_count", C.c_int32), ("fvc_word1", C.c_uint8 * 28),
        ("fvc_repeat", C.c_uint64),
        ("has_mutes", C.c_uint8), ("fvc_mute", C.c_uint8), ("audio_mute", C.c_uint8),
        ("has_command", C.c_uint8), ("command", C.c_char * 48),
    ]


EXPORTS = (
    "amps_recc_abi_version", "amps_recc_strerror", "amps_recc_burst_size", "amps_recc_create",
    "amps_recc_destroy", "amps_recc_reset", "amps_recc_push_symbols", "amps_recc_decode_bursts",
    "amps_recc_push_iq", "amps_recc_push_wideband", "amps_recc_drain", "amps_recc_debug_demod",
    "amps_recc_get_timing", "amps_recc_reply_words", "amps_recc_debug_channelize",
    "amps_bch_encode_words", "amps_bch_decode_words",
    "amps_recc_set_xlate", "amps_recc_push_raw", "amps_recc_debug_xlate", "amps_recc_set_timing",
    "amps_recc_drain_begin", "amps_recc_drain_end", "amps_recc_set_origin",
    "amps_recc_wait_event", "amps_recc_record_event", "amps_recc_refchain_symbols", "amps_recc_refchain_tables",
    "amps_recc_drain_bursts", "amps_recc_default_slicer", "amps_recc_default_wideband_decim", "amps_recc_debug_exact_slice",
    "amps_recc_rccl_unique_id", "amps_recc_rccl_init", "amps_recc_push_wideband_bcast", "amps_recc_drain_gather",
    "amps_recc_push_wideband_dist", "amps_recc_rccl_info", "amps_recc_rccl_abort", "amps_recc_rccl_set_timeout",
)
_NEW_IN_ABI4 = ("amps_recc_push_wideband_dist", "amps_recc_rccl_info", "amps_recc_rccl_abort", "amps_recc_rccl_set_timeout")
DIST_BROADCAST, DIST_SCATTER_ALLGATHER = 0, 1
DIST_MODES = {"broadcast": DIST_BROADCAST, "scatter_allgather": DIST_SCATTER_ALLGATHER, 0: 0, 1: 1}

_lib = None


class AmpsError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = load().amps_recc_strerror(code).decode() if _lib is not None else "library not loaded"
        super().__init__(f"{where}: {msg} ({code})")


def load():
    """dlopen the C-ABI library.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
                          "there is no CPU fallback for the RECC path")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.amps_recc_abi_version.restype = C.c_int
    if hasattr(L, "amps_recc_default_slicer"):      # absent only from A/B builds of earlier revisions (AMPS_RECC_LIB)
        L.amps_recc_default_slicer.restype = C.c_int
    if hasattr(L, "amps_recc_default_wideband_decim"):
        L.amps_recc_default_wideband_decim.restype = C.c_uint32
    L.amps_recc_strerror.argtypes = [C.c_int]
    L.amps_recc_strerror.restype = C.c_char_p
    L.amps_recc_burst_size.restype = C.c_size_t
    L.amps_recc_create.argtypes = [C.POINTER(vp), C.POINTER(Cfg)]
    L.amps_recc_destroy.argtypes = [vp]
    L.amps_recc_destroy.restype = None
    L.amps_recc_reset.argtypes = [vp]
    L.amps_recc_set_origin.argtypes = [vp, C.c_uint64]
    L.amps_recc_push_symbols.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amps_recc_decode_bursts.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
    L.amps_recc_push_iq.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int]
    L.amps_recc_push_wideband.argtypes = [vp, vp, C.c_size_t, C.c_int]
    if hasattr(L, "amps_recc_push_wideband_bcast"):   # absent only from A/B builds of earlier revisions (AMPS_RECC_LIB)
        L.amps_recc_rccl_unique_id.argtypes = [vp]
        L.amps_recc_rccl_init.argtypes = [vp, vp, C.c_int, C.c_int]
        L.amps_recc_push_wideband_bcast.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
    if hasattr(L, "amps_recc_drain_gather"):
        L.amps_recc_drain_gather.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_int]
    if hasattr(L, "amps_recc_push_wideband_dist"):
        L.amps_recc_push_wideband_dist.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        L.amps_recc_rccl_info.argtypes = [vp, C.POINTER(RcclInfo)]
        L.amps_recc_rccl_abort.argtypes = [vp]
        L.amps_recc_rccl_set_timeout.argtypes = [vp, C.c_uint32]
    L.amps_recc_drain.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amps_recc_refchain_symbols.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp, C.c_size_t, vp]
    L.amps_recc_refchain_tables.argtypes = [vp, vp, vp]
    L.amps_recc_wait_event.argtypes = [vp, vp]
    L.amps_recc_record_event.argtypes = [vp, vp]
    L.amps_recc_drain_bursts.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amps_recc_drain_begin.argtypes = [vp]
    L.amps_recc_drain_end.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amps_recc_debug_demod.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp, vp]
    L.amps_recc_get_timing.argtypes = [vp, C.POINTER(Timing), C.c_int]
    L.amps_recc_set_timing.argtypes = [vp, C.c_int]
    L.amps_recc_debug_channelize.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amps_recc_reply_words.argtypes = [vp, C.POINTER(Reply)]
    L.amps_recc_set_xlate.argtypes = [vp, C.POINTER(XlateCfg)]
    L.amps_recc_push_raw.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int]
    L.amps_recc_debug_xlate.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.amps_bch_encode_words.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp]
    L.amps_bch_decode_words.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp, vp]
    for name in EXPORTS:
        if name in ("amps_recc_default_slicer", "amps_recc_default_wideband_decim", "amps_recc_debug_exact_slice", "amps_recc_rccl_unique_id", "amps_recc_rccl_init",
                    "amps_recc_push_wideband_bcast", "amps_recc_drain_gather") + _NEW_IN_ABI4 and not hasattr(L, name):
            continue
        if name not in ("amps_recc_strerror", "amps_recc_burst_size", "amps_recc_destroy"):   # every other entry point returns int
            getattr(L, name).restype = C.c_int
    if L.amps_recc_burst_size() != BURST_DTYPE.itemsize:
        raise ImportError("amps_recc_burst_t layout mismatch between binding and library")
    _lib = L
    return L


def _hostptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _as_ptr(x, sync_torch=True):
    """numpy array -> (pointer, MEM_HOST, keepalive); torch CUDA tensor -> (pointer, MEM_DEVICE, keepalive).
    The library launches on its own non-blocking stream, which is not ordered against torch's: a device tensor is pushed
    only after the torch stream that produced it has drained (a pageable H2D copy or a generator kernel may still be in
    flight when .to() / randn() return).  Callers that order the streams themselves (Recc.wait_torch) pass sync_torch=False."""
    if isinstance(x, np.ndarray):
        return _hostptr(x), MEM_HOST, x
    if hasattr(x, "data_ptr"):
        if x.is_cuda:
            if sync_torch:
                import torch
                torch.cuda.current_stream(x.device).synchronize()
            return C.c_void_p(x.data_ptr()), MEM_DEVICE, x
        a = x.numpy()
        return _hostptr(a), MEM_HOST, a
    raise TypeError(type(x))


class Recc:
    """One handle = `n_channels` independent RECC receivers on one MI355X."""

    def __init__(self, n_channels=1, sps=None, max_samples=0, max_bursts=1024, device=-1, time_kernels=False,
                 stream=None, wideband=None, majority=False, unfused_wideband=False, sync_tolerance=0, slicer="default",
                 sync_torch=True, keep_bursts=False, fixed_timing=False):
        L = load()
        if isinstance(slicer, bool) or slicer not in _SLICER_FLAGS:        # a typo must not run a different numeric spec silently
            raise ValueError("slicer must be one of %r" % sorted(map(str, _SLICER_FLAGS)))
        self.slicer = SLICER_NAMES[L.amps_recc_default_slicer() if hasattr(L, "amps_recc_default_slicer") else 0] if _SLICER_FLAGS[slicer] == 0 else \
            {FLAG_SLICER_ATAN: "atan", FLAG_SLICER_PRODUCT: "product", FLAG_SLICER_SINE: "sine", FLAG_SLICER_EXACT: "exact"}[_SLICER_FLAGS[slicer]]
        self.sync_torch = sync_torch
        if wideband:
            # "decim" absent / 0 = the library default; the samples per symbol follow the decimation unless they are given
            wideband = dict(wideband)
            if not wideband.get("decim"):
                wideband["decim"] = int(L.amps_recc_default_wideband_decim()) if hasattr(L, "amps_recc_default_wideband_decim") else 512
            if sps is None:
                sps = 1536 // int(wideband["decim"])
        elif sps is None:
            sps = 10
        cfg = Cfg()
        cfg.struct_size = C.sizeof(Cfg)
        cfg.n_channels = n_channels
        cfg.samples_per_symbol = sps
        cfg.max_samples_per_push = max_samples
        cfg.max_bursts = max_bursts
        cfg.device = device
        cfg.flags = ((FLAG_TIME_KERNELS if time_kernels else 0) | (FLAG_MAJORITY if majority else 0)
                     | (FLAG_UNFUSED_WIDEBAND if unfused_wideband else 0)
                     | _SLICER_FLAGS[slicer]
                     | (FLAG_KEEP_BURSTS if keep_bursts else 0) | (FLAG_FIXED_TIMING if fixed_timing else 0))
        cfg.stream = stream
        cfg.sync_tolerance = sync_tolerance
        if wideband:
            cfg.wideband_channels = wideband["channels"]
            cfg.wideband_decim = wideband["decim"]
            cfg.wideband_taps_per_branch = wideband.get("taps_per_branch", 8)
            cfg.wideband_first_channel = wideband.get("first_channel", 0)
            cfg.wideband_groups = wideband.get("groups", 0)
            cfg.wideband_group = wideband.get("group", 0)
        self.n_channels, self.sps, self.max_bursts, self.max_samples = n_channels, sps, max_bursts, max_samples
        self.decim = int(wideband["decim"]) if wideband else None
        self._h = C.c_void_p()
        rc = L.amps_recc_create(C.byref(self._h), C.byref(cfg))
        if rc != 0:
            self._h = None
            raise AmpsError(rc, "amps_recc_create")

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.amps_recc_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        rc = load().amps_recc_reset(self._h)
        if rc:
            raise AmpsError(rc, "amps_recc_reset")

    # ---- seam (i): recc::work ----
    def push_symbols(self, syms, n=None):
        """syms: uint8 [C][ld] (numpy or torch cuda). One work() call per channel with noutput_items=n.
        Returns (bursts uint8 [k][3374], channels uint32 [k])."""
        if isinstance(syms, np.ndarray):
            syms = np.ascontiguousarray(syms, np.uint8)
        syms2 = syms.reshape(self.n_channels, -1)
        ld = syms2.shape[1]
        n = ld if n is None else n
        ptr, mem, keep = _as_ptr(syms2, self.sync_torch)
        out = np.zeros((self.max_bursts, CAPTURE), np.uint8)
        ch = np.zeros(self.max_bursts, np.uint32)
        nout = C.c_size_t(0)
        rc = load().amps_recc_push_symbols(self._h, ptr, ld, n, mem, _hostptr(out), _hostptr(ch), self.max_bursts, C.byref(nout))
        if rc:
            raise AmpsError(rc, "amps_recc_push_symbols")
        return out[:nout.value].copy(), ch[:nout.value].copy()

    def decode_bursts(self, bursts, channels=None):
        if isinstance(bursts, np.ndarray):
            bursts = np.ascontiguousarray(bursts, np.uint8).reshape(-1, CAPTURE)
        nb = bursts.shape[0]
        out = np.zeros(nb, BURST_DTYPE)
        if nb == 0:
            return out
        ptr, mem, keep = _as_ptr(bursts, self.sync_torch)
        chp = None
        if channels is not None:
            channels = np.ascontiguousarray(channels, np.uint32)
            chp = _hostptr(channels)
        rc = load().amps_recc_decode_bursts(self._h, ptr, nb, mem, chp, _hostptr(out))
        if rc:
            raise AmpsError(rc, "amps_recc_decode_bursts")
        return out

    # ---- seam (ii): fused IQ ----
    def push_iq(self, iq, nsamp=None):
        """iq: complex64 [C][ld] numpy array, or torch cuda tensor (complex64 [C][ld] or float32 [C][ld][2])."""
        if isinstance(iq, np.ndarray):
            iq = np.ascontiguousarray(iq, np.complex64).reshape(self.n_channels, -1)
            ld = iq.shape[1]
        else:
            ld = iq.shape[1]
        nsamp = ld if nsamp is None else nsamp
        ptr, mem, keep = _as_ptr(iq, self.sync_torch)
        rc = load().amps_recc_push_iq(self._h, ptr, ld, nsamp, mem)
        if rc:
            raise AmpsError(rc, "amps_recc_push_iq")

    def set_xlate(self, rate_hz=400e3, center_hz=160e3, decim=2, gain=0.0, cutoff_hz=0.0, width_hz=0.0):
        """Put the reference flow graph's channel filter (freq_xlating_fir_filter_ccc, grc/recctest.grc:889-937)
        in front of the IQ seam; zeros select the flow graph's gain / cutoff / transition width."""
        x = XlateCfg(C.sizeof(XlateCfg), decim, rate_hz, center_hz, gain, cutoff_hz, width_hz)
        rc = load().amps_recc_set_xlate(self._h, C.byref(x))
        if rc:
            raise AmpsError(rc, "amps_recc_set_xlate")

    def push_raw(self, iq, nsamp=None):
        """like push_iq, at the translate stage's input rate (e.g. the 400 ksps .raw captures of recctest.grc)."""
        if isinstance(iq, np.ndarray):
            iq = np.ascontiguousarray(iq, np.complex64).reshape(self.n_channels, -1)
        ld = iq.shape[1]
        nsamp = ld if nsamp is None else nsamp
        ptr, mem, keep = _as_ptr(iq, self.sync_torch)
        rc = load().amps_recc_push_raw(self._h, ptr, ld, nsamp, mem)
        if rc:
            raise AmpsError(rc, "amps_recc_push_raw")

    def debug_xlate(self, iq):
        """Translate stage only (test tap): complex64 [C][n] -> complex64 [C][nout]."""
        if isinstance(iq, np.ndarray):
            iq = np.ascontiguousarray(iq, np.complex64).reshape(self.n_channels, -1)
        n = iq.shape[1]
        ptr, mem, keep = _as_ptr(iq)
        cap = n + 8
        out = np.zeros((self.n_channels, cap), np.complex64)
        no = C.c_size_t(0)
        rc = load().amps_recc_debug_xlate(self._h, ptr, n, n, mem, _hostptr(out), cap, C.byref(no))
        if rc:
            raise AmpsError(rc, "amps_recc_debug_xlate")
        return out[:, :no.value].copy()

    def push_wideband(self, iq):
        if isinstance(iq, np.ndarray):
            iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
        n = iq.shape[0]
        ptr, mem, keep = _as_ptr(iq, self.sync_torch)
        rc = load().amps_recc_push_wideband(self._h, ptr, n, mem)
        if rc:
            raise AmpsError(rc, "amps_recc_push_wideband")

    # ---- one band over the GPUs of a node: RCCL inside the C ABI (include/amps_recc.h, amps_recc_push_wideband_bcast)
    @staticmethod
    def rccl_unique_id():
        """128 bytes from ncclGetUniqueId: one rank makes them, the application carries them to the others"""
        buf = (C.c_uint8 * 128)()
        rc = load().amps_recc_rccl_unique_id(buf)
        if rc:
            raise AmpsError(rc, "amps_recc_rccl_unique_id")
        return bytes(buf)

    def rccl_init(self, uid, nranks, rank):
        """collective: returns when all nranks handles (one per GPU / process) have joined"""
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        rc = load().amps_recc_rccl_init(self._h, buf, int(nranks), int(rank))
        if rc:
            raise AmpsError(rc, "amps_recc_rccl_init")

    def push_wideband_dist(self, iq, nsamp=None, root=0, mode="broadcast"):
        """every rank in step; the root passes its block (a CUDA tensor, used in place, or a numpy array, staged), the others None.
        mode: "broadcast" (flat ncclBroadcast) or "scatter_allgather".  Returns the number of samples pushed: the ROOT's, on every rank.
        The root passing None ends the stream: every rank's call raises AmpsError with code -errno.ENODATA, nothing is pushed."""
        ptr, mem, n = None, MEM_DEVICE, 0
        if iq is not None:
            if isinstance(iq, np.ndarray):
                iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
            have = int(iq.numel()) if hasattr(iq, "numel") else int(iq.size)      # complex samples
            n = have if nsamp is None else int(nsamp)
            if n > have:                                   # the C side would read past the end of the buffer
                raise ValueError("nsamp = %d exceeds the %d complex samples of the block" % (n, have))
            ptr, mem, keep = _as_ptr(iq, self.sync_torch)
        pushed = C.c_size_t(0)
        rc = load().amps_recc_push_wideband_dist(self._h, ptr, n, mem, int(root), DIST_MODES[mode], C.byref(pushed))
        if rc:
            raise AmpsError(rc, "amps_recc_push_wideband_dist")
        return int(pushed.value)

    def push_wideband_bcast(self, iq, nsamp=None, root=0):
        """push_wideband_dist(mode="broadcast"): the round-4 entry point"""
        return self.push_wideband_dist(iq, nsamp, root, "broadcast")

    def rccl_abort(self):
        """this rank leaves: its communicator is aborted, the peers' bounded waits end with -ETIMEDOUT"""
        rc = load().amps_recc_rccl_abort(self._h)
        if rc:
            raise AmpsError(rc, "amps_recc_rccl_abort")

    def rccl_set_timeout(self, milliseconds):
        rc = load().amps_recc_rccl_set_timeout(self._h, int(milliseconds))
        if rc:
            raise AmpsError(rc, "amps_recc_rccl_set_timeout")

    def rccl_info(self):
        """the communicator as RCCL reports it + the device it runs on + the timed data collectives, as a dict"""
        info = RcclInfo()
        info.struct_size = C.sizeof(RcclInfo)
        rc = load().amps_recc_rccl_info(self._h, C.byref(info))
        if rc:
            raise AmpsError(rc, "amps_recc_rccl_info")
        d = {k: getattr(info, k) for k, _ in RcclInfo._fields_ if k not in ("struct_size", "_pad", "device_uuid", "library")}
        d["device_uuid"] = bytes(info.device_uuid).hex()
        d["library"] = info.library.decode(errors="replace")
        d["last_mode"] = {0: "broadcast", 1: "scatter_allgather"}.get(info.last_mode)
        d["collective_gbps"] = (info.collective_bytes / (info.collective_ms * 1e-3) / 1e9) if info.collective_ms > 0 else None
        return d

    def drain_gather(self, root=0, cap=None):
        """every rank in step: each drains its own list, the root returns the records of ALL ranks sorted by (channel, position)
        (what one whole-band handle would have drained), the others an empty array"""
        cap = cap or self.max_bursts
        out = np.empty(cap, BURST_DTYPE)
        nout = C.c_size_t(0)
        rc = load().amps_recc_drain_gather(self._h, _hostptr(out), cap, C.byref(nout), int(root))
        if rc:
            raise AmpsError(rc, "amps_recc_drain_gather")
        return out[:nout.value].copy()

    def refchain_symbols(self, iq):
        """The flow graph's own sub-chain (quadrature_demod_cf -> clock_recovery_mm_ff -> binary_slicer_fb) on the device:
        complex64 [C][n] at 200 ksps -> list of uint8 symbol arrays, one per channel (continues across calls)."""
        if isinstance(iq, np.ndarray):
            iq = np.ascontiguousarray(iq, np.complex64).reshape(self.n_channels, -1)
        n = iq.shape[1]
        ptr, mem, keep = _as_ptr(iq, self.sync_torch)
        cap = n // 9 + 16
        out = np.zeros((self.n_channels, cap), np.uint8)
        ns = np.zeros(self.n_channels, np.uint32)
        rc = load().amps_recc_refchain_symbols(self._h, ptr, n, n, mem, _hostptr(out), cap, _hostptr(ns))
        if rc:
            raise AmpsError(rc, "amps_recc_refchain_symbols")
        return [out[c, :ns[c]].copy() for c in range(self.n_channels)]

    def refchain_tables(self):
        a, m = np.zeros(258, np.float32), np.zeros((129, 8), np.float32)
        rc = load().amps_recc_refchain_tables(self._h, _hostptr(a), _hostptr(m))
        if rc:
            raise AmpsError(rc, "amps_recc_refchain_tables")
        return a, m

    def wait_torch(self, stream=None):
        """Order later pushes behind everything enqueued so far on a torch CUDA stream (default: the current one), without
        blocking the host: an event is recorded on that stream and the handle's stream waits for it."""
        import torch
        ev = torch.cuda.Event()
        ev.record(stream or torch.cuda.current_stream())
        rc = load().amps_recc_wait_event(self._h, C.c_void_p(ev.cuda_event))
        if rc:
            raise AmpsError(rc, "amps_recc_wait_event")
        self._keep_ev = ev

    def record_torch_event(self):
        """The converse: a torch event recorded behind everything this handle has enqueued so far; a torch stream that
        waits for it (stream.wait_event) may then overwrite a buffer the handle's kernels read."""
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())          # creates the underlying hipEvent_t
        rc = load().amps_recc_record_event(self._h, C.c_void_p(ev.cuda_event))
        if rc:
            raise AmpsError(rc, "amps_recc_record_event")
        return ev

    def bch_encode(self, msg_bits):
        """uint8 [n][k] message bits -> uint8 [n][k+12] code words (k = 28: FOCC/FVC, k = 36: RECC)."""
        m = np.ascontiguousarray(msg_bits, np.uint8)
        n, k = m.shape
        out = np.zeros((n, k + 12), np.uint8)
        rc = load().amps_bch_encode_words(self._h, _hostptr(m), n, k, MEM_HOST, _hostptr(out))
        if rc:
            raise AmpsError(rc, "amps_bch_encode_words")
        return out

    def bch_decode(self, codewords):
        """uint8 [n][k+12] -> (msg uint8 [n][k], valid uint8 [n], nerrors uint8 [n])."""
        c = np.ascontiguousarray(codewords, np.uint8)
        n, nb = c.shape
        k = nb - 12
        msg, valid, nerr = np.zeros((n, k), np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        rc = load().amps_bch_decode_words(self._h, _hostptr(c), n, k, MEM_HOST, _hostptr(msg), _hostptr(valid), _hostptr(nerr))
        if rc:
            raise AmpsError(rc, "amps_bch_decode_words")
        return msg, valid, nerr

    def debug_channelize(self, iq):
        """Channelizer only (test tap): wideband complex64 [n] -> complex64 [C][nframes]."""
        if isinstance(iq, np.ndarray):
            iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
        n = iq.shape[0]
        ptr, mem, keep = _as_ptr(iq)
        cap = n // 512 + 2
        out = np.zeros((self.n_channels, cap), np.complex64)
        nf = C.c_size_t(0)
        rc = load().amps_recc_debug_channelize(self._h, ptr, n, mem, _hostptr(out), cap, C.byref(nf))
        if rc:
            raise AmpsError(rc, "amps_recc_debug_channelize")
        return out[:, :nf.value].copy()

    def drain(self, cap=None, copy=True):
        """Synchronise and return the decoded bursts since the last drain, sorted by (channel, position).
        copy=False returns a view of a buffer owned by this handle, valid until the next drain()."""
        cap = cap or self.max_bursts
        out = getattr(self, "_drain_buf", None)
        if out is None or out.shape[0] < cap:
            out = self._drain_buf = np.empty(cap, BURST_DTYPE)
        nout = C.c_size_t(0)
        rc = load().amps_recc_drain(self._h, _hostptr(out), cap, C.byref(nout))
        if rc:
            raise AmpsError(rc, "amps_recc_drain")
        return out[:nout.value].copy() if copy else out[:nout.value]

    def drain_bursts(self, cap=None):
        """drain() plus the 3374 captured symbol bytes of every record (handle created with keep_bursts=True)"""
        cap = cap or self.max_bursts
        out = np.empty(cap, BURST_DTYPE)
        sym = np.zeros((cap, CAPTURE), np.uint8)
        nout = C.c_size_t(0)
        rc = load().amps_recc_drain_bursts(self._h, _hostptr(out), _hostptr(sym), cap, C.byref(nout))
        if rc:
            raise AmpsError(rc, "amps_recc_drain_bursts")
        return out[:nout.value].copy(), sym[:nout.value].copy()

    def set_origin(self, first_sample):
        """absolute index of the first sample pushed after create / reset (multiple of 64)"""
        rc = load().amps_recc_set_origin(self._h, first_sample)
        if rc:
            raise AmpsError(rc, "amps_recc_set_origin")

    def drain_begin(self):
        """Close the current record list without waiting; pushes issued after this append to a second list."""
        rc = load().amps_recc_drain_begin(self._h)
        if rc:
            raise AmpsError(rc, "amps_recc_drain_begin")

    def drain_end(self, cap=None, copy=True):
        """Wait for the work enqueued before drain_begin() only, and return its records (sorted)."""
        cap = cap or self.max_bursts
        out = getattr(self, "_drain_buf", None)
        if out is None or out.shape[0] < cap:
            out = self._drain_buf = np.empty(cap, BURST_DTYPE)
        nout = C.c_size_t(0)
        rc = load().amps_recc_drain_end(self._h, _hostptr(out), cap, C.byref(nout))
        if rc:
            raise AmpsError(rc, "amps_recc_drain_end")
        return out[:nout.value].copy() if copy else out[:nout.value]

    def debug_demod(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
        n = iq.size
        d, s, g = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8)
        rc = load().amps_recc_debug_demod(self._h, _hostptr(iq), n, MEM_HOST, _hostptr(d), _hostptr(s), _hostptr(g))
        if rc:
            raise AmpsError(rc, "amps_recc_debug_demod")
        p = (n // 64) * 64
        return d[:p], s[:p], g[:p]

    def set_timing(self, mode):
        """mode: "off" | "all" | "dominant" (only the streaming kernel of the seam in use is bracketed by HIP events) |
        "sampled" (the dominant kernel of every eighth push)"""
        rc = load().amps_recc_set_timing(self._h, {"off": 0, "all": 1, "dominant": 2, "sampled": 3}[mode])
        if rc:
            raise AmpsError(rc, "amps_recc_set_timing")

    def timing(self, reset=False):
        t = Timing()
        rc = load().amps_recc_get_timing(self._h, C.byref(t), int(reset))
        if rc:
            raise AmpsError(rc, "amps_recc_get_timing")
        return {k: getattr(t, k) for k, _ in Timing._fields_ if k not in ("struct_size", "_pad")}


def reply_words(rec):
    rec = np.ascontiguousarray(rec)
    r = Reply()
    rc = load().amps_recc_reply_words(_hostptr(rec), C.byref(r))
    if rc:
        raise AmpsError(rc, "amps_recc_reply_words")
    return r
