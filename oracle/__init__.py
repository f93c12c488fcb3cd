"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package
(see oracle/amps_oracle.h for the parity status).  Nothing under gr_amps_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libamps_oracle.so")

CAPTURE = 3374
TRIGGER = 74

# mirrors amps_recc_burst_t (include/amps_recc.h); itemsize checked against the C side at load
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
assert BURST_DTYPE.itemsize == 728, BURST_DTYPE.itemsize


class Reply(C.Structure):
    _fields_ = [
        ("has_focc", C.c_uint8), ("focc_stream", C.c_int32), ("focc_nwords", C.c_int32),
        ("focc_word1", C.c_uint8 * 28), ("focc_word2", C.c_uint8 * 28),
        ("has_fvc", C.c_uint8), ("fvc_count", C.c_int32), ("fvc_word1", C.c_uint8 * 28),
        ("fvc_repeat", C.c_uint64),
        ("has_mutes", C.c_uint8), ("fvc_mute", C.c_uint8), ("audio_mute", C.c_uint8),
        ("has_command", C.c_uint8), ("command", C.c_char * 48),
    ]


def build(force=False):
    """Compile the C restatement (gcc). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("ref_chain.c", "fused_model.c", "amps_oracle.h")]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("amps_recc.h", "amps_recc_numerics.h")]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(s) <= os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s)):
            return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libamps_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    u8p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    L.orc_recc_new.restype = C.c_void_p
    L.orc_recc_free.argtypes = [C.c_void_p]
    L.orc_recc_reset.argtypes = [C.c_void_p]
    L.orc_recc_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_recc_work.restype = C.c_int
    L.orc_recc_peek.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p)]
    L.orc_manchester_decode_binbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
    L.orc_manchester_decode_binbuf.restype = C.c_size_t
    L.orc_manchester_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    L.orc_manchester_encode.restype = C.c_int
    L.orc_trigger.argtypes = [C.c_void_p]
    L.orc_bch63_decode.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.orc_bch63_decode.restype = C.c_int
    L.orc_bch63_encode.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_bch_generator.restype = C.c_uint32
    L.orc_recc_bch_decode.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_recc_bch_decode.restype = C.c_int
    L.orc_bch_encode_short.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_decode_burst.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.orc_decode_burst_mode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
    L.orc_reply_words.argtypes = [C.c_void_p, C.POINTER(Reply)]
    L.orc_parse_min.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_parse_min.restype = C.c_int
    L.orc_calc_min.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p]
    L.orc_called_digits.argtypes = [C.c_uint32, C.c_char_p, C.POINTER(C.c_int)]
    L.orc_focc_word1.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint64]
    L.orc_focc_word2_general.argtypes = [C.c_void_p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint]
    L.orc_fvc_word1_general.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    L.orc_focc_word2_voice_channel.argtypes = [C.c_void_p, C.c_uint, C.c_uint64, C.c_uint, C.c_uint]
    L.orc_firdes_low_pass_blackman.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_int]
    L.orc_firdes_low_pass_blackman.restype = C.c_int
    L.orc_freq_xlating_fir.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p]
    L.orc_freq_xlating_fir.restype = C.c_size_t
    L.orc_fast_atan2f.argtypes = [C.c_float, C.c_float]
    L.orc_fast_atan2f.restype = C.c_float
    L.orc_quadrature_demod.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_void_p]
    L.orc_mmse_taps.restype = f32p
    L.orc_chain_iq200.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orc_chain_iq200.restype = C.c_size_t
    L.orc_chain_iq400.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]
    L.orc_chain_iq400.restype = C.c_size_t
    L.orc_fused_new.argtypes = [C.c_uint32, C.c_int]
    L.orc_fused_new.restype = C.c_void_p
    L.orc_fused_free.argtypes = [C.c_void_p]
    L.orc_fused_set_tolerance.argtypes = [C.c_void_p, C.c_int]
    L.orc_fused_set_tolerance.restype = None
    L.orc_fused_set_majority.argtypes = [C.c_void_p, C.c_int]
    L.orc_fused_set_majority.restype = None
    L.orc_fused_set_slicer.argtypes = [C.c_void_p, C.c_int]
    L.orc_fused_set_slicer.restype = None
    L.orc_fused_set_tracking.argtypes = [C.c_void_p, C.c_int]
    L.orc_fused_set_tracking.restype = None
    L.orc_fused_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.orc_fused_push.restype = C.c_size_t
    L.orc_fused_processed.argtypes = [C.c_void_p]
    L.orc_fused_processed.restype = C.c_size_t
    for nm, rt in (("orc_fused_demod", f32p), ("orc_fused_soft", f32p), ("orc_fused_hard", u8p)):
        getattr(L, nm).argtypes = [C.c_void_p]
        getattr(L, nm).restype = rt
    L.orc_fm_discriminator.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _iq(a):
    """complex64 or float32 [...,2] -> contiguous float32 interleaved"""
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.ascontiguousarray(a, dtype=np.complex64).view(np.float32)
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


# ---------------------------------------------------------------- R1..R8
def trigger():
    t = np.zeros(TRIGGER, np.uint8)
    lib().orc_trigger(_ptr(t))
    return t


def manchester_encode(bits: str):
    out = np.zeros(2 * len(bits), np.uint8)
    rc = lib().orc_manchester_encode(bits.encode(), len(bits), _ptr(out))
    if rc != 0:
        raise ValueError("bit string must contain only 0/1")
    return out


def manchester_decode(src, nbits):
    src = _u8(src)
    assert src.size >= 2 * nbits
    dst = np.zeros(nbits, np.uint8)
    nb = C.c_int(0)
    bad = lib().orc_manchester_decode_binbuf(_ptr(src), _ptr(dst), nbits, C.byref(nb))
    return dst, int(bad), bool(nb.value)


class Recc:
    """R2: one reference recc block (lib/recc_impl.cc:93-145)."""

    def __init__(self):
        self._h = lib().orc_recc_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_recc_free(self._h)
            self._h = None

    def reset(self):
        lib().orc_recc_reset(self._h)

    def work(self, syms):
        syms = _u8(syms)
        out = np.zeros(CAPTURE, np.uint8)
        rc = lib().orc_recc_work(self._h, _ptr(syms), int(syms.size), _ptr(out))
        if rc < 0:
            raise ValueError("noutput_items beyond the reference's assert")
        return out if rc == 1 else None

    def state(self):
        ln, cs = C.c_uint64(), C.c_int64()
        lib().orc_recc_peek(self._h, C.byref(ln), C.byref(cs), None)
        return int(ln.value), int(cs.value)

    def run(self, stream, schedule):
        """Feed `stream` in chunks (an int or an iterable of ints, cycled); returns [(call_index, burst)]."""
        stream = _u8(stream)
        if isinstance(schedule, int):
            schedule = [schedule]
        out, off, call, k = [], 0, 0, 0
        while off < stream.size:
            n = min(int(schedule[k % len(schedule)]), stream.size - off)
            b = self.work(stream[off:off + n])
            if b is not None:
                out.append((call, b))
            off += n
            call += 1
            k += 1
        return out


def bch_generator():
    return int(lib().orc_bch_generator())


def bch_encode(msg_bits):
    msg = _u8(msg_bits)
    cw = np.zeros(msg.size + 12, np.uint8)
    lib().orc_bch_encode_short(_ptr(msg), int(msg.size), _ptr(cw))
    return cw


def bch63_decode(rx63):
    rx = _u8(rx63)
    assert rx.size == 63
    out = np.zeros(63, np.uint8)
    nf = C.c_int(0)
    ok = lib().orc_bch63_decode(_ptr(rx), _ptr(out), C.byref(nf))
    return bool(ok), out, int(nf.value)


def recc_bch_decode(bits48):
    src = _u8(bits48)
    assert src.size == 48
    dst = np.zeros(36, np.uint8)
    ok = lib().orc_recc_bch_decode(_ptr(src), _ptr(dst))
    return bool(ok), dst


def decode_bursts(bursts, channels=None, positions=None, majority=False):
    bursts = _u8(bursts).reshape(-1, CAPTURE)
    out = np.zeros(bursts.shape[0], BURST_DTYPE)
    for i in range(bursts.shape[0]):
        ch = 0 if channels is None else int(channels[i])
        pos = 0 if positions is None else int(positions[i])
        lib().orc_decode_burst_mode(_ptr(bursts[i]), ch, pos, C.c_void_p(out.ctypes.data + i * BURST_DTYPE.itemsize), int(majority))
    return out


def reply_words(rec):
    rec = np.ascontiguousarray(rec)
    r = Reply()
    lib().orc_reply_words(_ptr(rec), C.byref(r))
    return r


def parse_min(s):
    a, b = C.c_uint64(), C.c_uint64()
    ok = lib().orc_parse_min(s.encode(), C.byref(a), C.byref(b))
    return (int(a.value), int(b.value)) if ok else None


def calc_min(min1, min2):
    buf = C.create_string_buffer(11)
    lib().orc_calc_min(min1, min2, buf)
    return buf.value.decode()


def called_digits(v):
    buf = C.create_string_buffer(9)
    bad = C.c_int(0)
    lib().orc_called_digits(v, buf, C.byref(bad))
    return buf.value.decode(), bool(bad.value)


def _word28(fn, *args):
    w = np.zeros(28, np.uint8)
    fn(_ptr(w), *args)
    return "".join(str(int(b)) for b in w)


def focc_word1(multi, dcc, min1):
    return _word28(lib().orc_focc_word1, int(multi), dcc, min1)


def focc_word2_general(min2, msg_type, ordq, order):
    return _word28(lib().orc_focc_word2_general, min2, msg_type, ordq, order)


def fvc_word1_general(pscc, msg_type, ordq, order):
    return _word28(lib().orc_fvc_word1_general, pscc, msg_type, ordq, order)


def focc_word2_voice_channel(scc, min2, vmac, chan):
    return _word28(lib().orc_focc_word2_voice_channel, scc, min2, vmac, chan)


# ---------------------------------------------------------------- G1..G4
def firdes_low_pass(gain, fs, cutoff, width):
    taps = np.zeros(4096, np.float32)
    n = lib().orc_firdes_low_pass_blackman(gain, fs, cutoff, width, _ptr(taps), taps.size)
    assert n > 0
    return taps[:n].copy()


def freq_xlating_fir(iq, taps, fc, fs, decim):
    x = _iq(iq)
    n = x.size // 2
    taps = np.ascontiguousarray(taps, np.float32)
    out = np.zeros(2 * (n // decim + 1), np.float32)
    m = lib().orc_freq_xlating_fir(_ptr(x), n, _ptr(taps), taps.size, fc, fs, decim, _ptr(out))
    return out[:2 * m].view(np.complex64)


def fast_atan2f(y, x):
    return float(lib().orc_fast_atan2f(y, x))


def quadrature_demod(iq, gain=1.0):
    x = _iq(iq)
    out = np.zeros(x.size // 2, np.float32)
    lib().orc_quadrature_demod(_ptr(x), x.size // 2, gain, _ptr(out))
    return out


def mmse_taps():
    p = lib().orc_mmse_taps()
    return np.ctypeslib.as_array(p, shape=(129, 8)).copy()


def chain_iq200(iq, channel=0, chunk=4096, cap=256, want_symbols=False):
    """Reference chain from 200 ksps IQ: quad demod -> M&M -> slicer -> recc -> recc_decode."""
    x = _iq(iq)
    n = x.size // 2
    out = np.zeros(cap, BURST_DTYPE)
    syms = np.zeros(n // 9 + 16, np.uint8) if want_symbols else None
    nsym = C.c_size_t(0)
    k = lib().orc_chain_iq200(_ptr(x), n, channel, chunk, _ptr(out), cap,
                              _ptr(syms) if want_symbols else None, syms.size if want_symbols else 0, C.byref(nsym))
    if want_symbols:
        return out[:k].copy(), syms[:nsym.value].copy()
    return out[:k].copy()


def chain_iq400(iq, center_freq, channel=0, chunk=4096, cap=256):
    x = _iq(iq)
    out = np.zeros(cap, BURST_DTYPE)
    k = lib().orc_chain_iq400(_ptr(x), x.size // 2, center_freq, channel, chunk, _ptr(out), cap)
    return out[:k].copy()


# ---------------------------------------------------------------- fused model
def fm_discriminator(iq):
    x = _iq(iq)
    d = np.zeros(x.size // 2, np.float32)
    lib().orc_fm_discriminator(_ptr(x), x.size // 2, _ptr(d))
    return d


class Fused:
    """CPU model of the fused MI355X seam for one channel."""

    def __init__(self, channel=0, sps=10, tolerance=0, majority=False, slicer=None, tracking=True):
        # slicer: AMPS_SLICER_* of include/amps_recc_numerics.h (0 = A, 1 = B, 2 = C, 3 = D); None = AMPS_SLICER_DEFAULT, what a product
        # handle created with no slicer flag uses
        self._h = lib().orc_fused_new(channel, sps)
        if slicer is not None:
            lib().orc_fused_set_slicer(self._h, int(slicer))
        if not tracking:
            lib().orc_fused_set_tracking(self._h, 0)
        if tolerance:
            lib().orc_fused_set_tolerance(self._h, int(tolerance))
        if majority:
            lib().orc_fused_set_majority(self._h, 1)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_fused_free(self._h)
            self._h = None

    def push(self, iq, cap=64):
        x = _iq(iq)
        out = np.zeros(cap, BURST_DTYPE)
        k = lib().orc_fused_push(self._h, _ptr(x), x.size // 2, _ptr(out), cap)
        return out[:k].copy()

    def taps(self):
        n = lib().orc_fused_processed(self._h)
        if n == 0:
            z = np.zeros(0, np.float32)
            return z, z, np.zeros(0, np.uint8)
        d = np.ctypeslib.as_array(lib().orc_fused_demod(self._h), shape=(n,)).copy()
        s = np.ctypeslib.as_array(lib().orc_fused_soft(self._h), shape=(n,)).copy()
        g = np.ctypeslib.as_array(lib().orc_fused_hard(self._h), shape=(n,)).copy()
        return d, s, g


def fused_push_all(iq_2d, sps=10, block=None, tolerance=0, majority=False, slicer=None, tracking=True):
    """iq_2d: complex64 [C][N]; pushes every channel (optionally in blocks) and returns all records sorted."""
    iq_2d = np.asarray(iq_2d)
    recs = []
    for c in range(iq_2d.shape[0]):
        f = Fused(c, sps, tolerance, majority, slicer, tracking)
        n = iq_2d.shape[1]
        step = n if not block else block
        for off in range(0, n, step):
            blk = iq_2d[c, off:off + step]
            # an accepted burst holds the search off for 3448 symbols: the record buffer can never be too small
            recs.append(f.push(blk, cap=max(64, blk.shape[0] // (3000 * sps) + 16)))
    r = np.concatenate(recs) if recs else np.zeros(0, BURST_DTYPE)
    return r[np.lexsort((r["position"], r["channel"]))]
