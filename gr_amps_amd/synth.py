"""Synthetic AMPS RECC seizure bursts: protocol bit builders and a CPFSK modulator.

Host-side utility used by tests, bench.py and smoke() to make inputs with a known answer.  It is
the transmit-side mirror of the path (what a mobile station sends), built from TIA/EIA-553 as the
reference uses it:
  * word layout of RECC words A/B/C/called-address       lib/amps_packet.h:103-209
  * Manchester map bit 0 -> symbols (1,0), bit 1 -> (0,1)  lib/recc_impl.cc:51-65, lib/amps_packet.h:53-70
  * seizure precursor: 30-bit dotting + word sync 11100010010 + 7-bit coded DCC, then every word
    repeated 5 times (240 bits)                           lib/recc_impl.cc:70,76; lib/recc_decode_impl.cc:96-107
  * BCH(48,36): generator x^12+x^10+x^8+x^5+x^4+x^3+1, systematic, MSB first
  * FM deviation +-8 kHz, 10 kbit/s                        grc/ampsbs.grc:209,614
"""
import numpy as np

BCH_GEN = 0b1010100111001  # x^12 + x^10 + x^8 + x^5 + x^4 + x^3 + 1
DOTTING_BITS = 30
WORD_SYNC = "11100010010"
CODED_DCC = {0: "0000000", 1: "0011111", 2: "1100011", 3: "1111100"}  # TIA-553 table 2.7.1-1
BURST_PREFIX_BITS = DOTTING_BITS + len(WORD_SYNC)  # 41 bits before the coded DCC
CAPTURE_SYMS = 3374
TRIGGER_SYMS = 74


def bits_from_int(val, n):
    return [(val >> (n - 1 - i)) & 1 for i in range(n)]


def bch_parity(msg_bits):
    """12 parity bits of the shortened (63,51) code, message MSB first."""
    rem = 0
    for b in msg_bits:
        fb = ((rem >> 11) & 1) ^ (int(b) & 1)
        rem = (rem << 1) & 0xFFF
        if fb:
            rem ^= BCH_GEN & 0xFFF
    return bits_from_int(rem, 12)


def bch_encode(msg_bits):
    msg_bits = [int(b) & 1 for b in msg_bits]
    return msg_bits + bch_parity(msg_bits)


def min_to_fields(min10: str):
    """10-digit MIN -> (MIN1 24 bit, MIN2 10 bit), TIA-553 2.3.1 (as lib/amps_packet.h:305-349)."""
    def d3(s):
        a, b, c = (10 if ch == "0" else int(ch) for ch in s)
        return 100 * a + 10 * b + c - 111
    thous = 10 if min10[6] == "0" else int(min10[6])
    min1 = ((d3(min10[3:6]) & 0x3FF) << 14) | ((thous & 0xF) << 10) | (d3(min10[7:10]) & 0x3FF)
    return min1, d3(min10[0:3]) & 0x3FF


def word_a(nawc, T, S, E, ER, scm, min1, F=1):
    return [F] + bits_from_int(nawc, 3) + [T, S, E, ER] + bits_from_int(scm, 4) + bits_from_int(min1, 24)


def word_b(nawc, msg_type, ordq, order, lt, ep, scm4, mpci, sdcc1, sdcc2, min2, F=0):
    return ([F] + bits_from_int(nawc, 3) + bits_from_int(msg_type, 5) + bits_from_int(ordq, 3)
            + bits_from_int(order, 5) + [lt, ep, scm4] + bits_from_int(mpci, 2) + bits_from_int(sdcc1, 2)
            + bits_from_int(sdcc2, 2) + bits_from_int(min2, 10))


def word_c_serial(nawc, esn, F=0):
    return [F] + bits_from_int(nawc, 3) + bits_from_int(esn, 32)


def dial_code(ch):
    return {"0": 10, "*": 11, "#": 12}.get(ch, int(ch) if ch.isdigit() else 0)


def words_called(digits: str, nawc_after_last=0):
    """Called-address words (8 digits each, zero filled)."""
    words = []
    chunks = [digits[i:i + 8] for i in range(0, max(len(digits), 1), 8)]
    for k, chunk in enumerate(chunks):
        v = 0
        for i in range(8):
            v = (v << 4) | (dial_code(chunk[i]) if i < len(chunk) else 0)
        words.append([0] + bits_from_int(len(chunks) - 1 - k + nawc_after_last, 3) + bits_from_int(v, 32))
    return words


def make_message(kind, min10, esn=0, dialed="", scm=0b0110, rng=None):
    """Return the list of 36-bit words of a page response / registration / origination."""
    min1, min2 = min_to_fields(min10)
    if kind == "page_response":
        return [word_a(1, 0, 0, 1, 0, scm, min1), word_b(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, min2)]
    if kind == "registration":
        return [word_a(2, 1, 1, 1, 0, scm, min1), word_b(1, 0, 0, 0xD, 0, 0, 0, 0, 0, 0, min2),
                word_c_serial(0, esn)]
    if kind == "origination":
        called = words_called(dialed)
        nawc = 2 + len(called)
        return [word_a(nawc, 1, 1, 1, 0, scm, min1), word_b(nawc - 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, min2),
                word_c_serial(nawc - 2, esn)] + called
    raise ValueError(kind)


def random_message(rng):
    kind = ("page_response", "registration", "origination")[int(rng.integers(0, 3))]
    min10 = "".join(str(int(d)) for d in rng.integers(0, 10, 10))
    esn = int(rng.integers(0, 2 ** 32))
    nd = int(rng.integers(1, 17))
    dialed = "".join("1234567890*#"[int(i)] for i in rng.integers(0, 12, nd))
    return kind, min10, esn, dialed, make_message(kind, min10, esn, dialed)


def burst_bits(words36, dcc=0, pad_words=True, rng=None):
    """Bits of one seizure burst: dotting, word sync, coded DCC, each word x5.  With pad_words the
    unused word slots (up to 7) carry BCH-valid random filler so the 3374-symbol capture is fully
    defined by the transmitter (a real burst just ends: see `pad_words=False`)."""
    bits = [1, 0] * (DOTTING_BITS // 2) + [int(c) for c in WORD_SYNC] + [int(c) for c in CODED_DCC[dcc]]
    words = [list(w) for w in words36]
    if pad_words:
        rng = rng or np.random.default_rng(0)
        while len(words) < 7:
            words.append([0] + [int(b) for b in rng.integers(0, 2, 35)])
    for w in words:
        cw = bch_encode(w)
        bits += cw * 5
    return bits


def manchester(bits):
    """bit 0 -> (1,0); bit 1 -> (0,1)  (lib/recc_impl.cc:54-59)"""
    b = np.asarray(bits, np.uint8)
    out = np.empty(2 * b.size, np.uint8)
    out[0::2] = 1 - b
    out[1::2] = b
    return out


def symbol_stream(n_symbols, bursts, rng, idle="random"):
    """u8 0/1 symbol stream with `bursts` = [(offset, bits)] spliced in; idle = random symbols."""
    s = rng.integers(0, 2, n_symbols).astype(np.uint8) if idle == "random" else np.zeros(n_symbols, np.uint8)
    for off, bits in bursts:
        m = manchester(bits)
        s[off:off + m.size] = m[:max(0, n_symbols - off)]
    return s


def symbol_waveform(sym, sps, sym_ppm=0.0):
    """+-1 symbol values held for sps / (1 + sym_ppm 1e-6) samples each: a mobile whose bit clock runs sym_ppm parts per million
    fast (TIA-553 allows 10 kbit/s +- 1 bit/s = +-100 ppm).  sym_ppm = 0 is np.repeat(sym, sps)."""
    if sym_ppm == 0.0:
        return np.repeat(sym, sps)
    rate = (1.0 + sym_ppm * 1e-6) / sps                       # symbols per sample
    n = int(np.floor(len(sym) / rate))
    idx = np.minimum((np.arange(n) * rate).astype(np.int64), len(sym) - 1)
    return np.asarray(sym)[idx]


def fsk_modulate(n_samples, bursts, sps=10, fs=200e3, dev=8e3, snr_db=30.0, rng=None, dtype=np.complex64, sym_ppm=0.0, cfo_hz=0.0):
    """Complex baseband: carrier only during a burst (unit amplitude CPFSK, symbol 1 -> +dev,
    symbol 0 -> -dev), AWGN everywhere.  `bursts` = [(sample_offset, bits)].  Impairments of the mobile: sym_ppm = symbol-clock
    offset in parts per million, cfo_hz = carrier offset (both apply to every burst; the defaults draw exactly the samples
    the unimpaired generator always drew)."""
    rng = rng or np.random.default_rng(0)
    sigma = 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0)
    x = (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples)) * sigma
    for off, bits in bursts:
        sym = manchester(bits).astype(np.float64) * 2.0 - 1.0
        f = symbol_waveform(sym, sps, sym_ppm) * dev + cfo_hz
        n = min(f.size, n_samples - off)
        if n <= 0:
            continue
        ph = 2.0 * np.pi * np.cumsum(f[:n]) / fs + rng.uniform(0, 2 * np.pi)
        x[off:off + n] += np.exp(1j * ph)
    return x.astype(dtype)


def make_channel_block(n_samples, n_bursts, seed, sps=10, snr_db=30.0, first=4000, spacing=None, jitter=True, sym_ppm=0.0, cfo_hz=0.0):
    """One channel of config-1 style input with `n_bursts` random messages; returns (iq, truth)
    where truth = [(sample_offset, kind, min10, esn, dialed, words36)]."""
    rng = np.random.default_rng(seed)
    burst_len = (BURST_PREFIX_BITS + 7 + 7 * 240) * 2 * sps
    spacing = spacing or (burst_len + (TRIGGER_SYMS + 200) * sps)
    truth, bursts = [], []
    off = first + (int(rng.integers(0, 997)) if jitter else 0)
    for _ in range(n_bursts):
        if off + burst_len + 2 * sps > n_samples:
            break
        kind, min10, esn, dialed, words = random_message(rng)
        bits = burst_bits(words, dcc=int(rng.integers(0, 4)), rng=rng)
        bursts.append((off, bits))
        truth.append((off, kind, min10, esn, dialed, words))
        off += spacing + (int(rng.integers(0, 997)) if jitter else 0)
    iq = fsk_modulate(n_samples, bursts, sps=sps, fs=20e3 * sps, snr_db=snr_db, rng=rng, sym_ppm=sym_ppm, cfo_hz=cfo_hz)
    return iq, truth
