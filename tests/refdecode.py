"""A SECOND restatement of the reference's burst decode (R3, R5-R8), written separately from oracle/ref_chain.c and sharing no
code with it or with the kernels: plain Python over lists, the BCH verdict from tests/bchref.py (brute-force coset leaders + the
documented IT++ rule).  Two restatements written apart and agreeing on every field of every message class is what this repo can
offer in place of a reference-produced vector for the control flow of bursts_message (lib/recc_decode_impl.cc:81-169).

Each function cites the reference lines it follows.  Pure Python: for the handful of bursts the tests feed it."""
import bchref

_LEAD = None


def _leaders():
    global _LEAD
    if _LEAD is None:
        _LEAD = bchref.coset_leaders()
    return _LEAD


def manchester(sym, nbits):
    """lib/utils.cc:27-59: pairs (1,0) -> 0, (0,1) -> 1, (1,1) -> 0 + bad, (0,0) -> 1 + bad"""
    out, bad = [], 0
    for i in range(nbits):
        a, b = int(sym[2 * i]), int(sym[2 * i + 1])
        if (a, b) == (1, 1):
            out.append(0); bad += 1
        elif (a, b) == (0, 0):
            out.append(1); bad += 1
        elif (a, b) == (1, 0):
            out.append(0)
        else:
            assert (a, b) == (0, 1)
            out.append(1)
    return out, bad


def bch_valid(block48):
    """lib/recc_decode_impl.cc:53-79 returns itpp::BCH(63,2,true)::decode's flag on 15 zeros + the 48 bits: true when the
    word is within two flips of a code word, or in the S1 = 0 / S3-a-cube case IT++ also accepts; positions are not checked"""
    w = bchref.from_bits([0] * 15 + [int(b) & 1 for b in block48])
    r = bchref.polymod(w)
    if r in _leaders():
        return True
    return bchref.evaluate(w, 1) == 0 and bchref.is_cube(bchref.evaluate(w, 3))


def _get(bits, off, n):
    """amps_packet.h:118-143 get8 / get32 / get64: most significant bit first"""
    v = 0
    for i in range(n):
        v = (v << 1) | (int(bits[off + i]) & 1)
    return v


def extract_min_3(val):
    """amps_packet.h:277-302, quirks included (a leading digit above 9 prints as 0)"""
    m2 = val + 111
    d3 = m2 % 10
    m2 -= 10 if d3 == 0 else d3
    d2 = (m2 % 100) // 10
    m2 -= 100 if d2 == 0 else m2 % 100
    d1 = m2 // 100
    if d1 > 9:
        d1 = 0
    return "%d%d%d" % (d1, d2, d3)


def calc_min(min1, min2):
    """amps_packet.h:354-363"""
    thous = (min1 >> 10) & 0xF
    if thous > 9:
        thous = 0
    return extract_min_3(min2) + extract_min_3((min1 >> 14) & 0x3FF) + str(thous) + extract_min_3(min1 & 0x3FF)


def digits(word):
    """recc_word_called::digits, amps_packet.h:211-273: eight 4-bit codes, 0 ends, 13..15 end with a warning"""
    d, out, bad = _get(word, 4, 32), "", False
    for _ in range(8):
        v = (d >> 28) & 0xF
        if v == 0:
            break
        if v >= 13:
            bad = True
            break
        out += "0" if v == 10 else "*" if v == 11 else "#" if v == 12 else str(v)
        d = (d << 4) & 0xFFFFFFFF
    return out, bad


INVALID_WORD_A, E_ZERO, PAGE_RESPONSE, REGISTRATION, ORIGINATION, BAD_NAWC, UNKNOWN = range(7)


def decode(burst):
    """bursts_message, lib/recc_decode_impl.cc:81-169, on the 3374 symbol bytes of one burst.  Returns a dict of everything
    the message handlers are given (and of what the product's record exposes on the way)."""
    sym = [int(x) for x in burst]
    dcc, dcc_bad = manchester(sym[0:14], 7)                                    # :90-91
    words, errs = [], []
    for i in range(7):                                                          # :97-100
        w, e = manchester(sym[14 + 480 * i:14 + 480 * (i + 1)], 240)
        words.append(w); errs.append(e)
    valid, first = [], []
    for w in range(7):                                                          # :101-108: stop at the first valid repeat
        ok, rep = False, 5
        for r in range(5):
            if bch_valid(words[w][48 * r:48 * r + 48]):
                ok, rep = True, r
                break
        valid.append(ok); first.append(rep)
    out = {"dcc": dcc, "dcc_bad": dcc_bad, "manch_bad": errs, "valid": valid, "first_valid_rep": first,
           "esn": 0, "has_esn": 0, "dialed": "", "n_called_words": 0, "nawc_mismatch": False, "bad_digit": False}
    A, B = words[0], words[1]                                                   # the parsers read repeat 0 as received (:112, :117)
    a = {"F": A[0], "NAWC": _get(A, 1, 3), "T": A[4], "S": A[5], "E": A[6], "ER": A[7], "SCM": _get(A, 8, 4), "MIN1": _get(A, 12, 24)}
    b = {"F": B[0], "NAWC": _get(B, 1, 3), "MSG_TYPE": _get(B, 4, 5), "ORDQ": _get(B, 9, 3), "ORDER": _get(B, 12, 5), "LT": B[17],
         "EP": B[18], "SCM4": B[19], "MPCI": _get(B, 20, 2), "SDCC1": _get(B, 22, 2), "SDCC2": _get(B, 24, 2), "MIN2": _get(B, 26, 10)}
    out.update(a=a, b=b, min=calc_min(a["MIN1"], b["MIN2"]))
    zero_order = b["ORDER"] == 0 and b["ORDQ"] == 0 and b["MSG_TYPE"] == 0
    if not valid[0]:                                                            # :108-111
        out["cls"] = INVALID_WORD_A
    elif not a["E"]:                                                            # :113-116
        out["cls"] = E_ZERO
    elif a["T"] == 0 and zero_order:                                            # :121-122
        out["cls"] = PAGE_RESPONSE
    elif a["T"] == 1 and b["ORDER"] == 0xD:                                     # :123-138
        out["cls"] = REGISTRATION
        out["has_esn"] = a["S"]
        if a["S"] and a["NAWC"] > 1:
            c = words[2]
            out["esn"] = _get(c, 4, 32)
            out["nawc_mismatch"] = _get(c, 1, 3) != ((a["NAWC"] - 2) & 0xFF)
    elif a["T"] == 1 and (a["NAWC"] > 2 or zero_order):                         # :139-165
        nawc, nxt = a["NAWC"], 2
        out["has_esn"] = a["S"]
        if a["S"]:
            c = words[nxt]; nxt += 1
            out["esn"] = _get(c, 4, 32)
            nawc = (a["NAWC"] - 2) & 0xFF                                       # unsigned char arithmetic
            out["nawc_mismatch"] = _get(c, 1, 3) != nawc
        if nawc < 1 or nawc > 4:                                                # :155-158
            out["cls"] = BAD_NAWC
        else:
            out["cls"] = ORIGINATION
            while nawc > 0:
                d, bad = digits(words[nxt]); nxt += 1
                out["dialed"] += d
                out["bad_digit"] = out["bad_digit"] or bad
                out["n_called_words"] += 1
                nawc -= 1
    else:                                                                       # :166-168
        out["cls"] = UNKNOWN
    return out
