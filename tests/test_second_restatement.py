"""tests/refdecode.py (a second restatement of bursts_message, written apart from oracle/ref_chain.c) against the oracle on CPU
and against the HIP decode kernel on the GPU: random bursts steered into every message class, with bit errors (so that the
first-valid-of-five rule and the 'parse repeat 0 as received' quirk matter) and non-Manchester symbol pairs."""
import numpy as np
import pytest

import bchref
import oracle
import refdecode
from gr_amps_amd import capi


def _cw(msg36):
    m = 0
    for b in msg36:
        m = (m << 1) | int(b)
    m <<= 12
    w = m | bchref.polymod(m)
    return [(w >> (47 - i)) & 1 for i in range(48)]


def _bits(v, n):
    return [(v >> (n - 1 - i)) & 1 for i in range(n)]


def make_burst(rng, kind):
    nawc_a = int(rng.integers(0, 8))
    T, S, E = int(rng.integers(0, 2)), int(rng.integers(0, 2)), 1
    order, ordq, mtype = 0, 0, 0
    if kind == "page":
        T = 0
    elif kind == "registration":
        T, order = 1, 0xD
        ordq, mtype = int(rng.integers(0, 8)), int(rng.integers(0, 32))
    elif kind == "origination":
        T = 1
        nawc_a = int(rng.integers(1, 5)) + (2 if S else 0)
        if rng.integers(0, 2):
            order, ordq, mtype = int(rng.integers(0, 13)), int(rng.integers(0, 8)), int(rng.integers(0, 32))
            if order or ordq or mtype:
                nawc_a = max(nawc_a, 3)
    elif kind == "bad_nawc":
        T, S = 1, int(rng.integers(0, 2))
        nawc_a = int(rng.choice([0, 5, 6, 7])) if not S else int(rng.choice([0, 1, 2, 7]))
        if nawc_a <= 2:
            order, ordq, mtype = 0, 0, 0
        else:
            order = int(rng.integers(0, 13))
    elif kind == "e_zero":
        E = 0
    elif kind == "unknown":
        T, nawc_a, order = 1, int(rng.integers(0, 3)), int(rng.choice([1, 2, 3, 7, 0xC, 0xE, 0x1F]))
    elif kind == "random":
        T, S, E = (int(x) for x in rng.integers(0, 2, 3))
        order, ordq, mtype = int(rng.integers(0, 32)), int(rng.integers(0, 8)), int(rng.integers(0, 32))
    words = []
    words.append([1] + _bits(nawc_a, 3) + [T, S, E, int(rng.integers(0, 2))] + _bits(int(rng.integers(0, 16)), 4) + _bits(int(rng.integers(0, 1 << 24)), 24))
    words.append([0] + _bits(int(rng.integers(0, 8)), 3) + _bits(mtype, 5) + _bits(ordq, 3) + _bits(order, 5) +
                 [int(x) for x in rng.integers(0, 2, 3)] + _bits(int(rng.integers(0, 4)), 2) + _bits(int(rng.integers(0, 4)), 2) +
                 _bits(int(rng.integers(0, 4)), 2) + _bits(int(rng.integers(0, 1 << 10)), 10))
    for _ in range(5):
        if rng.integers(0, 3) == 0:          # a serial-number word
            nw = int(rng.integers(0, 8)) if rng.integers(0, 4) == 0 else (nawc_a - 2) & 7
            words.append([0] + _bits(nw, 3) + _bits(int(rng.integers(0, 1 << 32)), 32))
        else:                                # a called-address word: digit codes incl. terminators and invalid codes
            codes = [int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 13, 15], p=[.08] * 12 + [.02, .01, .01])) for _ in range(8)]
            d = 0
            for c in codes:
                d = (d << 4) | c
            words.append([0] + _bits(int(rng.integers(0, 8)), 3) + _bits(d, 32))
    bits = [int(x) for x in rng.integers(0, 2, 7)]                 # DCC: seven free bits here
    for wi, w in enumerate(words):
        cw = _cw(w)
        for r in range(5):
            while True:
                rep = list(cw)
                ne = int(rng.choice([0, 0, 0, 1, 2, 3, 4])) if not (kind == "invalid_a" and wi == 0) else int(rng.integers(3, 7))
                for p in rng.choice(48, size=ne, replace=False):
                    rep[int(p)] ^= 1
                if not (kind == "invalid_a" and wi == 0) or not refdecode.bch_valid(rep):
                    break                                              # half of all 63-bit words decode: draw until this one does not
            bits += rep
    sym = np.empty(2 * len(bits), np.uint8)
    sym[0::2] = [1 - b for b in bits]                              # '0' -> (1,0), '1' -> (0,1)
    sym[1::2] = bits
    for p in rng.choice(len(sym), size=int(rng.choice([0, 0, 3, 12])), replace=False):
        sym[int(p)] ^= 1                                           # (1,1) / (0,0) pairs: decoded with the reference's bias, counted bad
    assert len(sym) == 3374
    return sym


KINDS = ["page", "registration", "origination", "bad_nawc", "e_zero", "unknown", "invalid_a", "random"]


def _bursts(n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([make_burst(rng, KINDS[i % len(KINDS)]) for i in range(n)])


def _check(rec, want, i):
    assert list(rec["dcc"]) == want["dcc"] and int(rec["dcc_bad"]) == want["dcc_bad"], i
    assert list(rec["manch_bad"]) == want["manch_bad"], i
    assert [bool(v) for v in rec["valid"]] == want["valid"] and list(rec["first_valid_rep"]) == want["first_valid_rep"], i
    a, b = want["a"], want["b"]
    for k in ("F", "NAWC", "T", "S", "E", "ER", "SCM", "MIN1"):
        assert int(rec["a_" + k]) == a[k], (i, "a_" + k)
    for k in ("F", "NAWC", "MSG_TYPE", "ORDQ", "ORDER", "LT", "EP", "SCM4", "MPCI", "SDCC1", "SDCC2", "MIN2"):
        assert int(rec["b_" + k]) == b[k], (i, "b_" + k)
    assert rec["min"].decode() == want["min"], i
    assert int(rec["msg_class"]) == want["cls"], (i, int(rec["msg_class"]), want["cls"])
    assert int(rec["esn"]) == want["esn"] and int(rec["has_esn"]) == want["has_esn"], i
    assert rec["dialed"].decode() == want["dialed"] and int(rec["n_called_words"]) == want["n_called_words"], i
    assert bool(int(rec["flags"]) & 2) == want["nawc_mismatch"] and bool(int(rec["flags"]) & 4) == want["bad_digit"], i


def test_oracle_equals_the_second_restatement():
    bursts = _bursts(160, 2024)
    recs = oracle.decode_bursts(bursts)
    seen = set()
    for i in range(len(bursts)):
        want = refdecode.decode(bursts[i])
        _check(recs[i], want, i)
        seen.add(want["cls"])
    assert seen == set(range(7))                                   # every branch of bursts_message was taken


@pytest.mark.gpu
def test_hip_decode_equals_the_second_restatement(gpu):
    bursts = _bursts(96, 77)
    with capi.Recc(n_channels=1, max_bursts=8) as r:
        recs = r.decode_bursts(bursts)
    for i in range(len(bursts)):
        _check(recs[i], refdecode.decode(bursts[i]), i)
