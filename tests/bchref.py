"""Independent brute-force reference for BCH(63,51,t=2) used by the pin tests -- shares nothing with oracle/ or the kernels:
plain integer polynomial arithmetic over GF(2) and a tiny GF(64) built from x^6 + x + 1 (TIA/EIA-553 / IT++ BCH(63,2)).

A word is a 63-bit integer, bit 62 = first transmitted bit = coefficient of x^62 (the oracle's rx[j] is x^(62-j))."""
G = 0b1010100111001                      # x^12 + x^10 + x^8 + x^5 + x^4 + x^3 + 1


def polymod(a, m=G):
    dm = m.bit_length() - 1
    while a.bit_length() - 1 >= dm and a:
        a ^= m << (a.bit_length() - 1 - dm)
    return a


# GF(64): alpha = x modulo x^6 + x + 1
EXP = [0] * 126
LOG = [0] * 64
_v = 1
for _i in range(63):
    EXP[_i] = EXP[_i + 63] = _v
    LOG[_v] = _i
    _v <<= 1
    if _v & 64:
        _v ^= 0b1000011


def evaluate(word, power):
    """word(alpha^power)"""
    s = 0
    for e in range(63):
        if (word >> e) & 1:
            s ^= EXP[(e * power) % 63]
    return s


def is_cube(v):
    return v != 0 and LOG[v] % 3 == 0


def coset_leaders():
    """remainder mod g  ->  (weight, error pattern) of the unique pattern of weight <= 2 with that syndrome (d_min = 5)"""
    lead = {0: (0, 0)}
    for i in range(63):
        lead[polymod(1 << i)] = (1, 1 << i)
    for i in range(63):
        for j in range(i):
            e = (1 << i) | (1 << j)
            r = polymod(e)
            assert r not in lead          # d_min >= 5: all patterns of weight <= 2 have distinct syndromes
            lead[r] = (2, e)
    return lead


def bits(word):
    return [(word >> (62 - j)) & 1 for j in range(63)]


def from_bits(b):
    w = 0
    for j, v in enumerate(b):
        w |= int(v) << (62 - j)
    return w
