"""A SECOND statement of what the fused seams do BEHIND the slicer -- trigger test, runs of matching phases, hold-off, the wait for a
burst's tail, the capture with its timing tracking -- written from the prose of DESIGN.md 4.1 / 4.1b / 4.4 / 4.4b and
include/amps_recc_numerics.h in vectorised numpy over whole arrays, sharing no code with oracle/fused_model.c (a per-sample C loop)
or with the kernels.  Input: the slicer bits g[n] of ONE channel, whatever slicer spec produced them; output: (n_c, 3374 symbols)
of every burst the seam captures.  tests/test_cpu_tracking_restatement.py holds the CPU model to it (and, through
tests/refdecode.py, every record field); two statements by the same hand agreeing is what this repo can offer for a rule the
reference does not have.

Vocabulary (TIA-553 / lib/recc_impl.cc): a burst opens with 30 bits of dotting 1010.. and the 11-bit word sync 11100010010; the
reference's trigger is the Manchester image of the last 37 of those 41 bits, 74 symbols; the capture is the 3374 symbols behind it."""
from fractions import Fraction

import numpy as np

CAPTURE_SYMS, TRIGGER_SYMS, WORD_BITS, REPEATS, WORDS = 3374, 74, 48, 5, 7
DEDUP_SYMBOLS, TRACK_BLOCKS, WORD_SAMPLES = 2, 36, 64


def trigger_symbols():
    """Manchester: bit 1 -> symbols (0, 1), bit 0 -> (1, 0) (lib/utils.cc:27-59 read backwards)"""
    bits = ([1, 0] * 15 + [1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0])[-TRIGGER_SYMS // 2:]
    return np.array([s for b in bits for s in ((0, 1) if b else (1, 0))], np.uint8)


def matches(g, sps, tol=0):
    """M[n] = the 74 trigger symbols, one every sps samples and the last one AT n, differ from the slicer bits in at most tol places;
    what lies in front of the stream reads 1"""
    t = trigger_symbols()
    pad = sps * (TRIGGER_SYMS - 1)
    gp = np.concatenate([np.ones(pad, np.uint8), np.asarray(g, np.uint8)])
    wrong = np.zeros(len(g), np.int32)
    for k in range(TRIGGER_SYMS):
        wrong += gp[k * sps:k * sps + len(g)] != t[k]
    return wrong <= tol


def captures(g, sps, tol=0, track=True, n_done=None):
    """All (n_c, symbols) of a stream whose first n_done samples (a multiple of 64; default: as many whole 64-sample words as g holds)
    have been processed, in stream order."""
    g = np.asarray(g, np.uint8)
    n_done = (len(g) // WORD_SAMPLES) * WORD_SAMPLES if n_done is None else n_done
    g = g[:n_done]
    M = matches(g, sps, tol)
    D = DEDUP_SYMBOLS * sps
    # a RUN of matching phases starts where a match has no match among the D samples before it ...
    cum = np.concatenate([[0], np.cumsum(M)])
    idx = np.nonzero(M)[0]
    before = cum[idx] - cum[np.maximum(idx - D, 0)]
    starts = idx[before == 0]
    # ... and a start is looked at once the 64-sample word BEHIND its own has been processed (the run may reach into it)
    starts = starts[starts // WORD_SAMPLES + 1 < n_done // WORD_SAMPLES]
    span = sps * (CAPTURE_SYMS + 1) + (TRACK_BLOCKS if track else 0)       # what must follow n_c: strictly more than the capture, plus the most the timing can move
    out, next_allowed = [], 0
    for p in starts:
        if p < next_allowed:
            continue                                                        # inside the last accepted burst: trigger + capture hold the search off
        run = np.nonzero(M[p:p + D])[0]
        nc = int(p) + int(run[-1]) // 2                                     # the run's centre is the symbol timing
        next_allowed = nc + sps * (CAPTURE_SYMS + TRIGGER_SYMS)
        if nc + span < n_done:                                              # else it waits for its tail (and, at the end of a stream, for ever)
            out.append((nc, capture(g, nc, sps, track)))
    return out


def _bit(g, n):
    n = np.asarray(n)
    return np.where(n < 0, 1, g[np.maximum(n, 0)])


def capture(g, nc, sps, track=True):
    """Symbol i of the capture = slicer bit n_c + sps (i + 1) + delay; the delay starts at 0 and may change by one sample after each of
    36 blocks of bits: the trigger's own 37 bits (in front of the capture: measured, not kept), the 7 bits of the coded DCC together
    with the first 48-bit repeat, then the other 34 repeats.  From three samples per symbol on, a block's mid-bit transitions move everything
    BEHIND it; at two samples per symbol a block chooses its OWN delay by its Manchester violations."""
    sizes = [TRIGGER_SYMS // 2, 7 + WORD_BITS] + [WORD_BITS] * (WORDS * REPEATS - 1)
    assert len(sizes) == TRACK_BLOCKS and sum(sizes[1:]) * 2 == CAPTURE_SYMS
    sym = np.zeros(CAPTURE_SYMS, np.uint8)
    delay, first = 0, -(TRIGGER_SYMS // 2)
    for nb in sizes:
        k = np.arange(first, first + nb)
        if track and sps == 2:
            # two samples per symbol: the block is taken at whichever of the previous block's delay and its two neighbours shows the
            # fewest Manchester violations (both symbols of a bit equal) in this very block; the old delay wins a tie, then the earlier
            t3 = nc + sps * (2 * k[None, :] + 1) + delay + np.array([0, -1, 1])[:, None]
            viol = (_bit(g, t3) == _bit(g, t3 + sps)).sum(1)
            delay += (0, -1, 1)[int(np.argmin(viol))]                       # argmin returns the FIRST minimum: order (d, d - 1, d + 1)
        t = nc + sps * (2 * k + 1) + delay                                  # first sampling instant of bit k; the second is one symbol on
        a, b = _bit(g, t), _bit(g, t + sps)
        keep = k >= 0
        sym[2 * k[keep]], sym[2 * k[keep] + 1] = a[keep], b[keep]
        if track and sps > 2:
            between = _bit(g, t[:, None] + np.arange(1, sps)[None, :])      # the sps - 1 slicer bits between the two instants
            late = (between == a[:, None]).sum(1) - Fraction(sps - 1, 2)    # bits still equal to a: how late the mid-bit transition came
            pair = a != b                                                   # only a valid Manchester pair has a transition to measure
            falling, rising = late[pair & (a == 1)], late[pair & (a == 0)]
            if len(falling) and len(rising):                                # a carrier offset moves the two polarities apart: averaged separately
                m = sum(falling, Fraction(0)) / len(falling) + sum(rising, Fraction(0)) / len(rising)
                delay += 1 if m > 1 else -1 if m < -1 else 0
        first += nb
    return sym
