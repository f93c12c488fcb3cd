"""not gpu: nothing touches a register while an untracked load into it is in flight.

The fold role of chz12_kernel prefetches its inputs with inline-asm `global_load_dwordx2` (recc_channelizer.hip.h:
chz_load1_ring) so that the compiler's own vmcnt bookkeeping does not drain them at every barrier; the wave waits with a manual
`s_waitcnt vmcnt(8)` two half-steps later.  The compiler therefore believes the destination registers hold their values from the
moment the asm statement ends -- and is free to copy them (register-allocation copies at control-flow joins), which on the
hardware reads a register whose load has not landed.  It did exactly that once (wrong frames from the 12th on).  This test
scans the gfx950 assembly of every instantiation of the kernel: between an asm load and the second `s_waitcnt vmcnt(8)` that
follows it on the loop's path (the wait that covers it), no instruction may name one of its destination registers."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from gr_amps_amd import build

# the fast loader's untracked loads: scalar-base global loads (D = 512) or raw buffer loads against the workgroup's descriptor (D = 768)
LOAD = re.compile(r"\s*(?:global_load_dwordx2|buffer_load_dwordx2) (v\[\d+:\d+\]), v\d+, s\[")


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _touched(lines, rng, dst):
    for q in rng:
        t = lines[q].strip()
        if not t or t[0] in ";.":
            continue
        used = set()
        for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
            used |= _regs(tk)
        if used & dst:
            return q, t
    return None


LATCH = re.compile(r"\s*s_branch\s+(\.LBB\d+_\d+)")
WAIT = re.compile(r"\s*s_waitcnt vmcnt\((\d+)\)")


def scan(asm_text):
    """Per chz12_kernel instantiation: (name, asm loads, waits of the loop that holds them, hazards).  A load is followed along the
    loop's PATH: straight through the text, and at an unconditional `s_branch` to an EARLIER label (the loop's latch -- block
    placement rotates the loop, so its last half-step ends in that branch and not in the text that happens to follow it) on at the
    label.  Conditional branches are not taken: inside the loop they are its exits and the skips of its rare paths, wherever their
    targets were placed.  Vector memory returns in order, so `s_waitcnt vmcnt(N)` covers a load exactly when at least N younger
    loads have been issued behind it (D = 512: vmcnt(8), the second wait the walk meets; D = 768: vmcnt(13), the third or fourth):
    the walk counts the asm loads it passes and ends at the first wait whose N they reach.  Nothing before that may name one of the
    load's destination registers -- a spill store of a ring slot is how round 6 met this -- and a walk that never gets there is
    reported too."""
    txt = asm_text.split("\n")
    out, i = [], 0
    while i < len(txt):
        m = re.match(r"^(_ZN4amps12chz12_kernel\w+):", txt[i])
        if not m:
            i += 1
            continue
        j = i
        while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
            j += 1
        lines = txt[i:j]
        labels = {}
        for k, l in enumerate(lines):
            lm = re.match(r"^(\.LBB\d+_\d+):", l)
            if lm:
                labels[lm.group(1)] = k
        loads = [(k, _regs(LOAD.match(l).group(1))) for k, l in enumerate(lines) if LOAD.match(l)]
        # the manual waits of the unrolled loop that holds the asm loads: those with asm loads between them and their successor
        waits = [k for k, l in enumerate(lines) if WAIT.match(l) and int(WAIT.match(l).group(1)) in (8, 13)]
        loop = [w for n, w in enumerate(waits) if any(w < k < (waits[n + 1] if n + 1 < len(waits) else w + 2000) for k, _ in loads)]
        issues = []
        for k, dst in loads:
            pos, younger, latched, covered, steps = k + 1, 0, set(), False, 0
            while pos < len(lines) and steps < 50000:
                steps += 1
                t = lines[pos].strip()
                wm = WAIT.match(lines[pos])
                lm = LOAD.match(lines[pos])
                if wm and younger >= int(wm.group(1)):
                    covered = True
                    break
                if lm:
                    younger += 1
                    if _regs(lm.group(1)) & dst:
                        issues.append((m.group(1), k + 1, pos + 1, "loaded again before the covering wait: " + t))
                        covered = True
                        break
                elif t and t[0] not in ";.":
                    used = set()
                    for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                        used |= _regs(tk)
                    if used & dst:
                        issues.append((m.group(1), k + 1, pos + 1, t))
                        covered = True
                        break
                    bm = LATCH.match(lines[pos])
                    if bm and labels.get(bm.group(1), len(lines)) < pos and pos not in latched:
                        latched.add(pos)
                        pos = labels[bm.group(1)]
                        continue
                    if "s_endpgm" in t:
                        break
                pos += 1
            if not covered:
                issues.append((m.group(1), k + 1, pos, "path left the kernel before the covering wait"))
        out.append((m.group(1), len(loads), len(loop), issues))
        i = j
    return out


def test_scanner_sees_a_planted_hazard():
    asm = "\n".join(["_ZN4amps12chz12_kernelXX:", ".LBB0_1:", "s_waitcnt vmcnt(1)", "global_load_dwordx2 v[10:11], v1, s[2:3]",
                     "v_mov_b64_e32 v[20:21], v[10:11]", "s_cbranch_scc1 .LBB0_1", "s_barrier", "s_waitcnt vmcnt(1)",
                     "global_load_dwordx2 v[12:13], v1, s[2:3]", "s_barrier", "s_branch .LBB0_1", "v_add_f32 v0, v10, v12", ".Lfunc_end0:"])
    res = scan(asm)
    assert len(res) == 1 and len(res[0][3]) == 1 and "v_mov_b64" in res[0][3][0][3]
    # a spill store of a register whose load is in flight (what the D = 768 unfused kernel did before it lost its fast loader), and
    # the same code with the store behind the wait that covers the load
    bad = ["_ZN4amps12chz12_kernelYY:", ".LBB1_1:", "s_waitcnt vmcnt(1)", "global_load_dwordx2 v[10:11], v1, s[2:3]",
           "scratch_store_dwordx2 off, v[10:11], off", "global_load_dwordx2 v[12:13], v1, s[2:3]", "s_waitcnt vmcnt(0)", "s_endpgm", ".Lfunc_end1:"]
    good = bad[:4] + bad[5:7] + [bad[4]] + bad[7:]
    assert any("scratch_store" in h[3] for h in scan("\n".join(bad))[0][3]) and not scan("\n".join(good))[0][3]


@pytest.fixture(scope="module")
def asm_text():
    if not os.path.exists(build.hipcc()):
        pytest.skip("hipcc not installed")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "amps.s")
        subprocess.run([build.hipcc()] + flags + ["--cuda-device-only", "-S", "-I" + os.path.join(build._ROOT, "include"), "-I" + build.CSRC,
                                                  os.path.join(build.CSRC, "amps_recc.hip"), "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def test_no_register_with_a_load_in_flight_is_touched(asm_text):
    res = scan(asm_text)
    assert len(res) == 10, [r[0] for r in res]                     # the unfused form and the four slicer specs, at D = 512 and at D = 768
    for name, nloads, nwaits, issues in res:
        if "Li768E" in name and "Lin1E" in name:
            assert nloads == 0, (name, nloads)                     # the unfused form at D = 768 has no fast loader (recc_channelizer.hip.h)
        elif "Li768E" in name:
            assert nloads == 48 and nwaits == 8, (name, nloads, nwaits)   # four unrolled half-steps of twelve loads and two waits
        else:
            assert nloads == 48 and nwaits == 6, (name, nloads, nwaits)   # six unrolled half-steps of eight loads
        assert not issues, (name, issues[:4])


def role_bodies_with_scratch(asm):
    """Per chz12_kernel instantiation: the basic blocks that are the body of a role -- recognised by their arithmetic: a fold step
    (>= 100 v_pk_fma_f32), a radix-16 pass (>= 60 v_pk_add_f32) or a slicer step (>= 20 v_alignbit_b32) -- and contain a scratch
    instruction.  Round 5's spec D slicer (two frame buffers used alternately) lets the compiler spill a few row addresses that are
    reloaded once per 128 frames; that is fine ONLY as long as no role body touches scratch: a first version that did spilled inside
    the word step and was 10 % slower (profiles/EXPERIMENTS.md)."""
    txt = asm.split("\n")
    bad, i = [], 0
    while i < len(txt):
        m = re.match(r"^(_ZN4amps12chz12_kernel\w+):", txt[i])
        if not m:
            i += 1
            continue
        j = i
        while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
            j += 1
        cur, counts = None, {}
        for l in txt[i:j]:
            lm = re.match(r"^(\.LBB\d+_\d+):", l)
            if lm:
                cur = lm.group(1)
                counts[cur] = {"fma": 0, "add": 0, "align": 0, "scratch": 0, "edge": 0, "tests": 0}
                continue
            if cur is None:
                continue
            t = l.strip()
            c = counts[cur]
            c["fma"] += t.startswith("v_pk_fma_f32")
            c["add"] += t.startswith("v_pk_add_f32")
            c["align"] += t.startswith("v_alignbit_b32")
            c["scratch"] += t.startswith("scratch_")
            c["edge"] += bool(re.match(r"global_load_dwordx2 v\[\d+:\d+\], v\[\d+:\d+\], off", t))
            c["tests"] += t.startswith("s_cbranch")
        # (an EDGE half-step of the fold -- bounds-checked loads with 64-bit lane addresses: the head of a launch -- is no steady body, and
        # neither is a generic time step of the FFT / slicer roles: the five at either end of a workgroup's range test for the rare cases
        # inside their body, a steady step is straight-line code up to its barrier and the loop's one branch)
        bad += [(m.group(1), b, c) for b, c in counts.items()
                if c["scratch"] and not c["edge"] and c["tests"] <= 1 and (c["fma"] >= 100 or c["add"] >= 60 or c["align"] >= 20)]
        i = j
    return bad


def test_scratch_scanner_sees_a_planted_spill():
    body = ["v_alignbit_b32 v1, v2, v3, 31"] * 24
    asm = "\n".join(["_ZN4amps12chz12_kernelXX:", ".LBB0_1:"] + body + [".LBB0_2:"] + body + ["scratch_load_dword v0, off, off"] + [".Lfunc_end0:"])
    assert [b for _, b, _ in role_bodies_with_scratch(asm)] == [".LBB0_2"]


def test_no_role_body_of_the_filter_bank_touches_scratch(asm_text):
    assert role_bodies_with_scratch(asm_text) == []
