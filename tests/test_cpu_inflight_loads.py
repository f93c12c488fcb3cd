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

LOAD = re.compile(r"\s*global_load_dwordx2 (v\[\d+:\d+\]), v\d+, s\[")


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


def scan(asm_text):
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
        waits = [k for k, l in enumerate(lines) if "s_waitcnt vmcnt(8)" in l]
        loads = [(k, _regs(LOAD.match(l).group(1))) for k, l in enumerate(lines) if LOAD.match(l)]
        # the unrolled loop that holds the asm loads: its waits are the ones with loads between them and their successor
        loop = [w for n, w in enumerate(waits) if any(w < k < (waits[n + 1] if n + 1 < len(waits) else w + 2000) for k, _ in loads)]
        issues = []
        for k, dst in loads:
            nxt = [w for w in loop if w > k]
            if len(nxt) >= 2:
                segs = [range(k + 1, nxt[1])]
            else:       # the last two half-steps of the unrolled period: on to the loop's end, then from its head to the covering wait
                bar = next(q for q in range(loop[-1], len(lines)) if "s_barrier" in lines[q])    # the last half-step ends at its barrier
                tail_end = min(len(lines), bar + 12)                                              # (+ the loop's back branch)
                segs = [range(k + 1, tail_end), range(max(0, loop[0] - 30), loop[1 - len(nxt)])]
            for seg in segs:
                hit = _touched(lines, seg, dst)
                if hit:
                    issues.append((m.group(1), k + 1, hit[0] + 1, hit[1]))
                    break
        out.append((m.group(1), len(loads), len(loop), issues))
        i = j
    return out


def test_scanner_sees_a_planted_hazard():
    asm = "\n".join(["_ZN4amps12chz12_kernelXX:", "s_waitcnt vmcnt(8)", "global_load_dwordx2 v[10:11], v1, s[2:3]", "v_mov_b64_e32 v[20:21], v[10:11]",
                     "s_waitcnt vmcnt(8)", "global_load_dwordx2 v[12:13], v1, s[2:3]", "s_waitcnt vmcnt(8)", "v_add_f32 v0, v10, v12", "s_barrier", ".Lfunc_end0:"])
    res = scan(asm)
    assert len(res) == 1 and len(res[0][3]) >= 1 and "v_mov_b64" in res[0][3][0][3]


def test_no_register_with_a_load_in_flight_is_touched():
    if not os.path.exists(build.hipcc()):
        pytest.skip("hipcc not installed")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "amps.s")
        subprocess.run([build.hipcc()] + flags + ["--cuda-device-only", "-S", "-I" + os.path.join(build._ROOT, "include"), "-I" + build.CSRC,
                                                  os.path.join(build.CSRC, "amps_recc.hip"), "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        res = scan(open(out).read())
    assert len(res) == 5, [r[0] for r in res]                      # the unfused form and the four slicer specs
    for name, nloads, nwaits, issues in res:
        assert nloads == 48 and nwaits == 6, (name, nloads, nwaits)   # six unrolled half-steps of eight loads
        assert not issues, (name, issues[:4])
