"""CPU tests of the boundary: the C-ABI library loads without a GPU, exports every symbol that
include/amps_recc.h declares, refuses to run without a device (no CPU fallback), and its host-only
entry point (reply generation) equals the oracle's restatement of lib/recc_decode_impl.cc:181-272."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from gr_amps_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "amps_recc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(amps_(?:recc|bch)_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 14 and set(declared) == set(capi.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.amps_recc_abi_version() == 4
    assert L.amps_recc_burst_size() == capi.BURST_DTYPE.itemsize == oracle.BURST_DTYPE.itemsize == 728
    assert capi.BURST_DTYPE == oracle.BURST_DTYPE
    assert b"no usable HIP device" in L.amps_recc_strerror(-19)


def test_header_cites_the_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "amps_recc.h")).read()
    for cite in ("lib/recc_impl.cc:93-145", "lib/recc_decode_impl.cc:81-169", "lib/utils.cc:27-59",
                 "lib/amps_packet.h:103-274", "grc/recctest.grc"):
        assert cite in hdr


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.AmpsError) as e:
        capi.Recc(n_channels=1, sps=10, max_samples=4096)
    assert e.value.code == -19          # -ENODEV


def test_argument_validation_needs_no_gpu():
    L = capi.load()
    h = C.c_void_p()
    assert L.amps_recc_create(C.byref(h), None) == -22            # -EINVAL
    cfg = capi.Cfg()
    cfg.struct_size = 4                                           # wrong ABI size
    assert L.amps_recc_create(C.byref(h), C.byref(cfg)) == -22
    n = C.c_size_t(0)
    assert L.amps_recc_drain(None, None, 0, C.byref(n)) == -22
    assert L.amps_recc_push_iq(None, None, 0, 0, 0) == -22
    # the collective entry points (ABI 4) refuse a missing handle the same way, without touching a device or librccl
    assert L.amps_recc_rccl_init(None, None, 1, 0) == -22
    assert L.amps_recc_push_wideband_dist(None, None, 0, 0, 0, 0, C.byref(n)) == -22 and n.value == 0
    assert L.amps_recc_push_wideband_bcast(None, None, 0, 0, 0) == -22
    assert L.amps_recc_drain_gather(None, None, 0, C.byref(n), 0) == -22
    assert L.amps_recc_rccl_abort(None) == -22 and L.amps_recc_rccl_set_timeout(None, 1000) == -22
    info = capi.RcclInfo()
    info.struct_size = C.sizeof(capi.RcclInfo)
    assert L.amps_recc_rccl_info(None, C.byref(info)) == -22
    for code, text in ((-110, b"RCCL timeout"), (-107, b"aborted"), (-121, b"another rank")):       # -ETIMEDOUT, -ENOTCONN, -EREMOTEIO
        assert text in L.amps_recc_strerror(code)


@pytest.mark.parametrize("kind", ["page_response", "registration", "origination"])
def test_reply_words_equal_the_reference_restatement(kind):
    rng = np.random.default_rng(21)
    for dialed in ("5551212", "0", "*99#1234567"):
        words = synth.make_message(kind, "2125551212", esn=0x12345678, dialed=dialed)
        syms = synth.manchester(synth.burst_bits(words, rng=rng))[82:82 + 3374]
        rec = oracle.decode_bursts(syms[None, :])[0]
        a, b = capi.reply_words(rec), oracle.reply_words(rec)
        assert bytes(a) == bytes(b)
        if kind == "origination" and dialed == "0":
            assert "".join(map(str, a.focc_word2)) == oracle.focc_word2_general(int(rec["b_MIN2"]), 0, 0, 9)
    dropped = np.zeros((), oracle.BURST_DTYPE)
    r = capi.reply_words(dropped)
    assert not (r.has_focc or r.has_fvc or r.has_mutes or r.has_command)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gr_amps_amd/, include/ or bench.py's product path uses it."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "gr_amps_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "amps_oracle.h" in txt or "libamps_oracle" in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, bad
