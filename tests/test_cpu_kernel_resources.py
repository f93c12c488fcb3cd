"""not gpu: the occupancy-critical kernels stay inside their register / LDS budgets (hipcc's own kernel-resource-usage report for
gfx950, cross-compiled here).  A change that pushes one of them over silently halves its occupancy on the MI355X: the filter
bank needs 3 waves per SIMD beside 136 KB of LDS (<= 168 VGPRs, no scratch), the streaming kernel 4 (<= 128 VGPRs)."""
import os
import shutil

import pytest

from gr_amps_amd import build


@pytest.fixture(scope="module")
def res():
    if not os.path.exists(build.hipcc()) or not shutil.which("c++filt"):
        if os.path.exists(build.RESOURCES):        # the report cached beside the library by the last build()
            import json
            with open(build.RESOURCES) as f:
                return json.load(f)
        pytest.skip("hipcc / c++filt not installed and no cached kernel_resources.json")
    return build.kernel_resources()


def _one(res, prefix):
    hits = {k: v for k, v in res.items() if k.startswith(prefix)}
    assert hits, prefix
    return hits


def test_filter_bank_kernel_budget(res):
    for name, r in _one(res, "void amps::chz12_kernel<8, ").items():
        # D = 512, spec D (<8, 3, 512>, round 5: the slicer's two alternating frame buffers): a handful of row addresses live in scratch
        # and are reloaded once per 128 frames.  D = 768 (round 6; a twelve-slot ring with twelve loads per half-step): up to twelve
        # spilled registers, all of them in the workgroup's set-up and in the EDGE half-steps at the head of a launch.  Either way no
        # steady role body touches scratch and no ring slot with a load in flight is ever spilled: tests/test_cpu_inflight_loads.py
        # scans the assembly for both.
        spill_ok = 16 if "<8, 3, 512>" in name else 24 if "<8, -1, 768>" in name else 12 if ", 768>" in name else 0   # (-1: the unfused form, edge steps only)
        assert r["vgprs"] <= 168 and r["scratch_bytes_per_lane"] <= 4 * spill_ok and r["vgpr_spill"] <= spill_ok, (name, r)
        assert r["waves_per_simd"] == 3, (name, r)                     # 12 waves per workgroup, one workgroup per CU
        assert r["lds_bytes"] <= 160 * 1024, (name, r)                 # one workgroup per CU owns the LDS (16 frame buffers)
    assert len(_one(res, "void amps::chz12_kernel<8, ")) == 10         # unfused + four slicer specs, at either decimation


def test_streaming_kernel_budget(res):
    for sps in (3, 4, 5, 6, 8, 10, 12):
        for name, r in _one(res, "void amps::recc_front_kernel<%d, 1, false, false, " % sps).items():
            assert r["vgprs"] <= 128 and r["waves_per_simd"] >= 4, (name, r)
            assert r["scratch_bytes_per_lane"] <= 64, (name, r)        # a few spilled loop invariants, none in the tile loop
            assert r["lds_bytes"] <= 40 * 1024, (name, r)              # four workgroups per CU
    # the instantiations a handle takes by default at the bench's sample rate (depth 1; spec D = the default, spec A) spill nothing
    # at all since round 4 (the rare paths' lane addresses are no longer hoisted into kernel-wide invariants)
    # (round 6: spec D tests for its debug taps once per tile instead of once per sample -- -1.5 % kernel time -- and the second copy of
    # its sample loop costs one 64-bit segment invariant its register: stored in the kernel's set-up, reloaded once per SEGMENT, never in
    # the tile loop)
    for sl, spill_ok in ((0, 0), (3, 2)):
        for name, r in _one(res, "void amps::recc_front_kernel<10, 1, false, false, %d>" % sl).items():
            assert r["vgpr_spill"] <= spill_ok and r["scratch_bytes_per_lane"] <= 8 * spill_ok, (name, r)


def test_small_kernels_fit_many_per_cu(res):
    for name, r in list(_one(res, "void amps::recc_bits_kernel<3, false>").items()) + list(_one(res, "void amps::recc_bits_kernel<2, false>").items()):
        assert r["vgprs"] <= 168 and r["lds_bytes"] <= 8192 and r["scratch_bytes_per_lane"] == 0, (name, r)   # issue-bound: 3 waves per SIMD are enough
    # six narrow instantiations: the two capture rules reading hit lists from HBM (the IQ seam, AMPS_RECC_BITS_KERNEL=separate) and, for the
    # wideband seam, with the trigger search inside (round 6: 2 / 3 samples per symbol, exact / tolerant): 23.8 KB of static LDS + 14.5 KB of
    # decode scratch per workgroup -- four of them per CU still (832 channels on 256 CUs in one round)
    hits = _one(res, "void amps::recc_resolve_kernel<256, 512, ")
    assert len(hits) == 6, sorted(hits)
    for name, r in hits.items():
        assert r["vgprs"] <= 128 and r["waves_per_simd"] >= 4 and r["scratch_bytes_per_lane"] == 0, (name, r)
        assert r["lds_bytes"] <= (12 if ", 0, false>" in name else 24) * 1024, (name, r)


def test_no_kernel_spills_into_the_hot_path_unnoticed(res):
    """every kernel of the library is listed; anything with more than 256 B of scratch per lane would be a rewrite gone wrong"""
    assert len(res) >= 40
    worst = max(res.items(), key=lambda kv: kv[1].get("scratch_bytes_per_lane", 0))
    assert worst[1].get("scratch_bytes_per_lane", 0) <= 256, worst
