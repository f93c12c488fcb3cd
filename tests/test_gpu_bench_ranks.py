"""-m gpu: bench.py's N > 1 code on the one GPU a box has.  2, 4 and 8 ranks are launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), all on device 0 with torch.distributed's collectives over
gloo (AMPS_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device) and the LIBRARY's collectives (the *_abi modes) over the loop-back
stand-in of tests/loopccl (AMPS_RECC_RCCL_LIB); everything else -- rendezvous, barrier, max-over-ranks timing, the one-band split by
channel groups G = 2 / 4 / 8 with rank 0's block distributed every step, the HIP kernels of every rank, the self-describing record -- is
the code an 8-GPU node runs (BASELINE configs[4], the one configuration no round could run on hardware).  Every rank checks its own
records against what was planted in its channels (an assert inside bench.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# block sizes of these runs: whole groups of 64 frames at either decimation of the filter bank (64 x 1536 samples), so that every step
# consumes and decodes the same -- 2^24 and 2^23 are no whole number of 768-sample frames
NS24, NS23 = 170 * 64 * 1536, 85 * 64 * 1536


def _run(extra, n=2, samples=NS24, timeout=300):
    import loopccl
    env = dict(os.environ, AMPS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", AMPS_RECC_RCCL_LIB=loopccl.build(), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
           "--samples", str(samples), "--prewarm-ms", "20", "--no-cpu-baseline", "--no-power-sample"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                    # rank 0 prints ONE line
    return json.loads(lines[0])


def _check_ranks(ranks, n, with_comm):
    """the per-rank entries of a record: one per rank, each with the device the library ran on and its own timings"""
    assert [e["rank"] for e in ranks] == list(range(n))
    for e in ranks:
        assert len(e["device_uuid"]) == 32 and e["ms_per_step"] > 0 and e["kernel_ms"] > 0 and e["pci"].count(":") == 2
        if with_comm:                                           # the communicator the library owns, as the collective library reports it
            assert e["rccl_nranks"] == n and e["rccl_rank"] == e["rank"] and "loopccl" in e["rccl_library"]
        else:
            assert e["rccl_nranks"] is None


def _one_band_value(s, n_samples):
    return 832 * (n_samples / 1536.0) * s["steps"] / (s["ms_per_step"] * s["steps"] * 1e-3) * 1e-6


def test_two_ranks_whole_bands(gpu):
    """--dist bands: every rank its own 832-channel band, no collective in the data path; value = both bands"""
    d = _run([])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dist"] == "bands"
    c = d["config"]
    assert c["channels_per_gpu"] == 832 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * c["checked"]["planted"] > 0
    assert abs(d["value"] - 2 * 832 * (NS24 / 1536.0) * 3 / (d["ms_per_step"] * 3e-3) * 1e-6) < 1e-3 * d["value"]
    # the same invocation also measures what BASELINE configs[4] names: one band, rank 0's block broadcast inside the timed region
    s = d["secondary"]
    assert s["scaling"] == "strong" and s["collective"]["nranks"] == 2 and s["collective"]["bytes_per_step"] == 8 * NS24
    assert len(s["kernel_ms_per_rank"]) == 2 and all(k > 0 for k in s["kernel_ms_per_rank"])
    assert s["config"]["channels_per_gpu"] == 416 and s["config"]["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * s["config"]["checked"]["planted"] > 0
    assert abs(s["value"] - 832 * (NS24 / 1536.0) * s["steps"] / (s["ms_per_step"] * s["steps"] * 1e-3) * 1e-6) < 1e-3 * s["value"]
    _check_ranks(d["ranks"], 2, False)
    assert d["distinct_devices"] == 1 and d["process_group"] == {"backend": "gloo", "world_size": 2}     # (one GPU here: an 8-GPU record says 8)


@pytest.mark.parametrize("n", [4, 8])
def test_four_and_eight_ranks_under_the_launcher(gpu, n):
    """G = 4 and G = 8 under the launcher, not only in test_gpu_channel_groups.py.  ONE invocation, as the driver's
    scaling run makes it: the headline (a band per rank) + the one band broadcast through torch.distributed + the one band by scatter +
    all-gather issued inside the C ABI -- one JSON line with n per-rank entries."""
    ns = NS23
    d = _run([], n, ns)
    assert d["n_gpus"] == n and d["scaling"] == "weak" and d["dist"] == "bands" and d["config"]["channels_per_gpu"] == 832
    assert abs(d["value"] - n * 832 * (ns / 1536.0) * 3 / (d["ms_per_step"] * 3e-3) * 1e-6) < 1e-3 * d["value"]
    _check_ranks(d["ranks"], n, False)
    assert d["process_group"]["world_size"] == n
    for key, with_comm in (("secondary", False), ("secondary_abi", True)):
        s = d[key]
        assert "error" not in s, s
        assert s["scaling"] == "strong" and s["collective"]["nranks"] == n and s["collective"]["bytes_per_step"] == 8 * ns
        assert s["collective"]["gbps"] > 0 and s["collective"]["timed"] >= 1
        assert len(s["kernel_ms_per_rank"]) == n and all(k > 0 for k in s["kernel_ms_per_rank"])
        c = s["config"]
        assert c["channels_per_gpu"] == 832 // n and "%d interleaved channel groups" % n in c["parallelism"]
        assert c["checked"]["planted"] >= 1 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.9 * c["checked"]["planted"]
        assert abs(s["value"] - _one_band_value(s, ns)) < 1e-3 * s["value"]
        _check_ranks(s["ranks"], n, with_comm)
    assert d["secondary_abi"]["collective"]["op"] == "scatter_allgather" and "libamps_recc" in d["secondary_abi"]["collective"]["issued_by"]


@pytest.mark.parametrize("n,mode", [(8, "scatter_allgather_abi"), (2, "broadcast_abi")])
def test_one_band_modes_as_the_headline(gpu, n, mode):
    ns = NS23
    d = _run(["--dist", mode], n, ns)
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["dist"] == mode
    c = d["config"]
    assert c["channels_per_gpu"] == 832 // n and "channel groups" in c["parallelism"]
    assert c["checked"]["planted"] >= 1 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.9 * c["checked"]["planted"]
    assert abs(d["value"] - 832 * (ns / 1536.0) * 3 / (d["ms_per_step"] * 3e-3) * 1e-6) < 1e-3 * d["value"]
    _check_ranks(d["ranks"], n, mode.endswith("_abi"))
    assert d["collective"]["gbps"] > 0 and d["collective"]["bytes_per_step"] == 8 * ns
    if mode.endswith("_abi"):                                   # one more step drained through amps_recc_drain_gather: the whole band's bursts at rank 0
        assert c["records_gathered_at_rank0_in_one_step"] >= 0.9 * 416


def test_two_ranks_one_band_by_channel_groups(gpu):
    """--dist broadcast: ONE band, rank 0's block broadcast every step, each rank decodes its interleaved channel group
    (cfg.wideband_groups = 2); value = the one band"""
    d = _run(["--dist", "broadcast"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dist"] == "broadcast"
    c = d["config"]
    assert c["channels_per_gpu"] == 416 and "channel groups" in c["parallelism"]
    assert c["checked"]["planted"] > 50 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * c["checked"]["planted"]
    assert abs(d["value"] - 832 * (NS24 / 1536.0) * 3 / (d["ms_per_step"] * 3e-3) * 1e-6) < 1e-3 * d["value"]


@pytest.mark.parametrize("mode,in_place", [("broadcast_abi", False), ("scatter_allgather_abi", False), ("broadcast_abi", True)])
def test_one_rank_distribution_inside_the_c_abi(gpu, mode, in_place):
    """--dist broadcast_abi / scatter_allgather_abi with a world of one rank (AMPS_BENCH_FORCE_DIST=1: real RCCL, which refuses two ranks on one device): the
    communicator id over torch.distributed's control plane, ncclCommInitRank + ncclBroadcast issued by the library, the records checked"""
    # (AMPS_RECC_RCCL_FORCE_COLLECTIVE: a one-rank communicator uses a device block in place since round 6 -- here the data collective
    # itself is the point)
    env = dict(os.environ, AMPS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", AMPS_RECC_RCCL_FORCE_COLLECTIVE="0" if in_place else "1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--samples", str(NS24),
           "--prewarm-ms", "20", "--no-cpu-baseline", "--no-other-specs", "--secondary", "none", "--dist", mode]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dist"] == mode and d["n_gpus"] == 1
    assert d["ranks"][0]["rccl_nranks"] == 1 and "rccl" in d["ranks"][0]["rccl_library"]       # the REAL librccl
    assert (d["collective"]["timed"] == 0) if in_place else (d["collective"]["gbps"] > 0)          # in place: the header exchange alone, no data collective
    c = d["config"]
    assert c["channels_per_gpu"] == 832 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * c["checked"]["planted"] > 0
    # one more step drained through amps_recc_drain_gather (the ranks' lists merged at rank 0 by RCCL)
    assert c["records_gathered_at_rank0_in_one_step"] >= c["checked"]["decoded_with_transmitted_MIN"]
