"""-m gpu: bench.py's N > 1 code on the one GPU a box has.  Two ranks are launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on device 0 with the collectives over gloo
(AMPS_BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device; everything else -- rendezvous, barrier, max-over-ranks timing, the
one-band split by channel groups with rank 0's block broadcast every step, the HIP kernels of both ranks -- is the code an
8-GPU node runs).  Every rank checks its own records against what was planted in its channels (an assert inside bench.py)."""
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


def _run(extra):
    env = dict(os.environ, AMPS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--samples", str(1 << 24), "--prewarm-ms", "20", "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                    # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_whole_bands(gpu):
    """--dist bands: every rank its own 832-channel band, no collective in the data path; value = both bands"""
    d = _run([])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dist"] == "bands"
    c = d["config"]
    assert c["channels_per_gpu"] == 832 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * c["checked"]["planted"] > 0
    assert abs(d["value"] - 2 * 832 * ((1 << 24) / 1536.0) * 3 / (d["ms_per_step"] * 3e-3) * 1e-6) < 1e-3 * d["value"]
    # the same invocation also measures what BASELINE configs[4] names: one band, rank 0's block broadcast inside the timed region
    s = d["secondary"]
    assert s["scaling"] == "strong" and s["collective"]["nranks"] == 2 and s["collective"]["bytes_per_step"] == 8 << 24
    assert len(s["kernel_ms_per_rank"]) == 2 and all(k > 0 for k in s["kernel_ms_per_rank"])
    assert s["config"]["channels_per_gpu"] == 416 and s["config"]["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * s["config"]["checked"]["planted"] > 0
    assert abs(s["value"] - 832 * ((1 << 24) / 1536.0) * s["steps"] / (s["ms_per_step"] * s["steps"] * 1e-3) * 1e-6) < 1e-3 * s["value"]


def test_two_ranks_one_band_by_channel_groups(gpu):
    """--dist broadcast: ONE band, rank 0's block broadcast every step, each rank decodes its interleaved channel group
    (cfg.wideband_groups = 2); value = the one band"""
    d = _run(["--dist", "broadcast"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dist"] == "broadcast"
    c = d["config"]
    assert c["channels_per_gpu"] == 416 and "channel groups" in c["parallelism"]
    assert c["checked"]["planted"] > 50 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * c["checked"]["planted"]
    assert abs(d["value"] - 832 * ((1 << 24) / 1536.0) * 3 / (d["ms_per_step"] * 3e-3) * 1e-6) < 1e-3 * d["value"]


def test_one_rank_broadcast_inside_the_c_abi(gpu):
    """--dist broadcast_abi with a world of one rank (AMPS_BENCH_FORCE_DIST=1: real RCCL, which refuses two ranks on one device): the
    communicator id over torch.distributed's control plane, ncclCommInitRank + ncclBroadcast issued by the library, the records checked"""
    env = dict(os.environ, AMPS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--samples", str(1 << 24),
           "--prewarm-ms", "20", "--no-cpu-baseline", "--no-other-specs", "--secondary", "none", "--dist", "broadcast_abi"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dist"] == "broadcast_abi" and d["n_gpus"] == 1
    c = d["config"]
    assert c["channels_per_gpu"] == 832 and c["checked"]["decoded_with_transmitted_MIN"] >= 0.97 * c["checked"]["planted"] > 0
    # one more step drained through amps_recc_drain_gather (the ranks' lists merged at rank 0 by RCCL)
    assert c["records_gathered_at_rank0_in_one_step"] >= c["checked"]["decoded_with_transmitted_MIN"]
