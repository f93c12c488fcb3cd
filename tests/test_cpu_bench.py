"""CPU-side pieces of bench.py: the cpu_baseline leg (oracle timed on the host) and the profile look-up carry the fields
the bench contract names."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gr_amps_amd import synth  # noqa: E402


def test_cpu_baseline_fields():
    iq = np.stack([synth.make_channel_block(1 << 16, 1, seed=i)[0] for i in range(2)])
    cb = bench.cpu_baseline(iq, 10, 1.0)
    for k in ("value", "unit", "cores", "kind", "sample", "all_cores_value", "all_cores", "with_channel_filter"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "Msym/s" and cb["value"] > 0
    assert cb["with_channel_filter"]["value"] > 0 and cb["with_channel_filter"]["value"] < cb["value"]


def test_flop_models_and_profile_lookup():
    # the compute roofline's flop model: fold 32768 + FFT 37376 flops per frame, slicer per active bin on top
    assert bench.chz_flops_per_frame(8, "product", 0) == 1024 * 8 * 4 + 10240 * 2 + 2816 * 6
    assert bench.chz_flops_per_frame(8, "atan", 832) > bench.chz_flops_per_frame(8, "sine", 832) > bench.chz_flops_per_frame(8, "product", 832)
    assert 15.0 < bench.chz_flops_per_frame(8, "sine", 832) / (512 * 8) < 20.0          # ~18 flop per input byte
    assert bench.front_flops_per_sample("product", 10) == 3
    e = bench.profile_traffic("wideband832:sine")
    assert e is None or (e["hbm_bytes_per_launch"] >= e["algorithmic_bytes_per_launch"] == 8 << 27 and "source" in e)


def test_gpus_n_spawns_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher in the environment starts two ranks (torch.distributed.run on 127.0.0.1),
    which rendezvous, reduce and print ONE JSON line with n_gpus = 2 (CPU plumbing mode: gloo, no kernels)."""
    import subprocess
    cmd = bench.spawn_command(2, ["--gpus", "2"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["AMPS_BENCH_CPU_PLUMBING"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist", "broadcast"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0 and line["dist"] == "broadcast"


def test_committed_bench_line_has_the_contract_fields():
    pdir = os.path.join(ROOT, "profiles")
    tag = sorted(t for t in os.listdir(pdir) if os.path.exists(os.path.join(pdir, t, "bench_default.json")))[-1]
    line = json.loads(open(os.path.join(pdir, tag, "bench_default.json")).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert "workload" in line["config"] and line["vs_baseline"] is None
    if tag >= "r02":
        assert line["roofline"]["traffic"] is None and "roofline_compute" in line
        assert abs(line["roofline_compute"]["frac"] - line["roofline_compute"]["achieved"] / 157.3) < 1e-3
    if tag >= "r06":
        # the line checks its own event timing against the host clock: the dominant kernel's events + the other kernels of a step fit the step
        for part in (line, line.get("secondary") or line):
            ev = part["roofline"]["events_vs_step"]
            assert ev["consistent"] and ev["kernel_ms_plus_other_kernels"] <= 1.02 * ev["ms_per_step"], ev
        r = line["roofline"]
        assert abs(r["frac"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 8e12) < 2e-3
