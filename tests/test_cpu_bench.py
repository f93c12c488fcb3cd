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


def test_profile_entries_match_the_workloads():
    for key, alg in (("wideband832", 8 << 27), ("direct832", 832 * 262144 * 8)):
        e = bench.profile_entry(key)
        assert e is not None and e["algorithmic_bytes_per_launch"] == alg
        assert bench.traffic_from_profiles(key) == e["hbm_bytes_per_launch"] >= alg
        assert 0.0 < e["valu_issue_frac"] < 1.0
    line = json.loads(open(os.path.join(ROOT, "profiles", "r01", "bench_default.json")).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert "workload" in line["config"] and line["vs_baseline"] is None
