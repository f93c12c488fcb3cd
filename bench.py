#!/usr/bin/env python
"""bench.py -- AMPS RECC receive path on MI355X: Manchester symbols demodulated AND decoded per second.

A "step" is one pass of the hot path over one resident batch of synthetic IQ:
    amps_recc_push_iq / amps_recc_push_wideband (device pointer) + amps_recc_drain (records to host).
Inputs are in HBM before the timed region starts.  N>1: one process per GPU (torchrun), each rank
owns its own band of channels (independent 30 kHz channels are the data-parallel axis; no data-path
collective), value = symbols processed by all ranks / max-over-ranks time  -> "scaling": "weak".

Prints ONE JSON line (rank 0).  See DESIGN.md section 6 for the definitions of roofline/cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SYMBOL_DIRECT = 80.0   # 8 B/sample x 10 samples/symbol, IQ read once
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec (6290 GB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="direct832", choices=["direct832", "direct1", "wideband832"])
    ap.add_argument("--samples", type=int, default=0, help="per-channel samples per step (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_batch(torch, dev, C, N, sps, seed):
    """C channels x N samples of config-1 style IQ (CPFSK seizure bursts in AWGN) resident on `dev`.
    A base set of distinct channels is synthesised on the CPU and tiled across the band."""
    from gr_amps_amd import synth
    base = min(C, 16)
    nb = max(1, N // 90000)
    iq, per_base = [], []
    for c in range(base):
        x, t = synth.make_channel_block(N, nb, seed=seed * 1000 + c, sps=sps)
        iq.append(x)
        per_base.append(len(t))
    iq = np.stack(iq)
    d = torch.from_numpy(iq).to(dev)
    reps = (C + base - 1) // base
    d = d.repeat(reps, 1)[:C].contiguous()
    expected = sum(per_base[c % base] for c in range(C))   # records per step
    return d, iq, expected


def cpu_baseline(iq_base, sps, budget_s):
    """Reference CPU chain (oracle restatement: quadrature demod -> M&M clock recovery -> slicer ->
    recc trigger search/capture -> Manchester -> BCH -> parse), single thread, on a bounded sample of
    the same workload; then all host cores with one channel per thread."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    n = iq_base.shape[1]
    t0 = time.perf_counter()
    done = 0
    k = 0
    while True:
        oracle.chain_iq200(iq_base[k % iq_base.shape[0]], channel=k)
        done += n
        k += 1
        el = time.perf_counter() - t0
        if el > budget_s * 0.5 or k >= 4 * iq_base.shape[0]:
            break
    single = done / sps / el
    cores = os.cpu_count() or 1
    reps = max(1, int(budget_s * 0.5 * single * sps / n))  # channels each worker can do in the remaining budget
    jobs = cores * reps
    t1 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda j: len(oracle.chain_iq200(iq_base[j % iq_base.shape[0]], channel=j)), range(jobs)))
    el2 = time.perf_counter() - t1
    allc = jobs * n / sps / el2
    return {
        "value": round(single / 1e6, 4), "unit": "Msym/s", "cores": 1, "kind": "port",
        "sample": "%d channel-blocks of %d samples @200 ksps through the restated reference chain "
                  "(quad demod, M&M, slicer, recc, recc_decode), 1 thread" % (k, n),
        "all_cores_value": round(allc / 1e6, 4), "all_cores": cores,
    }


def main():
    a = parse()
    import torch
    from gr_amps_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the RECC path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    sps = 10
    if a.workload == "direct1":
        C, N = 1, a.samples or (1 << 26)
    else:
        C, N = 832, a.samples or (1 << 18)
    batch, iq_base, expected = make_batch(torch, dev, C, N, sps, seed=rank + 1)
    r = capi.Recc(n_channels=C, sps=sps, max_samples=N, max_bursts=max(4096, 2 * expected), device=local, time_kernels=True)

    def step():
        r.push_iq(batch)
        return r.drain(copy=False)

    for _ in range(a.warmup):
        recs = step()
    # sanity: the decode path really ran -- every planted burst came back valid
    if a.warmup:
        assert len(recs) == expected, (len(recs), expected)
        assert recs["valid"].all()
    r.timing(reset=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    nrec = 0
    for _ in range(a.steps):
        nrec += len(step())
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    tm = r.timing()
    syms_per_step_rank = C * N / sps
    total_syms = syms_per_step_rank * a.steps * world
    value = total_syms / el
    front_ms = tm["ms_front"] / max(1, tm["launches_front"])
    traffic = None   # HBM bytes per launch from the PMC passes of the same command (profiles/rNN/traffic.json)
    for tag in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        tj = os.path.join(ROOT, "profiles", tag, "traffic.json")
        if os.path.exists(tj):
            t = json.load(open(tj))
            if t.get("algorithmic_bytes_per_launch") == C * N * 8:
                traffic = t["hbm_bytes_per_launch"]
            break
    ach = ALG_BYTES_PER_SYMBOL_DIRECT * syms_per_step_rank / (front_ms * 1e-3) / 1e9 if front_ms > 0 else 0.0
    out = {
        "metric": "AMPS RECC Manchester symbols demodulated+decoded per second (real-time channels = value/0.02)",
        "value": round(value / 1e6, 3), "unit": "Msym/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d RECC channels x %d fc32 IQ samples @200 ksps per step per GPU, channel-major, "
                               "fused demod+sync+BCH decode, records drained every step" % (a.workload, C, N),
                   "channels_per_gpu": C, "samples_per_channel": N, "samples_per_symbol": sps,
                   "realtime_channels_per_gpu": round(value / world / 20e3, 1),
                   "bursts_decoded_per_step_per_gpu": nrec // max(1, a.steps), "parallelism": "channels sharded x%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "kernel": "recc_front_kernel<10>", "kernel_ms": round(front_ms, 4),
                     "frac_of_measured_copy_ceiling_6290": round(ach / 6290.0, 4),
                     "other_kernels_ms_per_step": {k: round(tm[k] / a.steps, 4) for k in ("ms_resolve", "ms_decode", "ms_carry")}},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(iq_base[:, :min(N, 1 << 18)], sps, a.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
