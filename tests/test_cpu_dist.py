"""world_size-2 gloo tests of the multi-GPU form of the path: channels are
partitioned across ranks, there is no collective inside the data path, and the gathered records of
the ranks equal the single-process result.  Where a GPU is visible each rank runs its shard on the HIP
engine (capi.Recc on its own device: rank % device_count); on a box without GPUs (this container) the
shard is computed with the CPU oracle, and the test covers the sharding, broadcast and gather logic of
gr_amps_amd/shard.py that bench.py and a multi-GPU deployment use."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from gr_amps_amd import shard
    for n in (1, 7, 8, 832, 833, 6656):
        for w in (1, 2, 3, 4, 8):
            t = shard.shard_table(n, w)
            assert t[0][0] == 0 and t[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(t, t[1:]))
            sizes = [hi - lo for lo, hi in t]
            assert max(sizes) - min(sizes) <= 1
    assert shard.shard_table(832, 8) == [(104 * r, 104 * (r + 1)) for r in range(8)]


def test_channel_groups_partition_the_band():
    """cfg.wideband_groups: G interleaved groups of the 832-channel band from bin 96 -- disjoint, complete, near-equal, and made of
    runs of 64 / G adjacent channels (what lets a rank's FFT pass 3 keep 64 / G of its 64 lanes)"""
    from gr_amps_amd import shard
    for G in (1, 2, 4, 8):
        parts = [shard.group_channels(832, 96, G, r) for r in range(G)]
        assert sorted(c for p in parts for c in p) == list(range(832))
        assert max(map(len, parts)) - min(map(len, parts)) <= 64 // G
        for r, p in enumerate(parts):
            assert all((((96 + c) % 1024) % 64) // (64 // G) == r for c in p)
    assert [len(shard.group_channels(832, 96, 8, r)) for r in range(8)] == [104] * 8
    with pytest.raises(ValueError):
        shard.group_channels(832, 96, 3, 0)


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from gr_amps_amd import shard, synth
        C, N = 6, 50000
        if mode == "bcast":      # one shared block, produced on rank 0, broadcast to everyone
            block = torch.zeros((C, N, 2), dtype=torch.float32)
            if rank == 0:
                iq = np.stack([synth.make_channel_block(N, 1, seed=900 + c)[0] for c in range(C)])
                block.copy_(torch.from_numpy(iq.view(np.float32).reshape(C, N, 2)))
            shard.broadcast_block(block, src=0)
            iq = block.numpy().reshape(C, 2 * N).view(np.complex64)
        else:                    # every rank already holds (only) its own channels
            iq = np.stack([synth.make_channel_block(N, 1, seed=900 + c)[0] for c in range(C)])
        lo, hi = shard.shard_range(C, rank, world)
        if torch.cuda.is_available():                     # the product path: this rank's channel group on its own GPU
            from gr_amps_amd import capi
            with capi.Recc(n_channels=hi - lo, sps=10, max_samples=N, max_bursts=64, device=rank % torch.cuda.device_count()) as r:
                r.push_iq(np.ascontiguousarray(iq[lo:hi]))
                local = r.drain()
        else:
            local = oracle.fused_push_all(iq[lo:hi])      # rank-local channel numbering 0..hi-lo
        allrec = shard.gather_records(local, oracle.BURST_DTYPE, channel_offset=lo, dst=0)
        if rank == 0:
            ref = oracle.fused_push_all(np.stack([synth.make_channel_block(N, 1, seed=900 + c)[0] for c in range(C)]))
            q.put((len(allrec), allrec.tobytes() == ref.tobytes(), sorted(set(int(c) for c in allrec["channel"]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["band", "bcast"])
def test_two_rank_sharding_equals_single_process(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "band" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    n, same, chans = q.get(timeout=10)
    assert n == 6 and same and chans == [0, 1, 2, 3, 4, 5]
