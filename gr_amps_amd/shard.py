"""Channel sharding across the GPUs of one node.

30 kHz RECC channels are independent after channelisation (no cross-channel state anywhere in
lib/recc_impl.h:31-43), so the multi-GPU form of the path is a partition of channels: one process
per GPU, each owning a contiguous channel group, no collective inside the data path.  The only
exchanges are (a) optionally distributing one shared input block from rank 0 (broadcast over
RCCL/xGMI when every GPU must see the same wideband stream) and (b) collecting the small burst
records.  Works with any torch.distributed backend: nccl (= RCCL) on the GPUs, gloo in the CPU tests.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous balanced split: the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_table(n_items, world):
    return [shard_range(n_items, r, world) for r in range(world)]


def group_channels(n_channels, first_bin, groups, group, m=1024):
    """Channels (band numbering, ascending) of interleaved channel group `group` of `groups` -- cfg.wideband_groups of the C ABI: the
    channels whose FFT bin k = (first_bin + c) mod m has (k mod 64) in [group * 64/groups, (group + 1) * 64/groups).  The one-band
    multi-GPU split: every rank folds the whole wideband stream, but runs the last FFT pass, the slicer and everything behind them
    for its own group only (include/amps_recc.h; bench.py --dist broadcast / --groups)."""
    if groups not in (1, 2, 4, 8) or not 0 <= group < groups:
        raise ValueError("groups must be 1, 2, 4 or 8 and 0 <= group < groups")
    w = 64 // groups
    return [c for c in range(n_channels) if (((first_bin + c) % m) % 64) // w == group]


def broadcast_block(block, src=0, group=None):
    """Broadcast a torch tensor (the shared IQ block) from `src` to every rank, in place."""
    import torch.distributed as dist
    dist.broadcast(block, src=src, group=group)
    return block


def gather_records(records, dtype, channel_offset=0, dst=0, group=None):
    """Collect structured burst records from every rank on `dst` (returns None elsewhere).
    `channel_offset` maps the rank-local channel index to the band-wide one."""
    import torch
    import torch.distributed as dist
    rec = np.ascontiguousarray(records).copy()
    if rec.size:
        rec["channel"] += np.uint32(channel_offset)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    n = torch.tensor([rec.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts + [1])
    buf = torch.zeros(nmax * dtype.itemsize, dtype=torch.uint8, device=dev)
    if rec.size:
        buf[:rec.size * dtype.itemsize] = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    if rank != dst:
        return None
    parts = [b.cpu().numpy()[:c * dtype.itemsize].view(dtype) for b, c in zip(bufs, counts)]
    out = np.concatenate(parts) if parts else np.zeros(0, dtype)
    return out[np.lexsort((out["position"], out["channel"]))]
