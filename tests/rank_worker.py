"""One rank of a multi-rank test of the library's collective entry points (tests/test_gpu_rccl_ranks.py starts N of these, all on the
box's one GPU, with AMPS_RECC_RCCL_LIB = the loop-back stand-in).  Everything goes through the C ABI (gr_amps_amd.capi); the result
of the rank is a JSON file (+ the gathered records on the root).  No torch here: host blocks in, host records out."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D, FIRST, CW = 512, 96, 832


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--nranks", type=int, required=True)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--scenario", required=True)
    ap.add_argument("--mode", default="broadcast")
    ap.add_argument("--root", type=int, default=0)
    a = ap.parse_args()
    from gr_amps_amd import capi
    out = {"rank": a.rank, "scenario": a.scenario, "events": []}

    def code_of(fn, *args, **kw):
        try:
            return 0, fn(*args, **kw)
        except capi.AmpsError as e:
            return e.code, None

    def finish():
        with open(os.path.join(a.dir, "rank%d.json" % a.rank), "w") as f:
            json.dump(out, f)

    # the communicator id: rank 0 makes it, a file carries it (the control plane is the application's)
    idf = os.path.join(a.dir, "id.bin")
    if a.rank == 0:
        uid = capi.Recc.rccl_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            if time.time() - t0 > 120:
                raise SystemExit("no id file")
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    x = np.load(os.path.join(a.dir, "x.npy"), mmap_mode="r") if a.rank == a.root else None
    n = int(np.load(os.path.join(a.dir, "n.npy")))
    G = a.nranks if a.nranks in (2, 4, 8) else 0
    group = a.rank
    if a.scenario == "bad_groups" and a.rank == a.nranks - 1:
        group = 0                                             # this rank's handle was built for somebody else's group
    wb = {"channels": 1024, "decim": 512, "taps_per_branch": 8, "first_channel": FIRST}
    if G:
        wb.update(groups=G, group=group)
    cap_frames = n // D + 72
    if a.scenario == "small_rank" and a.rank == a.nranks - 1:
        cap_frames = (n // 3) // D + 72                       # this rank's handle takes a third of what the others take
    with capi.Recc(n_channels=CW, sps=3, max_samples=cap_frames, max_bursts=64 if a.rank else 96, wideband=wb, time_kernels=True) as r:
        rc, _ = code_of(r.rccl_init, uid, a.nranks, a.rank)
        out["init"] = rc
        if rc:
            out["second_init_after_failure"] = None
            finish()
            return
        out["info"] = r.rccl_info()
        if a.scenario in ("dist", "small_rank"):
            cuts = [0, 2000000, 5000064, n] if a.scenario == "dist" else [0, n // 3, 2 * (n // 3), n]
            if a.scenario == "small_rank":
                # the root offers the whole stream at once: larger than the smallest rank's capacity -> -E2BIG at the root, -EREMOTEIO
                # elsewhere, no data collective; the communicator stays usable and the thirds go through
                rc, _ = code_of(r.push_wideband_dist, np.asarray(x[:n]) if x is not None else None, None, a.root, a.mode)
                out["events"].append(["oversize", rc])
            pushed = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                blk = np.ascontiguousarray(x[lo:hi]) if x is not None else None
                pushed.append(r.push_wideband_dist(blk, None, a.root, a.mode))
            tail = np.zeros(64 * D, np.complex64) if x is not None else None
            pushed.append(r.push_wideband_dist(tail, None, a.root, a.mode))
            # a block so short that the scatter leaves most ranks with an EMPTY chunk (they post no receive at all)
            crumb = np.zeros(100, np.complex64) if x is not None else None
            pushed.append(r.push_wideband_dist(crumb, None, a.root, a.mode))
            out["pushed"] = pushed
            recs = r.drain_gather(root=a.root, cap=4096)
            out["gathered"] = int(len(recs))
            if a.rank == a.root:
                np.save(os.path.join(a.dir, "gathered.npy"), recs)
            out["second_gather"] = int(len(r.drain_gather(root=a.root)))
            out["info_after"] = r.rccl_info()
        elif a.scenario == "root_error":
            blk = np.ascontiguousarray(x[:1000000]) if x is not None else None
            out["events"].append(["good", r.push_wideband_dist(blk, None, a.root, a.mode)])
            rc, _ = code_of(r.push_wideband_dist, blk, 0, a.root, a.mode)            # the root offers a block of no samples: its error, everybody's verdict
            out["events"].append(["root_without_samples", rc])
            rc, _ = code_of(r.push_wideband_dist, None, None, a.root, a.mode)        # the root has nothing more: END OF STREAM, the same answer on every rank,
            out["events"].append(["end_of_stream", rc])                               # no data collective, and the communicator carries on
            out["events"].append(["good_again", r.push_wideband_dist(blk, None, a.root, a.mode)])
            mode = a.mode if a.rank != a.nranks - 1 else ("scatter_allgather" if a.mode == "broadcast" else "broadcast")
            rc, _ = code_of(r.push_wideband_dist, blk, None, a.root, mode)            # the ranks disagree on the mode
            out["events"].append(["mode_mismatch", rc])
            out["events"].append(["good_after_mismatch", r.push_wideband_dist(blk, None, a.root, a.mode)])
            out["gathered"] = int(len(r.drain_gather(root=a.root)))
        elif a.scenario == "peer_leaves":
            blk = np.ascontiguousarray(x[:1000000]) if x is not None else None
            out["events"].append(["good", r.push_wideband_dist(blk, None, a.root, a.mode)])
            if a.rank == a.nranks - 1:
                r.rccl_abort()                                # this rank leaves the game (its flow graph stopped)
                rc, _ = code_of(r.push_wideband_dist, blk, None, a.root, a.mode)
                out["events"].append(["after_own_abort", rc])
            else:
                r.rccl_set_timeout(1500)
                t0 = time.time()
                rc, _ = code_of(r.push_wideband_dist, blk, None, a.root, a.mode)
                out["events"].append(["peer_gone", rc, round(time.time() - t0, 2)])
                rc, _ = code_of(r.push_wideband_dist, blk, None, a.root, a.mode)
                out["events"].append(["after_timeout", rc])
                rc, _ = code_of(r.drain_gather, a.root)
                out["events"].append(["gather_after_timeout", rc])
            # what the handle found behind the aborted collective is void: the seams say so until the handle is reset, then it works on
            rc, _ = code_of(r.drain)
            out["events"].append(["drain_before_reset", rc])
            rc, _ = code_of(r.push_wideband, np.zeros(64 * D, np.complex64))
            out["events"].append(["push_before_reset", rc])
            r.reset()
            r.drain()
            r.push_wideband(np.zeros(64 * D, np.complex64))
            out["plain_drain_after"] = int(len(r.drain()))
            out["info_after"] = r.rccl_info()
        elif a.scenario == "bad_groups":
            pass
        else:
            raise SystemExit("unknown scenario " + a.scenario)
    finish()


if __name__ == "__main__":
    main()
