"""traffic.json of a profile round from the PMC passes of scripts/profile_round.sh: HBM bytes per launch of the two dominant
kernels = 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950's FETCH_SIZE counts half of wide coalesced reads, MI355X_MICROARCH.md), and the
share of VALU issue slots used.  usage: make_traffic_json.py <prof_dir> <slicer>"""
import collections
import csv
import glob
import json
import sys

d, slicer = sys.argv[1], sys.argv[2]


def means(sub):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(d + "/pmc*/pmc_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


out = []
# (the filter bank at its two decimations: D = 768 -- the library default since round 6, the bench's step is 11 * 256 * 64 * 768 samples
# there -- and D = 512 with its 2^27-sample step; the kernel names end in the decimation)
for key, sub, alg in (("wideband832", ", 768>(", 8 * 11 * 256 * 64 * 768), ("wideband832", ", 512>(", 8 << 27), ("direct832", "recc_front_kernel<10", 832 * 262144 * 8)):
    m = means(sub)
    if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
        continue
    e = {"key": "%s:%s%s" % (key, slicer, ":768" if "768" in sub else ""), "kernel_match": sub, "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
         "correction": "gfx950 FETCH_SIZE reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM): bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024",
         "hbm_bytes_per_launch": 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024, "algorithmic_bytes_per_launch": alg,
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (scripts/profile_round.sh)"}
    e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch"] / alg, 4)
    if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        e["SQ_INSTS_VALU"] = m["SQ_INSTS_VALU"]
        e["GRBM_GUI_ACTIVE"] = m["GRBM_GUI_ACTIVE"]
        e["valu_issue_frac"] = round(m["SQ_INSTS_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), 4)
        e["valu_issue_note"] = "wave64 VALU instructions per launch x 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)"
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"):
        if k in m:
            e[k] = m[k]
    out.append(e)
print(json.dumps(out, indent=1))
