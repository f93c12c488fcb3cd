import csv, sys, glob, collections
# usage: pmc_summary.py <prof_dir> [kernel substring]
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "recc_front"
for f in sorted(glob.glob(d + "/pmc*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-24s n=%3d mean=%.4g" % (k, len(v), sum(v) / len(v)))
