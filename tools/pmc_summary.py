"""Sum rocprofv3 --pmc csv output per kernel and counter: [PMC_KERNEL=substring] python tools/pmc_summary.py gpurun_out/pmc_TAG_*"""
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(float); nd = defaultdict(set)
KERNEL = os.environ.get("PMC_KERNEL", "k_trace")
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if KERNEL not in k:
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]].add(r["Dispatch_Id"])
for c in sorted(tot):
    n = max(len(nd[c]), 1)
    print("%-28s total %.6g  per_dispatch %.6g  (%d dispatches)" % (c, tot[c], tot[c] / n, n))
