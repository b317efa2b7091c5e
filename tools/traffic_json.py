"""HBM bytes per launch and kernel from two rocprofv3 --pmc passes (read requests, write requests):
python tools/traffic_json.py RD_DIR WR_DIR BENCH_LOG > traffic.json

Reads: (RDREQ - RDREQ_32B) * 64 B + RDREQ_32B * 32 B, then doubled: on gfx950 the read-request counter tallies the
128-byte requests of coalesced streaming reads at 64 B (MI355X_MICROARCH.md, HBM section; calibrated there for
16 B/lane loads, so for the 8 B/lane loads of these kernels the doubled figure is an upper bound and the raw one a
lower bound - both are kept).  Writes: WRREQ_64B * 64 B + (WRREQ - WRREQ_64B) * 32 B."""
import csv, glob, json, os, re, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); nd = defaultdict(set)
for d in sys.argv[1:3]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
cfg = {}
for line in open(sys.argv[3]):
    if line.startswith("{"):
        cfg = json.loads(line)
out = {"workload": "sbc_full_mc", "rays_per_step": cfg.get("config", {}).get("rays_per_step_per_gpu"), "kernels": {},
       "method": "TCC_EA0_RDREQ/_WRREQ over separate --pmc passes; reads doubled per the gfx950 note of MI355X_MICROARCH.md"}
for k, c in tot.items():
    n = max(len(nd[(k, "TCC_EA0_RDREQ_sum")]), 1)
    rd_raw = (c["TCC_EA0_RDREQ_sum"] - c["TCC_EA0_RDREQ_32B_sum"]) * 64 + c["TCC_EA0_RDREQ_32B_sum"] * 32
    wr = c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32
    out["kernels"][k] = {"launches": n, "read_bytes_raw_per_launch": rd_raw / n, "read_bytes_per_launch": 2 * rd_raw / n,
                         "write_bytes_per_launch": wr / n, "hbm_bytes_per_launch": (2 * rd_raw + wr) / n}
print(json.dumps(out, indent=1))
