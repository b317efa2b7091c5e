"""rocprofv3 --pmc csv output of tools/profile_counters.sh -> JSON per kernel (k_level_hit, k_level_phys<..>, k_trace):
python tools/counters_json.py BENCH_LOG PMC_DIR...

Per dispatch averages, and per unit for the nearest-hit kernel (unit = one live ray of one step: units per launch from
the bench line of the same run).  Conventions: SQ_* instruction counters count wave-instructions, so FP64 flops =
(ADD + MUL + TRANS + 2 FMA) x 64 counts all 64 lanes (lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)
says how many were on); SQ_ACTIVE_* / SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE is
summed over the 8 XCDs.  HBM bytes: reads (RDREQ - RDREQ_32B) * 64 + RDREQ_32B * 32 raw, doubled for the figure
compared with byte counts (gfx950 tallies the 128-byte requests of streaming reads at 64 B: MI355X_MICROARCH.md, HBM);
writes WRREQ_64B * 64 + (WRREQ - WRREQ_64B) * 32."""
import csv, glob, json, os, re, sys, hashlib
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tot = defaultdict(lambda: defaultdict(float)); nd = defaultdict(lambda: defaultdict(set))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1).replace(" ", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k][r["Counter_Name"]].add(r["Dispatch_Id"])
cfg = {}
for line in open(sys.argv[1]):
    if line.startswith("{"):
        cfg = json.loads(line)
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(ROOT, "raytracer_amd", "csrc", "*"))):
    h.update(open(f, "rb").read())
out = {"workload": cfg.get("config", {}).get("workload", "").split(":")[0], "rays_per_step": cfg.get("config", {}).get("rays_per_step_per_gpu"), "src_sha256": h.hexdigest(),
       "bench_line": {k: cfg.get(k) for k in ("value", "ms_per_step")}, "kernels": {}}
dom = cfg.get("roofline", {})
for k, c in tot.items():
    per = {n: v / max(len(nd[k][n]), 1) for n, v in c.items()}
    g = lambda n: per.get(n, 0.0)  # noqa: E731
    flop = (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64") + 2 * g("SQ_INSTS_VALU_FMA_F64")) * 64
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    rd_raw = (g("TCC_EA0_RDREQ_sum") - g("TCC_EA0_RDREQ_32B_sum")) * 64 + g("TCC_EA0_RDREQ_32B_sum") * 32
    wr = g("TCC_EA0_WRREQ_64B_sum") * 64 + (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum")) * 32
    e = {"dispatches": max(len(v) for v in nd[k].values()), "gpu_cycles_per_dispatch": cyc, "fp64_flop_per_dispatch": flop,
         "valu_issue_frac": 4 * g("SQ_ACTIVE_INST_VALU") / (cyc * 1024) if cyc else None,
         "lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else None,
         "wave_cycles_waiting_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
         "wave_cycles_issue_stall_frac": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
         "hbm_read_bytes_raw_per_dispatch": rd_raw, "hbm_read_bytes_per_dispatch": 2 * rd_raw, "hbm_write_bytes_per_dispatch": wr,
         "hbm_bytes_per_dispatch": 2 * rd_raw + wr, "per_dispatch": per}
    if k.startswith("k_level_hit") and dom.get("kernel") == "k_level_hit" and dom.get("units_per_launch"):
        u = dom["units_per_launch"]
        e.update(units_per_dispatch=u, fp64_flop_per_unit=flop / u, hbm_bytes_per_unit=(2 * rd_raw + wr) / u, hbm_bytes_raw_per_unit=(rd_raw + wr) / u,
                 valu_insts_per_64_units=64 * g("SQ_INSTS_VALU") / u, salu_insts_per_64_units=64 * g("SQ_INSTS_SALU") / u)
    out["kernels"]["k_level_hit" if k.startswith("k_level_hit") else k] = e
print(json.dumps(out, indent=1))
