#!/usr/bin/env python3
"""tools/resusage.py < hipcc -Rpass-analysis=kernel-resource-usage stderr: one line per kernel"""
import re, subprocess, sys
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    try: nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception: nm = r["name"]
    nm = nm.replace("(anonymous namespace)::", "").split("(")[0]
    print("%-34s VGPR %4s AGPR %3s scratch %5s occ %2s sgpr-spill %4s vgpr-spill %4s LDS %6s" % (nm, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
