#!/bin/bash
# tools/pmc_quick.sh NAME: one counter pass of a short one-lane bench with build/rt_NAME.so, per-kernel sums of the counters
export TMPDIR=/tmp
NAME=$1; shift
CTRS=${PMC_CTRS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SCRATCH SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"}
rm -rf gpurun_out/pmcq_$NAME
RT_LIB=$PWD/build/rt_$NAME.so timeout -s KILL 150 rocprofv3 --pmc $CTRS -d gpurun_out/pmcq_$NAME -o pmc --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --no-one-lane "$@" > gpurun_out/pmcq_$NAME.log 2>&1 || { tail -5 gpurun_out/pmcq_$NAME.log; exit 1; }
python3 - $NAME <<'PY'
import csv, glob, collections, sys
for f in glob.glob('gpurun_out/pmcq_%s/**/*counter_collection.csv' % sys.argv[1], recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40]; agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen: seen.add(r['Dispatch_Id']); n[k] += 1
    for k, v in agg.items():
        if 'k_level' in k or 'k_trace' in k: print(k, n[k], ' '.join('%s=%.4g' % (c.replace('SQ_', ''), x / max(1, n[k])) for c, x in sorted(v.items())))
PY
rm -rf gpurun_out/pmcq_$NAME
