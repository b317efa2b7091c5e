#!/bin/bash
# HBM traffic of the bench kernels from the L2 memory-side request counters, one rocprofv3 --pmc run per pass
# (MI355X_MICROARCH.md, HBM / rocprofv3 sections).  Usage: tools/profile_traffic.sh TAG [bench args]
# Writes gpurun_out/pmc_TAG_{rd,wr}/ and gpurun_out/traffic_TAG.json (copy to profiles/traffic.json to have bench.py report it).
TAG=$1; shift
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $@"
run() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" -d gpurun_out/pmc_${TAG}_$name -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }; }
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python3 tools/traffic_json.py gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr gpurun_out/pmc_${TAG}_rd.log > gpurun_out/traffic_${TAG}.json && cat gpurun_out/traffic_${TAG}.json
