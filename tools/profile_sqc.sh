#!/bin/bash
# scalar data cache / instruction cache hit rates of the bench kernel (two short PMC passes).  Usage: tools/profile_sqc.sh TAG [bench args]
TAG=$1; shift
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --rays 8388608 $@"
run() { name=$1; shift; timeout -s KILL 90 rocprofv3 --pmc "$@" -d gpurun_out/pmc_${TAG}_$name -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }; }
run dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run tc SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL
echo done
