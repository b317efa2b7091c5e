#!/bin/bash
# Counter passes of the bench command, one rocprofv3 --pmc run per pass (no trace domains mixed in), reduced to
# profiles-style JSON: FP64 flops executed, VALU issue fraction, lane utilisation, HBM bytes - per kernel and per unit.
# Usage: tools/profile_counters.sh TAG [bench args]   ->  gpurun_out/counters_TAG.json (copy to profiles/counters.json)
TAG=$1; shift
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --lanes 1 --no-one-lane $@"   # the bench's own batch (33.5 M rays per step), one lane: every launch alone on the GPU
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" -d gpurun_out/pmc_${TAG}_$name -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }; }
run f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SCRATCH
python3 tools/counters_json.py gpurun_out/pmc_${TAG}_f64.log gpurun_out/pmc_${TAG}_f64 gpurun_out/pmc_${TAG}_rd gpurun_out/pmc_${TAG}_wr > gpurun_out/counters_${TAG}.json && rm -rf gpurun_out/pmc_${TAG}_*/ && head -c 1500 gpurun_out/counters_${TAG}.json
