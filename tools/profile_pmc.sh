#!/bin/bash
# PMC passes for the bench kernel, one rocprofv3 run per pass (no trace domains mixed in).  Usage: tools/profile_pmc.sh TAG [bench args]
# Output: gpurun_out/pmc_TAG_<pass>/ ; summarise with tools/pmc_summary.py
TAG=$1; shift
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $@"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d gpurun_out/pmc_${TAG}_$name -o pmc --output-format csv -- python3 bench.py $ARGS > gpurun_out/pmc_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${TAG}_$name.log; exit 1; }; }
run inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
run fetch TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
run write TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE
run misc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SCRATCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
echo done
