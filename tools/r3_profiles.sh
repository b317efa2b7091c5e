#!/bin/bash
# tools/r3_profiles.sh: the measurements DESIGN.md section 5.0 quotes, one GPU call.  Outputs under gpurun_out/ (copied to profiles/ by hand).
source tools/gpu_steps.sh
export TMPDIR=/tmp
step r3p_counters 600 tools/profile_counters.sh r3
cp gpurun_out/counters_r3.json profiles/counters.json   # (on the box: the bench line below takes its executed figures from it; copy it back by hand afterwards)
step r3p_bench 400 python bench.py --steps 10 --warmup 2
step r3p_ks2 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p_ks2 -o ks --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-one-lane
step r3p_ks1 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3p_ks1 -o ks --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-one-lane --lanes 1
for w in sbc_full_mc_diffuse sbc_full_mc_blackcones arbc_camera arbc_nodice sbc_lightcollection; do step r3p_bench_$w 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-one-lane --workload $w; done
step r3p_hostin 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-one-lane --host-inputs
step r3p_sweep 300 python tools/sweep_rays.py --lo 15 --hi 26
step r3p_dropin 200 python tools/bench_dropin.py
step r3p_dropin_np 200 python tools/bench_dropin.py --numpy-host-work
