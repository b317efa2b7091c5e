#!/bin/bash
# tools/r3_collect.sh: copies what tools/r3_profiles.sh left under gpurun_out/ into profiles/ and prints the figures DESIGN.md 5.0 quotes
cp gpurun_out/counters_r3.json profiles/counters.json
grep '^{"metric"' gpurun_out/r3p_bench.log | tail -n1 > profiles/r3_bench_sbc_full_mc.json
for w in sbc_full_mc_diffuse sbc_full_mc_blackcones arbc_camera arbc_nodice sbc_lightcollection; do grep '^{"metric"' gpurun_out/r3p_bench_$w.log | tail -n1 > profiles/r3_bench_$w.json; done
grep '^{"metric"' gpurun_out/r3p_hostin.log | tail -n1 > profiles/r3_bench_sbc_full_mc_host_inputs.json
cp gpurun_out/r3p_ks2/ks_kernel_stats.csv profiles/r3_kernel_stats.csv; cp gpurun_out/r3p_ks1/ks_kernel_stats.csv profiles/r3_kernel_stats_one_lane.csv
grep '^{"workload"' gpurun_out/r3p_sweep.log > profiles/r3_sweep.json
{ grep '^{"rays"' gpurun_out/r3p_dropin.log; grep '^{"rays"' gpurun_out/r3p_dropin_np.log; } > profiles/r3_dropin.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
c = json.load(open('profiles/counters.json')); print('counters sha matches sources:', c.get('src_sha256') == bench.csrc_sha256())
d = json.load(open('profiles/r3_bench_sbc_full_mc.json')); o = d['one_lane']
print('bench %.4g inter/s %.1f ms/step basis=%s frac=%.3f alg=%.3f launch %.3f ms | one lane %.4g %.1f ms/step frac=%.3f hit %.1f phys %.1f | cpu %.4g' % (d['value'], d['ms_per_step'], d['roofline']['basis'], d['roofline']['frac'], d['roofline']['algorithmic_frac'], d['roofline']['avg_launch_ms'], o['value'], o['ms_per_step'], o['roofline']['frac'], o['by_kernel']['k_level_hit']['ms_per_step'], o['by_kernel']['k_level_phys']['ms_per_step'], d['cpu_baseline']['value']))
for w in ['sbc_full_mc_diffuse', 'sbc_full_mc_blackcones', 'arbc_camera', 'arbc_nodice', 'sbc_lightcollection', 'sbc_full_mc_host_inputs']:
    print(w, '%.3g' % json.load(open('profiles/r3_bench_%s.json' % w))['value'])
s = json.load(open('profiles/r3_sweep.json')); print(' '.join('%d:%.3g' % (x['rays'], x['interactions_per_s']) for x in s['sizes']))
for l in open('profiles/r3_dropin.json'): x = json.loads(l); print(x['host_work'], '%.3g rec/s' % x['records_per_s'])
for f in ('profiles/r3_kernel_stats.csv', 'profiles/r3_kernel_stats_one_lane.csv'):
    for l in list(open(f))[1:4]: p = l.split('",'); print(f[-18:], p[0][-40:], p[1].split(',')[2])
k = c['kernels']['k_level_hit']; print('hit valu/64 %.0f salu/64 %.0f flop/unit %.0f lane util %.3f issue %.3f' % (k['valu_insts_per_64_units'], k['salu_insts_per_64_units'], k['fp64_flop_per_unit'], k['lane_utilisation'], k['valu_issue_frac']))
PY
