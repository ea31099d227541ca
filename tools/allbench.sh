# every workload of DESIGN.md §5 on one GPU:   bash tools/allbench.sh | tee gpurun_out/allbench.txt
for a in "--workload cfg1 --steps 64 --warmup 8" "--workload cfg2" "--workload cfg2 --steps 20 --warmup 5" "--workload cfg4 --warmup 8" "--workload cfg3 --steps 4 --warmup 1" "--workload cfg5 --steps 4 --warmup 1" "--workload shipped --steps 8 --images-per-launch 1" "--workload shipped --steps 8 --images-per-launch 8"; do
  echo "== $a"; python bench.py $a --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; s=d.get('single_image')
print('value %.0f  ms/step %.3f  kernel_us %.1f  bound %s frac %.3f  algorithmic_hbm_frac %.3f  t_err_mm %.1f  r_err %.3f  poses/launch %s  passes %d  kernel_share_of_step %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['bound'], r['frac'], r['algorithmic_hbm']['frac'], d['median_t_err_m']*1e3, d['median_r_err_deg'], d['config'].get('poses_per_launch'), d['passes'], d['checks']['kernel_share_of_step']))
if s: print('   single image per launch chain: value %.0f  ms/step %.3f  kernel_us %.1f  valu_frac %s  algorithmic_hbm_frac %.3f' % (s['value'], s['ms_per_step'], s['avg_launch_ms']*1e3, s['valu_frac'], s['algorithmic_hbm_frac']))
print('   traffic', r['traffic'], 'valu', r['valu'] and (r['valu']['instr_per_point_pose'], r['valu']['busy_frac_profiled']), r['kernel'])"
done
