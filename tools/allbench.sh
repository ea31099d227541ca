for a in "--workload cfg1 --images-per-launch 1 --steps 64 --warmup 8" "--workload cfg1 --images-per-launch 8 --steps 64 --warmup 8" "--workload cfg1 --images-per-launch 64 --steps 128 --warmup 64" "--workload cfg2 --images-per-launch 1" "--workload cfg2" "--workload cfg4 --steps 64 --warmup 8" "--workload cfg3 --steps 4 --warmup 1" "--workload cfg5 --steps 4 --warmup 1"; do
  echo "== $a"; python bench.py $a --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('value %.0f  ms/step %.3f  kernel_us %.1f  frac %.3f  t_err_mm %.1f  r_err %.2f  ipl %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms']*1e3, r['frac'], d['median_t_err_m']*1e3, d['median_r_err_deg'], d['config'].get('images_per_launch')))"
done
