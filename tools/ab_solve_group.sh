H="--odometry-scans 0 --polar-scans 0 --slam-scans 0 --polar-odometry-scans 0 --no-cpu-baseline --no-config2 --cpp-drive-scans 0 --replica-steps 0 --distinct-inputs 0 --no-auto-region"
for g in 0 1 0 1; do
  RANDT_SOLVE_GROUP=$g python bench.py $H --steps 2000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('group', sys.argv[1], 'value %.3f M  sustained %.3f M  host %.1f us  chip-filling solve launch %.1f us  sustained solve launches %.2f us  single batch %.1f us' % (d['value']/1e6, d['sustained']['value']/1e6, d['host_enqueue_ms_per_step']*1e3, r['avg_launch_us'], r.get('sustained',{}).get('us_per_512_launch',0), d['single_batch']['batch_latency_us']))" $g
done
