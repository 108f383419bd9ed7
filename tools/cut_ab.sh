# the data-parallel step at world size 1 behind a real RCCL group of one: one cut (scene tokens) against two cuts
# (scene tokens + after Q-Former layer K, two-arena parameter storage), alternating runs of bench.py on one box
K=${1:-6}; PAIRS=${2:-3}
export SIG3D_SINGLE_RANK_PG=1
for i in $(seq 1 $PAIRS); do
  for cut in 0 $K; do
    python bench.py --force-reducer --qf-cut $cut --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d.get('comm') or {}
print('qf_cut %s: %.3f ms/step  exposed %s ms  buckets %s  bytes %s' % ('$cut', d['ms_per_step'], c.get('exposed_ms'), c.get('buckets'), c.get('bytes_per_step')))"
  done
done
