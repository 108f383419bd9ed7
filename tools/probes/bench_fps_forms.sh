# bench.py with the geometry chain's FPS as the cooperative kernel / the block-list kernel (4 waves per scene beside the
# step), alternating on one box (what the DRIVER's clock sees: 20 steps).  The s_setprio arms DESIGN.md section 4j mentions
# (blocks-4 / blocks-8 with raised wave priority, no effect) ran on a probe build whose switches were not kept.
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  SIG3D_FPS_BLOCKS=0 python bench.py --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | line "coop    "
  SIG3D_FPS_BLOCKS=1 python bench.py --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | line "blocks-4"
done
