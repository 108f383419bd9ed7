# bench.py with the geometry chain's FPS in several forms, alternating on one box (what the DRIVER's clock sees: 20 steps)
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
  SIG3D_FPS_BLOCKS=0 python bench.py --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | line "coop            "
  SIG3D_DBG_FPS_PRIO=0 python bench.py --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | line "blocks-4        "
  SIG3D_DBG_FPS_PRIO=1 python bench.py --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | line "blocks-4 setprio"
  SIG3D_DBG_FPS_PRIO=1 SIG3D_DBG_FPS_WAVES=8 python bench.py --no-variants --no-cpu-baseline --no-ops-roofline --steps 20 --warmup 5 2>/dev/null | line "blocks-8 setprio"
done
