# generic same-box A/B: bash tools/ab.sh "ENV=1" "ENV=0" ...   (two rounds, 50 steps each)
run() { env "$@" python bench.py --no-cpu-baseline --no-variants --steps 50 2>/dev/null > gpurun_out/knob.json; python -c "
import json,sys; d=json.load(open('gpurun_out/knob.json')); print('%-40s %.3f  loss %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['final_loss']))" "$@"; }
for r in 1 2; do for v in "$@"; do run $v; done; done
