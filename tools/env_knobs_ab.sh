# which HIP / HSA environment knobs move the graphed step?  (unknown names are ignored by the runtime)
run() { env "$@" python bench.py --no-cpu-baseline --no-variants --steps 50 2>/dev/null > gpurun_out/knob.json; python -c "
import json,sys; d=json.load(open('gpurun_out/knob.json')); print('%-45s %.3f  timeouts %s' % (' '.join(sys.argv[1:]), d['ms_per_step'], d['fps_timeouts']))" "$@"; }
for r in 1 2; do
run X=1
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=3
done
