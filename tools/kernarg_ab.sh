# A/B of HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) on the graphed step: unset / 1 / 0
for r in 1 2; do for v in unset 1 0; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  python bench.py --no-cpu-baseline --no-variants --steps 50 2>/dev/null > gpurun_out/ka_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/ka_$v.json')); print('dev_kernarg', '$v', d['ms_per_step'])"
done; done
