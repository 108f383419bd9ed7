# One GPU call on the head after the late CPU-only changes (the layer kernel's translation units split, load_state_dict):
# the GPU suite, smoke(), and one default bench line.  Outputs under gpurun_out/head_check/.
mkdir -p gpurun_out/head_check
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/head_check/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/head_check/smoke.txt 2>&1
python bench.py > gpurun_out/head_check/bench.json 2> gpurun_out/head_check/bench.err
tail -3 gpurun_out/head_check/gpu_tests.txt; tail -2 gpurun_out/head_check/smoke.txt; cut -c1-300 gpurun_out/head_check/bench.json
