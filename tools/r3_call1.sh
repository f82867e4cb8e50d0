set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_comb.py tests/test_gpu_biquad_sine.py -q -x > gpurun_out/r3b_new_tests.log 2>&1; echo "new tests rc=$?"; tail -25 gpurun_out/r3b_new_tests.log
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/r3b_tests.log 2>&1; echo "all tests rc=$?"; tail -15 gpurun_out/r3b_tests.log
timeout -k 10 300 python tools/comb_probe.py > gpurun_out/r3b_comb.json 2> gpurun_out/r3b_comb.err; echo "comb rc=$?"; cat gpurun_out/r3b_comb.json
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu > gpurun_out/r3b_c2_20.json 2>gpurun_out/r3b_c2_20.err; cat gpurun_out/r3b_c2_20.json
timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu > gpurun_out/r3b_c2_200.json 2>gpurun_out/r3b_c2_200.err; cat gpurun_out/r3b_c2_200.json
