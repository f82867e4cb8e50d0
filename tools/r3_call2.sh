set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_supersaw_segments.py tests/test_gpu_biquad_sine.py tests/test_gpu_voice_bank.py -q -x > gpurun_out/r3c_new_tests.log 2>&1; echo "new tests rc=$?"; tail -25 gpurun_out/r3c_new_tests.log
timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/r3c_tests.log 2>&1; echo "all tests rc=$?"; tail -15 gpurun_out/r3c_tests.log
timeout -k 10 200 python tools/shard_probe.py supersaw > gpurun_out/r3c_shard_ss.txt 2>&1; cat gpurun_out/r3c_shard_ss.txt
timeout -k 10 200 python tools/shard_probe.py > gpurun_out/r3c_shard_c5.txt 2>&1; cat gpurun_out/r3c_shard_c5.txt
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu > gpurun_out/r3c_c2_20.json 2>gpurun_out/r3c_c2_20.err; cut -c1-400 gpurun_out/r3c_c2_20.json
bash tools/kernel_trace.sh r3c_c2_trace bench.py --steps 200 --warmup 20 --no-cpu --no-extras > /dev/null 2>&1; echo trace rc=$?; head -12 gpurun_out/r3c_c2_trace.md; tail -5 gpurun_out/r3c_c2_trace.log
bash tools/kernel_trace.sh r3c_ss_trace tools/ss_probe.py supersaw > /dev/null 2>&1; echo "ss trace rc=$? (0 = no crash at exit under rocprofv3)"; tail -3 gpurun_out/r3c_ss_trace.log; head -8 gpurun_out/r3c_ss_trace.md
