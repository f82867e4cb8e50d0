set -o pipefail
timeout -k 10 500 python -m pytest tests -q -m gpu > gpurun_out/r3d_tests.log 2>&1; echo "all tests rc=$?"; tail -12 gpurun_out/r3d_tests.log
timeout -k 10 200 python tools/shard_probe.py supersaw > gpurun_out/r3d_shard_ss.txt 2>&1; cat gpurun_out/r3d_shard_ss.txt
timeout -k 10 100 python tools/c2_steps.py > gpurun_out/r3d_c2_steps.txt 2>&1; cat gpurun_out/r3d_c2_steps.txt
PGX_POOL_RESERVE=0 timeout -k 10 100 python tools/c2_steps.py > gpurun_out/r3d_c2_steps_noreserve.txt 2>&1; cat gpurun_out/r3d_c2_steps_noreserve.txt
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu > gpurun_out/r3d_c2_20.json 2>gpurun_out/r3d_c2_20.err; cut -c1-300 gpurun_out/r3d_c2_20.json
bash tools/kernel_trace.sh r3d_shard_trace tools/shard_probe.py supersaw > /dev/null 2>&1; echo trace rc=$?; head -14 gpurun_out/r3d_shard_trace.md
