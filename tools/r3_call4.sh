set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_supersaw_segments.py tests/test_gpu_voice_bank.py tests/test_gpu_rccl_single.py tests/test_gpu_fuzz.py -q > gpurun_out/r3e_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r3e_tests.log
timeout -k 10 200 python tools/shard_probe.py supersaw > gpurun_out/r3e_shard_ss.txt 2>&1; cat gpurun_out/r3e_shard_ss.txt
bash tools/kernel_trace.sh r3e_shard_trace tools/shard_probe.py supersaw > /dev/null 2>&1; echo trace rc=$?; head -10 gpurun_out/r3e_shard_trace.md
