set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_adsr_random.py tests/test_gpu_parity.py -q -x > gpurun_out/r3ab_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r3ab_tests.log | cut -c1-300
timeout -k 10 200 python tools/shard_probe.py | tail -4
