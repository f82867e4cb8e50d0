set -o pipefail
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/r3ak_tests.log 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/r3ak_tests.log | head
./tools/microbench/adsr_par_debug 64 | grep -E "block [345]|slowest|rounds per|verify us"
python tools/shard_probe.py | tail -4
