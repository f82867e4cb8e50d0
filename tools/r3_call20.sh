set -o pipefail
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/r3aa_tests.log 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed" gpurun_out/r3aa_tests.log | head -20
timeout -k 10 200 python tools/shard_probe.py | tail -4
