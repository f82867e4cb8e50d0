set -o pipefail
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/r3u_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3u_tests.log
python tools/c2_steps.py
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu | cut -c1-260
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu | cut -c1-260
timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu | cut -c1-260
