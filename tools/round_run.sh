set -o pipefail
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r1_tests.log 2>&1 && tail -2 gpurun_out/r1_tests.log \
&& timeout -k 10 600 python bench.py > gpurun_out/r1_bench.json 2> gpurun_out/r1_bench.err && cut -c1-600 gpurun_out/r1_bench.json \
&& timeout -k 10 400 python tools/bench_suite.py cpu > gpurun_out/r1_suite.md 2>&1 \
&& bash tools/kernel_trace.sh r1_bench_trace bench.py --steps 50 --warmup 5 > /dev/null 2>&1; echo trace rc=$?; head -30 gpurun_out/r1_bench_trace.md
