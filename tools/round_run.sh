set -o pipefail
# usage (GPU box): bash tools/round_run.sh <tag>   -- GPU test suite, bench line (timed), kernel trace of the bench
tag=${1:-r2}
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_tests.log 2>&1; rc=$?; tail -5 gpurun_out/${tag}_tests.log; [ $rc -eq 0 ] || exit $rc
t0=$SECONDS
timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
echo "bench wall $((SECONDS - t0)) s" | tee gpurun_out/${tag}_bench_wall.txt
cut -c1-300 gpurun_out/${tag}_bench.json
bash tools/kernel_trace.sh ${tag}_bench_trace bench.py --steps 50 --warmup 5 --no-cpu > /dev/null 2>&1; echo trace rc=$?; head -12 gpurun_out/${tag}_bench_trace.md
