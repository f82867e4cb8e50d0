set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_biquad_sine.py tests/test_gpu_biquad_settled.py tests/test_gpu_full_size_properties.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_look_ahead.py -q > gpurun_out/r3v_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3v_tests.log
python tools/c2_kernel_probe.py
python tools/biquad_probe.py > /dev/null; python - <<'PY'
import bench, pygmu2_amd as pg
for f in (1_000_000, 33_000_000):
    r = bench.biquad_kernel_roofline(pg, f, 50)
    print("filter alone", f, round(r["avg_launch_ms"]*1e3,2), "us", r["frac"])
PY
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu | cut -c1-260
timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu | cut -c1-260
