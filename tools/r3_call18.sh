for v in "" "$PWD/pygmu2_amd/libpygmu_hip_2b.so" "" "$PWD/pygmu2_amd/libpygmu_hip_2b.so"; do
  echo "== lib: ${v:-default (one barrier)}"
  export PGX_LIB_PATH=$v; [ -z "$v" ] && unset PGX_LIB_PATH
  python tools/c2_kernel_probe.py | tail -2
  python - <<'PY'
import bench, pygmu2_amd as pg
r = bench.biquad_kernel_roofline(pg, 33_000_000, 50)
print("filter alone 33M", round(r["avg_launch_ms"]*1e3,2), "us")
PY
  for i in 1 2 3; do timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu | cut -c90-130; done
done
