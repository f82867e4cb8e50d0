set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_comb.py tests/test_gpu_parity.py tests/test_gpu_look_ahead.py -q > gpurun_out/r3n_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3n_tests.log
python tools/comb_kernel_probe.py
python tools/ns_rows.py nocpu
