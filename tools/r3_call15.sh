set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_comb.py tests/test_gpu_parity.py tests/test_gpu_look_ahead.py tests/test_gpu_fuzz.py tests/test_gpu_biquad_sine.py tests/test_gpu_wav.py tests/test_gpu_suite_pes.py -q > gpurun_out/r3s_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3s_tests.log
python tools/comb_probe.py nocpu nobank
