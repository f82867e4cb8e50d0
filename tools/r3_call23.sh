set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_comb.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_look_ahead.py -q > gpurun_out/r3al_tests.log 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed|Error" gpurun_out/r3al_tests.log | head
python tools/comb_probe.py nocpu nobank 2>/dev/null
