set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_ladder_segmented.py tests/test_gpu_comb.py -q > gpurun_out/r3m_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3m_tests.log
python tools/comb_kernel_probe.py
bash tools/pmc_round.sh r3
