set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_supersaw_segments.py tests/test_gpu_voice_bank.py tests/test_gpu_rccl_single.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q > gpurun_out/r3j_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3j_tests.log
./tools/microbench/ss_phases 64 4 48000
python tools/ssbank_probe.py 64 | head -2
timeout -k 10 200 python tools/shard_probe.py supersaw
