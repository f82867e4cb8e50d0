set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_supersaw_segments.py tests/test_gpu_voice_bank.py tests/test_gpu_rccl_single.py tests/test_gpu_fuzz.py tests/test_gpu_full_size_properties.py tests/test_gpu_fullsize.py -q > gpurun_out/r3k_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r3k_tests.log
timeout -k 10 200 python tools/shard_probe.py supersaw
timeout -k 10 200 python tools/ss_probe.py
