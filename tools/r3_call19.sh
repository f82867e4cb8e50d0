set -o pipefail
timeout -k 10 400 python -m pytest tests/test_gpu_ladder_segmented.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_voice_bank.py tests/test_gpu_look_ahead.py -q > gpurun_out/r3z_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3z_tests.log
python - <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import bench_suite as B
from oracle.golden_cases import S
sine = lambda f, a: S("SinePE", frequency=f, amplitude=a)
spec = S("LadderPE", source=S("BlitSawPE", frequency=110.0), frequency=S("MixPE", inputs=[S("ConstantPE", value=1200.0), sine(0.5, 600.0)]), resonance=0.3, mode="lp24", drive=1.0, oversample=2)
print("LadderPE (modulated cutoff)", {k: round(v, 1) for k, v in B.device_rates(spec).items()})
PY
