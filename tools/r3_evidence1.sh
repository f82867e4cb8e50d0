set -o pipefail
bash tools/pmc_round.sh r3 > /dev/null
bash tools/pmc_pass.sh r3_pmc_conv_mfma SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU_MFMA_MOPS_F32,SQ_WAVE_CYCLES,GRBM_GUI_ACTIVE tools/conv_probe.py 96000 5 > /dev/null 2>&1; echo "mfma pmc rc=$?"
python tools/shard_probe.py supersaw > gpurun_out/r3_shard_supersaw.txt 2>&1; cat gpurun_out/r3_shard_supersaw.txt
python tools/shard_probe.py > gpurun_out/r3_shard_c5.txt 2>&1; cat gpurun_out/r3_shard_c5.txt
./tools/microbench/ss_phases 64 4 48000 > gpurun_out/r3_ss_phases.txt; cat gpurun_out/r3_ss_phases.txt
python tools/comb_kernel_probe.py > gpurun_out/r3_comb_kernel.json; cat gpurun_out/r3_comb_kernel.json
python tools/c2_kernel_probe.py > gpurun_out/r3_c2_kernel.txt; cat gpurun_out/r3_c2_kernel.txt
python tools/c2_steps.py > gpurun_out/r3_c2_steps.txt; cat gpurun_out/r3_c2_steps.txt
python tools/conv_probe.py 96000 5
