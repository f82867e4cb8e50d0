bash tools/kernel_trace.sh r3r_shard_c5_trace tools/shard_probe.py > /dev/null 2>&1; echo trace rc=$?; head -24 gpurun_out/r3r_shard_c5_trace.md
bash tools/pmc_pass.sh r3_pmc_conv_mfma SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU_MFMA_MOPS_F32,SQ_WAVE_CYCLES,GRBM_GUI_ACTIVE tools/conv_probe.py 96000 5 > /dev/null 2>&1; echo pmc rc=$?; cat gpurun_out/r3_pmc_conv_mfma.md | grep conv_mfma
