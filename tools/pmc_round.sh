#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_round.sh <tag>
# HBM traffic (FETCH_SIZE and WRITE_SIZE, separate passes, each with --kernel-trace only) of the kernels the bench
# prices against the HBM roofline: the C2 filter at its three launch sizes, the FFT convolution at the three
# C3 shapes, the sharded mixes' kernels.
set -o pipefail
tag=${1:-r2}
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc_pass.sh ${tag}_pmc_biquad_$c $c tools/biquad_probe.py > /dev/null 2>&1 || echo "biquad $c failed"
  bash tools/pmc_pass.sh ${tag}_pmc_c3_$c $c tools/c3_probe.py > /dev/null 2>&1 || echo "c3 $c failed"
  bash tools/pmc_pass.sh ${tag}_pmc_mixes_$c $c tools/ss_probe.py > /dev/null 2>&1 || echo "mixes $c failed"
  bash tools/pmc_pass.sh ${tag}_pmc_c2fused_$c $c tools/c2_fused_probe.py > /dev/null 2>&1 || echo "c2 fused $c failed"
  bash tools/pmc_pass.sh ${tag}_pmc_comb_$c $c tools/comb_kernel_probe.py pmc > /dev/null 2>&1 || echo "comb $c failed"
done
ls gpurun_out | grep ${tag}_pmc | head -30
