#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_sq.sh <tag> <script.py> [args...]
# Three rocprofv3 --pmc passes of SQ counters (issue, waits, instruction mix) over `python3 <script.py> args`;
# per-(kernel, grid) averages in gpurun_out/<tag>_sq_{a,b,c}.md
tag=$1; shift
bash tools/pmc_pass.sh ${tag}_sq_a SQ_WAVES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU "$@" > /dev/null 2>&1 || echo "pass a failed"
bash tools/pmc_pass.sh ${tag}_sq_b SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM "$@" > /dev/null 2>&1 || echo "pass b failed"
bash tools/pmc_pass.sh ${tag}_sq_c SQ_ACTIVE_INST_MISC,SQ_INST_CYCLES_SALU,SQ_INST_CYCLES_SMEM,SQ_WAIT_INST_LDS,SQ_IFETCH,SQ_INST_LEVEL_SMEM,SQ_INST_LEVEL_VMEM,SQ_THREAD_CYCLES_VALU "$@" > /dev/null 2>&1 || echo "pass c failed"
ls gpurun_out | grep ${tag}_sq
