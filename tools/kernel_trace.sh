#!/bin/bash
# usage: tools/kernel_trace.sh <name> <script.py> [args...]   (run on the GPU box, from the repo root)
# rocprofv3 kernel trace + stats of `python3 <script.py> args`, summarised per (kernel, grid) into
# gpurun_out/<name>.md.  The program follows `--` directly (no env/bash hop, see the gpurun rules).
set -eo pipefail
name=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
script=$root/$1; shift
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 "$script" "$@" > "$out.log" 2>&1
cd "$root"
python3 tools/summarize_trace.py "$(find "$out" -name '*kernel_trace.csv' | head -1)" > "$out.md"
cat "$out.md"
