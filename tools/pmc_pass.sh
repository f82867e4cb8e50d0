#!/bin/bash
# usage: tools/pmc_pass.sh <name> <COUNTER[,COUNTER..]> <script.py> [args...]   (GPU box, repo root)
# One rocprofv3 --pmc pass (own run, with --kernel-trace only, as the pool requires) of
# `python3 <script.py> args`; per-(kernel, grid) averages go to gpurun_out/<name>.md.
set -eo pipefail
name=$1; counters=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
script=$root/$1; shift
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc ${counters//,/ } --kernel-trace --output-format csv -d "$out" -o t -- python3 "$script" "$@" > "$out.log" 2>&1
cd "$root"
python3 tools/summarize_pmc.py "$(find "$out" -name '*counter_collection.csv' | head -1)" > "$out.md"
cat "$out.md"
