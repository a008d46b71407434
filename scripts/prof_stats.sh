#!/bin/bash
# usage: bash scripts/prof_stats.sh <tag> <steps> [-m kernel-substring] [VAR=value ...] -- <python script + args>
# rocprofv3 --kernel-trace --stats of one program (the program itself sits directly after `--`); prints per-kernel
# averages and per-step totals.  Output: gpurun_out/prof_<tag>/ (cleared first), the program's stdout/stderr in run.log.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; STEPS=$2; shift; shift
MATCH=""
if [ "$1" = "-m" ]; then MATCH=$2; shift; shift; fi
while [ "$1" != "--" ] && [ $# -gt 0 ]; do export "$1"; shift; done
shift
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1
rc=$?
if [ $rc -ne 0 ] || [ ! -f "$OUT/p_kernel_stats.csv" ]; then echo "prof_stats($TAG): the profiled run failed (rc $rc)"; tail -5 "$OUT/run.log"; exit 1; fi
python3 "$ROOT/scripts/prof_parse.py" stats "$OUT" "$STEPS" "$MATCH"
