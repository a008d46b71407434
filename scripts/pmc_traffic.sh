#!/bin/bash
# usage: bash scripts/pmc_traffic.sh <tag> <out.json (repo-relative; on the GPU box only paths under gpurun_out/ travel back -- or re-run scripts/prof_parse.py here on the merged CSVs)> <16B-reader kernel substrings, comma-separated or ""> -- <python script + args>
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md's recipe).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; OUTJ=$2; DBL=$3; shift; shift; shift
[ "$1" = "--" ] && shift
OUT=$ROOT/gpurun_out/traffic_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/$c.log" 2>&1 || { echo "pass $c failed (see $OUT/$c.log)"; exit 1; }
done
python3 "$ROOT/scripts/prof_parse.py" traffic "$OUT" "$ROOT/$OUTJ" "$DBL"
