#!/bin/bash
# usage: bash scripts/rq_ab.sh <variant names...>: the pairs kernel's duration and the sweep's time for each build (pcreg_amd/variants/<name>.so)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for v in "$@"; do
  bash scripts/prof_stats.sh ab_$v 3 -m segp_rerank PCREG_LIB=$ROOT/pcreg_amd/variants/$v.so -- scripts/sweep_prof.py 3 > gpurun_out/ab_$v.txt 2>&1
  echo "$v: $(grep segp_rerank_pairs gpurun_out/ab_$v.txt | head -1 | cut -c60-140)"
done
for rep in 1 2; do for v in "$@"; do
  echo "$v $(PCREG_LIB=$ROOT/pcreg_amd/variants/$v.so python scripts/sweep_bench.py 2>/dev/null | grep -o 'segmented_ms.: [0-9.]*')"
done; done
