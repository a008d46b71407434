#!/bin/bash
# usage: bash scripts/ransac_phase_ab.sh <variant names...>: the sweep's batched RANSAC launch per build (timing-only builds: -DNO_MOM /
# -DNO_DENSE skip phases of ransac_hyp32_kernel and give wrong results)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for v in "$@"; do
  bash scripts/prof_stats.sh rp_$v 3 -m ransac_hyp32 PCREG_LIB=$ROOT/pcreg_amd/variants/$v.so -- scripts/sweep_prof.py 3 > gpurun_out/rp_$v.txt 2>&1
  echo "$v: $(grep ransac_hyp32 gpurun_out/rp_$v.txt | head -1 | cut -c60-140)"
done
