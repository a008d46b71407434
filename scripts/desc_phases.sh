#!/bin/bash
# the descriptor call's time (1 M points, 100 k keypoints) with desc_kernel cut after each phase (PCREG_DESC_STOP, an
# EXPERIMENTS build: pcreg_amd/variants/exp.so): 1 collection, 4 keys + min/max, 5 histogram, 6 K-th + tie counts, 2 selection,
# 7 moments, 8 Jacobi, 3 vote + histogram pass, 0 everything
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIBV=${1:-$ROOT/pcreg_amd/variants/exp.so}
for st in 1 4 5 6 2 7 8 3 0; do echo -n "stop $st: "; PCREG_LIB=$LIBV PCREG_DESC_STOP=$st python scripts/desc_dev_bench.py 1000000 100000 double 2>/dev/null | tail -1; done
