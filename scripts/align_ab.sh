# usage: bash scripts/align_ab.sh <variant names...>: AlignPoints_KNN batched (4096 x 3000) for each build, interleaved three times
for rep in 1 2 3; do for v in "$@"; do echo -n "$v: "; PCREG_LIB=pcreg_amd/variants/$v.so REPS=10 python scripts/align_dev_bench.py 2>/dev/null | tail -1; done; done
