# usage: bash scripts/desc_ab.sh <variant names...>: the descriptor call's time (1 M points, 100 k keypoints) for each build, interleaved three times
for rep in 1 2 3; do for v in "$@"; do echo -n "$v: "; PCREG_LIB=pcreg_amd/variants/$v.so python scripts/desc_dev_bench.py 1000000 100000 double 2>/dev/null | tail -1; done; done
