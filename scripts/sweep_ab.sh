# usage: bash scripts/sweep_ab.sh <variant names...>: the segmented sweep's time for each build, interleaved twice
for rep in 1 2; do for v in "$@"; do echo -n "$v: "; PCREG_LIB=pcreg_amd/variants/$v.so python scripts/sweep_bench.py | grep -o '"segmented_ms": [0-9.]*'; done; done
