# usage: bash scripts/knn_ab.sh <variant names...>: the search call's time for each build, interleaved REPS times (boxes drift)
for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do echo -n "$v: "; PCREG_LIB=pcreg_amd/variants/$v.so python scripts/knn_sweep.py | tail -1; done; done
