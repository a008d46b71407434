# usage: bash scripts/step_ab.sh <variant names...>: ms_per_step of the headline bench for each build, interleaved twice
for rep in 1 2; do for v in "$@"; do echo -n "$v: "; PCREG_LIB=pcreg_amd/variants/$v.so python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*'; done; done
