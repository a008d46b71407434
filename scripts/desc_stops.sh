# usage: bash scripts/desc_stops.sh <mode: double|single> <variant.so ...>: phase times of the descriptor call via PCREG_DESC_STOP (EXPERIMENTS builds)
M=$1; shift
for v in "$@"; do for s in 0 1 4 6 2 7 8 3; do echo -n "$v $M stop=$s: "; PCREG_LIB=$v PCREG_DESC_STOP=$s python scripts/desc_dev_bench.py 1000000 100000 $M | tail -1; done; done
