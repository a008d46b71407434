#!/bin/bash
# usage: bash scripts/desc_valu_by_phase.sh <variant.so>: vector / scalar instructions of desc_kernel up to each PCREG_DESC_STOP
# (EXPERIMENTS build), one rocprofv3 --pmc pass per stop; differences between consecutive stops = the phase's instructions
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$1
OUT=$ROOT/gpurun_out/pmc_desc_phase
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PCREG_LIB=$ROOT/$LIB
for s in 1 4 6 2 7 8 3 0; do
  export PCREG_DESC_STOP=$s
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/s$s" -o p -- python3 "$ROOT/scripts/desc_dev_bench.py" 1000000 100000 double > "$OUT/s$s.log" 2>&1 || echo "stop $s failed"
  python3 "$ROOT/scripts/prof_parse.py" counters "$OUT/s$s" desc_kernel "$OUT/s$s.json" > /dev/null
  python3 -c "
import json
d=json.load(open('$OUT/s$s.json'))['counters']
print('stop $s', {k:int(v['max']/400000) for k,v in d.items()})
"
done
