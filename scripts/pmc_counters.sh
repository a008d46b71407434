#!/bin/bash
# usage: bash scripts/pmc_counters.sh <tag> <kernel-substring> <out.json (repo-relative; on the GPU box only paths under gpurun_out/ travel back -- or re-run scripts/prof_parse.py here on the merged CSVs)> [VAR=value ...] -- <python script + args>
# SQ / TCP / TCC counters of one kernel, one rocprofv3 --pmc pass per group (no trace options next to --pmc).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; K=$2; OUTJ=$3; shift; shift; shift
while [ "$1" != "--" ] && [ $# -gt 0 ]; do export "$1"; shift; done
shift
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o p -- python3 "$ROOT/$1" "${@:2}" > "$OUT/g$i.log" 2>&1 || echo "group $i failed (see $OUT/g$i.log): $grp"
done
python3 "$ROOT/scripts/prof_parse.py" counters "$OUT" "$K" "$ROOT/$OUTJ"
