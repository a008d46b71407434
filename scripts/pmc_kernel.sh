# usage: bash scripts/pmc_kernel.sh <kernel-substring> <python script + args...>: SQ counters of one kernel
K=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_k/g$i -o p -- python3 $GRAFT_REPO_ROOT/"$@" > /dev/null 2>&1 || echo "group $i failed: $grp"
done
python3 - "$K" <<EOF2
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in sorted(glob.glob("/root/repo/gpurun_out/pmc_k/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:28s} {sum(v)/len(v):16.0f}  ({len(v)} dispatches)")
EOF2
