# usage: bash scripts/pmc_mfma.sh <kernel-substring> <out.json> <python script + args...>
# SQ issue/matrix-pipe counters of one kernel (separate --pmc passes, program directly after `--`).
K=$1; OUT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_m; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc_m
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_m/g$i -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_m/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - "$K" "$GRAFT_REPO_ROOT/$OUT" <<EOF2
import csv,glob,collections,sys,json
acc=collections.defaultdict(list)
for f in sorted(glob.glob("/root/repo/gpurun_out/pmc_m/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res={k:{"mean":sum(v)/len(v),"dispatches":len(v)} for k,v in sorted(acc.items())}
for k,v in res.items(): print(f"{k:32s} {v['mean']:18.0f}  ({v['dispatches']} dispatches)")
json.dump({"kernel":sys.argv[1],"counters":res}, open(sys.argv[2],"w"), indent=1)
EOF2
