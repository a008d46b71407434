# per-kernel PMC counters of the SAD candidates kernel; one rocprofv3 pass per counter group
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sad/g$i -o p -- python3 $GRAFT_REPO_ROOT/scripts/desc_bench.py 5000 20000 980 SAD > /dev/null 2>&1 || echo "group $i failed: $grp"
done
python3 - <<EOF2
import csv,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("/root/repo/gpurun_out/pmc_sad/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "sad16_candidates" in r["Kernel_Name"] and int(r["Grid_Size"])>100000*16:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
EOF2
