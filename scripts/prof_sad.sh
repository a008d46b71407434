cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sad16 -o sad16 -- python3 $GRAFT_REPO_ROOT/scripts/desc_bench.py 5000 20000 980 SAD > /dev/null 2>&1; python3 - <<EOF2
import csv
rows=list(csv.DictReader(open("/root/repo/gpurun_out/prof_sad16/sad16_kernel_trace.csv")))
tot=0
for r in rows:
    d=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    if int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])>4096: tot+=d; print(r["Kernel_Name"].replace("pcreg::(anonymous namespace)::","")[:50], r["Grid_Size_X"], r["Grid_Size_Y"], d/1e3,"us")
print("total of the large call", tot/1e3, "us")
EOF2
