# usage: PCREG_KNN_VARIANT=.. bash scripts/prof_knn.sh  -> per-kernel average durations of the search call
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_knn -o knn -- python3 $GRAFT_REPO_ROOT/scripts/knn_sweep.py > /dev/null 2>&1; python3 - <<EOF2
import csv
for r in list(csv.DictReader(open("/root/repo/gpurun_out/prof_knn/knn_kernel_stats.csv")))[:12]:
    print(r["Name"].replace("pcreg::(anonymous namespace)::","")[:60], r["Calls"], "avg us", float(r["AverageNs"])/1e3)
EOF2
