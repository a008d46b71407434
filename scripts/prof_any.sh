# usage: bash scripts/prof_any.sh <script + args>: per-kernel averages (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_any -o p -- python3 $GRAFT_REPO_ROOT/"$@" > /dev/null 2>&1; python3 - <<EOF2
import csv
for r in list(csv.DictReader(open("/root/repo/gpurun_out/prof_any/p_kernel_stats.csv")))[:18]:
    print(r["Name"].replace("pcreg::(anonymous namespace)::","")[:64], r["Calls"], "avg us %.1f" % (float(r["AverageNs"])/1e3), "total ms %.2f" % (float(r["TotalDurationNs"])/1e6))
EOF2
