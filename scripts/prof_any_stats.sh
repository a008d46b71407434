# usage: bash scripts/prof_any_stats.sh <python script + args...>: rocprofv3 --kernel-trace --stats, top kernels by total time
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_any; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_any
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_any -o p -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_any/out.txt 2>&1
python3 - <<'EOF2'
import csv
rows = list(csv.DictReader(open("/root/repo/gpurun_out/prof_any/p_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    n = r["Name"].replace("pcreg::(anonymous namespace)::", "").replace("void ", "")[:64]
    print("%-64s %6d calls  avg us %8.1f  total ms %8.2f  %5.1f %%" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("kernels total ms %.2f, launches %d" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
EOF2
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_any/out.txt
