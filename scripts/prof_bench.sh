# usage: bash scripts/prof_bench.sh [bench args]: rocprofv3 --kernel-trace --stats of the headline step; prints per-kernel per-step times
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bench; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_bench/line.json 2>/dev/null
python3 - <<'EOF2'
import csv
rows = list(csv.DictReader(open("/root/repo/gpurun_out/prof_bench/p_kernel_stats.csv")))
steps = 23
for r in rows:
    n = r["Name"].replace("pcreg::(anonymous namespace)::", "")[:72]
    print("%-72s %5d avg us %9.1f  per-step us %8.1f" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
print(len(rows), "kernels; launches per step: %.1f" % (sum(int(r["Calls"]) for r in rows) / steps))
EOF2
