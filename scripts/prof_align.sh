# usage: bash scripts/prof_align.sh "<ENV=val ...>" ... : kernel-trace scripts/align_bench.py with each env set; print the align kernels' avg
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_align
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_align -o p -- python3 $GRAFT_REPO_ROOT/scripts/align_bench.py $ALIGN_ARGS > /dev/null 2>&1
  python3 - "$e" <<EOF2
import csv,sys
for r in csv.DictReader(open("/root/repo/gpurun_out/prof_align/p_kernel_stats.csv")):
    if "align_points" in r["Name"]: print(sys.argv[1], "|", r["Name"].replace("pcreg::(anonymous namespace)::","")[:60], "calls", r["Calls"], "max us %.1f" % (float(r["MaxNs"])/1e3))
EOF2
done
