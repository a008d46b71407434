# usage: bash scripts/prof_desc.sh "<ENV=val ...>" ... : kernel-trace scripts/desc_compute_bench.py with each env set; print desc_kernel's time
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_desc
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_desc -o p -- python3 $GRAFT_REPO_ROOT/scripts/desc_compute_bench.py $DESC_ARGS > /dev/null 2>&1
  python3 - "$e" <<EOF2
import csv,sys
for r in csv.DictReader(open("/root/repo/gpurun_out/prof_desc/p_kernel_stats.csv")):
    if "desc_kernel" in r["Name"]: print(sys.argv[1], "|", r["Name"].replace("pcreg::(anonymous namespace)::","")[:40], "calls", r["Calls"], "max ms %.2f" % (float(r["MaxNs"])/1e6))
EOF2
done
