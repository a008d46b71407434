# usage: bash scripts/prof_exp.sh "<ENV=val ...>" ... : kernel-trace the search with each env set; print the candidates kernel's avg
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_exp
  env $e rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_exp -o p -- python3 $GRAFT_REPO_ROOT/scripts/knn_sweep.py > /dev/null 2>&1
  python3 - "$e" <<EOF2
import csv,sys
for r in csv.DictReader(open("/root/repo/gpurun_out/prof_exp/p_kernel_stats.csv")):
    if "knn_candidates" in r["Name"]: print(sys.argv[1], "|", r["Name"].replace("pcreg::(anonymous namespace)::","")[:50], "avg us %.1f" % (float(r["AverageNs"])/1e3))
EOF2
done
