# usage: bash scripts/pmc_traffic.sh <out.json> <python script + args...>
# HBM traffic per kernel of any script: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (KiB per dispatch, mean over
# the dispatches of a kernel), program directly after `--`.  FETCH is doubled for 16-B/lane readers as in pmc_knn.sh? No:
# reported raw here, with the raw KiB next to it -- the kernels profiled with this script read 8-B lanes.
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_t
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_t/$c -o p -- python3 $GRAFT_REPO_ROOT/"$@" > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - "$GRAFT_REPO_ROOT/$OUT" <<EOF2
import csv, glob, collections, json, re, sys
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/root/repo/gpurun_out/pmc_t/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("pcreg::(anonymous namespace)::", "").replace("void ", ""))
                acc[n].append(float(r["Counter_Value"]))
    out[c] = {k: (max(v), len(v)) for k, v in acc.items() if "at::" not in k and "elementwise" not in k}
rows = {}
for n in sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"])):
    f, nf = out["FETCH_SIZE"].get(n, (0.0, 0)); w, nw = out["WRITE_SIZE"].get(n, (0.0, 0))
    rows[n] = {"fetch_kib_max_dispatch": round(f, 1), "write_kib_max_dispatch": round(w, 1), "dispatches": max(nf, nw)}
    print(f"{n[:56]:56s} fetch {f/1024:10.2f} MiB  write {w/1024:10.2f} MiB  ({max(nf, nw)} dispatches, largest)")
json.dump({"note": "KiB of the LARGEST dispatch of each kernel (the scripts warm up on a small problem first); raw counter values", "kernels": rows}, open(sys.argv[1], "w"), indent=1)
EOF2
