# HBM traffic of the search call: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (KiB per dispatch),
# summed over the kernels of one pcreg_dev_knn2_points_f32 call.  Writes gpurun_out/pmc_knn/summary.json.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_knn/$c -o p -- python3 $GRAFT_REPO_ROOT/scripts/knn_sweep.py > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<EOF2
import csv, glob, collections, json, re
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/root/repo/gpurun_out/pmc_knn/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("pcreg::(anonymous namespace)::", "").replace("void ", ""))
                acc[n].append(float(r["Counter_Value"]))
    out[c] = {k: sum(v) / len(v) for k, v in acc.items() if "at::" not in k and "elementwise" not in k}
names = sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]))
tot = 0.0
rows = {}
for n in names:
    f, w = out["FETCH_SIZE"].get(n, 0.0), out["WRITE_SIZE"].get(n, 0.0)
    b = (2 * f + w) * 1024            # FETCH doubled (16-B/lane reads report half on gfx950), KiB -> bytes
    rows[n] = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "bytes": int(b)}
    tot += b
    print(f"{n[:56]:56s} fetch {f:10.1f} KiB  write {w:10.1f} KiB  -> {b/1e6:8.2f} MB")
print("total per search call: %.1f MB" % (tot / 1e6))
json.dump({"kernels": rows, "total_bytes": int(tot)}, open("/root/repo/gpurun_out/pmc_knn/summary.json", "w"), indent=1)
EOF2
