"""profiles/traffic.json from a parsed PMC traffic run (scripts/pmc_traffic.sh on bench.py): the HBM bytes per launch of the
search's dominant kernel, tied to the sha256 of the kernel sources it was measured on (bench.py reports `null` for any other build).
usage: python scripts/make_traffic_json.py <parsed traffic json> <Q> <rows> <label>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash, ROOT

SOURCES = ["pcreg_amd/csrc/knn_mfma16.hip", "pcreg_amd/csrc/knn_fast_common.hpp"]
src, Q, rows, label = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
k = [v for n, v in json.load(open(src))["kernels"].items() if "knn_candidates_f16" in n]
assert len(k) == 1, "expected exactly one knn_candidates_f16 kernel in " + src
path = os.path.join(ROOT, "profiles", "traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out = {n: v for n, v in out.items() if isinstance(v, dict) or n.startswith("_")}
out[f"knn_search:Q{Q}:M{rows}"] = {"bytes": k[0]["bytes"], "fetch_kib": k[0]["fetch_kib_largest_dispatch"], "fetch_factor": k[0]["fetch_factor"],
                                   "write_kib": k[0]["write_kib_largest_dispatch"], "sources": SOURCES, "sources_sha256": kernel_source_hash(SOURCES),
                                   "measured": label}
out["_note"] = ("knn_candidates_f16_pipe_kernel: (fetch_factor * FETCH_SIZE + WRITE_SIZE) KiB per launch; FETCH doubled per MI355X_MICROARCH.md "
                "(16-B/lane LDS-DMA reads report half); separate --pmc passes (scripts/pmc_traffic.sh); bench.py compares sources_sha256 with the build it runs")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[f"knn_search:Q{Q}:M{rows}"]))
