"""Parsers behind scripts/prof_stats.sh, pmc_counters.sh and pmc_traffic.sh (rocprofv3 CSV output -> tables / JSON).
Every reader takes the output directory of THIS run; a missing file is an error, never an old run's numbers."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    return name.replace("pcreg::(anonymous namespace)::", "").replace("pcreg::", "").replace("void ", "")


def stats(outdir: str, steps: float, match: str) -> None:
    f = os.path.join(outdir, "p_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    shown = [r for r in rows if match in r["Name"]] if match else rows[:40]
    for r in shown:
        print("%-66s %6d calls  avg us %9.1f  max us %9.1f  per-step us %8.1f  %5.1f %%" % (
            short(r["Name"])[:66], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            float(r["TotalDurationNs"]) / 1e3 / steps, 100 * float(r["TotalDurationNs"]) / tot))
    own = [r for r in rows if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
    print("%d kernels, total %.2f ms; launches of this library per step: %.1f" % (len(rows), tot / 1e6, sum(int(r["Calls"]) for r in own) / steps))


def counters(outdir: str, kernel: str, out_json: str) -> None:
    acc = collections.defaultdict(list)
    files = sorted(glob.glob(os.path.join(outdir, "g*", "**", "*counter_collection.csv"), recursive=True))
    if not files:
        sys.exit(f"no counter files under {outdir}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        sys.exit(f"no dispatch of a kernel matching {kernel!r} in {outdir}")
    res = {k: {"mean": sum(v) / len(v), "max": max(v), "dispatches": len(v)} for k, v in sorted(acc.items())}
    for k, v in res.items():
        print(f"{k:34s} mean {v['mean']:18.0f}  max {v['max']:18.0f}  ({v['dispatches']} dispatches)")
    json.dump({"kernel": kernel, "counters": res}, open(out_json, "w"), indent=1)


def traffic(outdir: str, out_json: str, double_fetch: str) -> None:
    """FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  double_fetch: comma-separated kernel substrings whose reads are 16 B
    per lane (global_load_dwordx4 / LDS-DMA): gfx950 reports half of those bytes (MI355X_MICROARCH.md), so they are doubled."""
    dbl = [d for d in double_fetch.split(",") if d]
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        files = glob.glob(os.path.join(outdir, c, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            sys.exit(f"no {c} pass under {outdir}")
        for f in files:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    acc[re.sub(r"\(.*", "", short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
        out[c] = {k: v for k, v in acc.items() if "at::" not in k and "elementwise" not in k}
    rows, total = {}, 0.0
    for n in sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"])):
        f = out["FETCH_SIZE"].get(n, [0.0]); w = out["WRITE_SIZE"].get(n, [0.0])
        k = 2.0 if any(d in n for d in dbl) else 1.0
        fb, wb = k * max(f) * 1024, max(w) * 1024
        rows[n] = {"fetch_kib_largest_dispatch": round(max(f), 1), "fetch_factor": k, "write_kib_largest_dispatch": round(max(w), 1),
                   "bytes": int(fb + wb), "dispatches": max(len(f), len(w))}
        total += fb + wb
        print(f"{n[:60]:60s} fetch {fb / 2**20:10.2f} MiB  write {wb / 2**20:10.2f} MiB  ({rows[n]['dispatches']} dispatches, largest)")
    print("sum over kernels (largest dispatch each): %.1f MB" % (total / 1e6))
    json.dump({"note": "bytes of the LARGEST dispatch of each kernel; FETCH doubled where fetch_factor = 2 (16-B-per-lane readers, gfx950)",
               "kernels": rows, "total_bytes": int(total)}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    {"stats": lambda a: stats(a[0], float(a[1]), a[2] if len(a) > 2 else ""),
     "counters": lambda a: counters(a[0], a[1], a[2]),
     "traffic": lambda a: traffic(a[0], a[1], a[2] if len(a) > 2 else "")}[sys.argv[1]](sys.argv[2:])
