"""Print the kernels of the LAST burst in a rocprofv3 kernel trace (gpurun_out/prof_any/p_kernel_trace.csv)."""
import csv, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_any/p_kernel_trace.csv"
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
groups = [[rows[0]]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 1_000_000: groups.append([])
    groups[-1].append(b)
g = groups[-1]; t0 = int(g[0]["Start_Timestamp"])
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].replace("pcreg::(anonymous namespace)::", "")[:64]))
print("total %.1f us, %d kernels" % ((int(g[-1]["End_Timestamp"]) - t0) / 1e3, len(g)))
