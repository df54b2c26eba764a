"""Developer probe: per-stream timeline of the LAST fit in a rocprofv3 kernel trace (x_kernel_trace.csv of fit_only.py):
busy time and gaps per stream, and the sequence of launches on the panel stream for a window of the fit."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kind"] == "KERNEL_DISPATCH"]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the last fit starts at the last gram_kernel group
gi = [i for i, r in enumerate(rows) if "gram_kernel" in r["Kernel_Name"]]
first = gi[-1]
while first - 1 in gi or (first > 0 and "gram" in rows[first - 1]["Kernel_Name"]):
    first -= 1
fit = rows[first:]
t0 = fit[0]["s"]; t1 = max(r["e"] for r in fit)
print(f"last fit: {(t1 - t0) / 1e6:.2f} ms, {len(fit)} launches")
streams = {}
for r in fit:
    streams.setdefault(r["Stream_Id"], []).append(r)
def short(n):
    for k in ("syrk_lower", "panel_chain", "potf2", "gemm_f64", "gram", "copy", "fill", "pairdist"):
        if k in n: return k
    return n[:20]
for sid, rs in streams.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    kinds = {}
    for r in rs:
        k = short(r["Kernel_Name"]); a = kinds.setdefault(k, [0, 0]); a[0] += 1; a[1] += r["e"] - r["s"]
    print(f"stream {sid}: {len(rs)} launches, busy {busy / 1e6:.2f} ms, first {(rs[0]['s'] - t0) / 1e6:.2f} last {(rs[-1]['e'] - t0) / 1e6:.2f}  " +
          "  ".join(f"{k}: {v[0]} x {v[1] / v[0] / 1e3:.1f} us" for k, v in kinds.items()))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
hi = float(sys.argv[3]) if len(sys.argv) > 3 else lo + 0.06
print(f"--- launches between {lo:.2f} and {hi:.2f} of the fit")
for r in fit:
    f = (r["s"] - t0) / (t1 - t0)
    if lo <= f <= hi:
        print(f"  t={(r['s'] - t0) / 1e3:9.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f}  stream {r['Stream_Id']}  {short(r['Kernel_Name']):12s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}")
