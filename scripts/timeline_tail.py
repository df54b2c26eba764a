"""Developer probe: the last K kernel launches of a rocprofv3 --kernel-trace csv, in order, with durations and the gaps between them."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get("Kind", "KERNEL_DISPATCH") == "KERNEL_DISPATCH"]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sel = rows[-K:]
t0 = sel[0]["s"]
prev = None
tot = 0
for a in sel:
    gap = (a["s"] - prev) / 1e3 if prev is not None else 0.0
    name = a["Kernel_Name"].split("(")[0].replace("void fr::", "").replace("fr::", "")[:44]
    wgs = int(a["Grid_Size_X"]) // int(a["Workgroup_Size_X"])
    print(f"t={(a['s'] - t0) / 1e3:9.1f} us  gap {gap:6.1f}  dur {(a['e'] - a['s']) / 1e3:8.1f}  {name:44s} grid {wgs} x {a['Grid_Size_Y']}")
    prev = a["e"]; tot += a["e"] - a["s"]
print(f"span {(sel[-1]['e'] - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us")
