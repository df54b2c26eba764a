"""Developer probe: what the wide solves (scripts/solve_split.py under rocprofv3 --kernel-trace) spend their time on: launches
after the last trailing update of the fit, grouped by kernel and grid size, with the gaps between consecutive launches."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
last = max(i for i, r in enumerate(rows) if "syrk_lower" in r["Kernel_Name"] or "potf2" in r["Kernel_Name"])
sol = rows[last + 1:]
t0, t1 = sol[0]["s"], max(r["e"] for r in sol)
print(f"after the fit: {len(sol)} launches over {(t1 - t0) / 1e6:.1f} ms")
groups = {}
for r in sol:
    name = r["Kernel_Name"].split("(")[0].replace("void fr::", "").replace("fr::", "")
    key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"])))
    g = groups.setdefault(key, [0, 0])
    g[0] += 1; g[1] += r["e"] - r["s"]
tot = sum(g[1] for g in groups.values())
print(f"sum of kernel durations {tot / 1e6:.1f} ms")
for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {key[0]:50s} wgs {key[1]:7d}  x {g[0]:5d}  avg {g[1] / g[0] / 1e3:9.1f} us  total {g[1] / 1e6:8.2f} ms")
gaps = sorted(((b["s"] - a["e"]) for a, b in zip(sol, sol[1:])), reverse=True)
print("gap total (positive) %.2f ms; largest %s us" % (sum(g for g in gaps if g > 0) / 1e6, [round(g / 1e3, 1) for g in gaps[:8]]))
if len(sys.argv) > 2:
    k = int(sys.argv[2])
    for a in sol[:k]:
        print(f"  t={(a['s'] - t0) / 1e3:9.1f} us dur {(a['e'] - a['s']) / 1e3:8.1f}  {a['Kernel_Name'].split('(')[0].replace('void fr::', '')[:40]:40s} grid {int(a['Grid_Size_X']) // int(a['Workgroup_Size_X'])} x {a['Grid_Size_Y']}")
