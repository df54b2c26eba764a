"""Developer probe: per panel step of the LAST fit in a rocprofv3 kernel trace of `fit_only.py <n>` (nb = 512 regime): the trailing
update's duration against the step's period (start to start of consecutive trailing updates), the main stream's idle time inside
the period and the panel stream's busy time -- which of the two bounds each step.   step_periods.py <kernel_trace.csv> <n>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
n = int(sys.argv[2])
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
gi = [i for i, r in enumerate(rows) if "gram_kernel" in r["Kernel_Name"]]
fit = rows[gi[-1]:]
upd = [r for r in fit if "syrk_lower" in r["Kernel_Name"]]
main_id = upd[0]["Stream_Id"]
main = [r for r in fit if r["Stream_Id"] == main_id]
panel = [r for r in fit if r["Stream_Id"] != main_id]
print(f"last fit {(max(r['e'] for r in fit) - fit[0]['s']) / 1e6:.2f} ms; {len(upd)} trailing updates")
print(" step   rest     upd us   TF/s   period us  main idle us  panel busy us  kernel")
for j in range(len(upd) - 1):
    a, b = upd[j], upd[j + 1]
    rest = n - 512 * (j + 2)
    dur = (a["e"] - a["s"]) / 1e3
    per = (b["s"] - a["s"]) / 1e3
    busy = sum(min(r["e"], b["s"]) - max(r["s"], a["s"]) for r in main if r["e"] > a["s"] and r["s"] < b["s"]) / 1e3
    pb = sum(min(r["e"], b["s"]) - max(r["s"], a["s"]) for r in panel if r["e"] > a["s"] and r["s"] < b["s"]) / 1e3
    fl = rest * (rest + 1) * 512
    print(f"{j:5d} {rest:6d} {dur:10.1f} {fl / dur / 1e6:6.1f} {per:10.1f} {per - busy:12.1f} {pb:13.1f}  {'persist' if 'persist' in a['Kernel_Name'] else 'plain'}")
