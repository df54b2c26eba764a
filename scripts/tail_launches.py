"""Developer probe: the last K launches of a rocprofv3 kernel trace with start offsets, durations and gaps (what a single call --
e.g. the sample_at of config_run.py 4 -- is made of).   tail_launches.py <kernel_trace.csv> [K]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
rows = rows[-K:]
t0 = rows[0]["s"]
prev_e = t0
for r in rows:
    name = r["Kernel_Name"].replace("fr::", "").split("(")[0][:46]
    print(f"t={(r['s'] - t0) / 1e3:8.1f} us  gap {(r['s'] - prev_e) / 1e3:7.1f}  dur {(r['e'] - r['s']) / 1e3:7.1f}  stream {r['Stream_Id']}  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}  {name}")
    prev_e = max(prev_e, r["e"])
