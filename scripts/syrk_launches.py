"""Developer probe: the trailing-update launches of the last fit in a rocprofv3 kernel trace of `fit_only.py 32768`: size, duration and
rate of each (the bench line's roofline fraction is their flop-weighted average)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH" and "syrk_lower" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = 32768
# the library's rule (chol.hip, width()): 2048-column panels while more than 22528 rows remain, 1024 down to 16384, then 512
k = 0; out = []
widths = []
rem = n
while rem > 0:
    w = 512 if rem <= 16384 else (2048 if rem > 22528 else 1024)
    w = min(w, rem); widths.append(w); rem -= w
rows = rows[-(len(widths) - 2):]  # one trailing update per panel but the last two (the last one's is part of the look-ahead update)
# SYRK j uses K = widths[j], result = rows after panel j+1
pos = 0
tot_f = tot_t = 0
for j, r in enumerate(rows):
    kb = widths[j]; kb2 = widths[j + 1]
    rest2 = n - (pos + kb + kb2)
    fl = rest2 * (rest2 + 1) * kb
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out.append((j, rest2, kb, dur, fl / dur / 1e6))
    tot_f += fl; tot_t += dur
    pos += kb
for o in out: print("launch %2d rest %5d K %4d  %8.1f us  %5.1f TF/s" % o)
print("total %.1f ms, %.1f TF/s" % (tot_t / 1e3, tot_f / tot_t / 1e6))
