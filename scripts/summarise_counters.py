"""Developer tool: fold rocprofv3 passes (kernel stats + FETCH_SIZE / WRITE_SIZE / SQ counters) of one workload into a JSON
summary under profiles/.   summarise_counters.py <prefix under gpurun_out> <output json> <workload description>"""
import collections
import csv
import glob
import json
import os
import sys

prefix, out_path, desc = sys.argv[1], sys.argv[2], sys.argv[3]


def newest(pat):
    fs = glob.glob(pat) or glob.glob(pat.replace("/runc/", "/**/"), recursive=True) or glob.glob(pat.replace("/runc/", "/"))
    fs.sort(key=os.path.getmtime)
    return fs[-1]


out = {"workload": desc,
       "units": "FETCH_SIZE doubled (gfx950 correction, scripts/fetch_calib), WRITE_SIZE as reported; bytes are L2 memory-side "
                "requests (Infinity-Cache hits included); MFMA rate from SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 flop over the "
                "kernel's total duration (kernel-stats pass)",
       "kernels": {}}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ("fetch", "write", "sq"):
    for row in csv.DictReader(open(newest(f"gpurun_out/{prefix}_{tag}/runc/*_counter_collection.csv"))):
        agg[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]] += float(row["Counter_Value"])
st = {}
for row in csv.DictReader(open(newest(f"gpurun_out/{prefix}_stats/runc/*_kernel_stats.csv"))):
    st[row["Name"].split("(")[0].replace("void ", "")] = (int(row["Calls"]), float(row["TotalDurationNs"]))
for k, a in agg.items():
    if k not in st:
        continue
    calls, tot = st[k]
    if tot < 2e6:
        continue
    e = {"launches": calls, "total_ms": round(tot / 1e6, 3)}
    if "FETCH_SIZE" in a:
        e["fetch_GB"] = round(a["FETCH_SIZE"] * 1024 * 2 / 1e9, 3)
    if "WRITE_SIZE" in a:
        e["write_GB"] = round(a["WRITE_SIZE"] * 1024 / 1e9, 3)
    if "fetch_GB" in e and "write_GB" in e:
        e["fabric_TB_per_s"] = round((e["fetch_GB"] + e["write_GB"]) / (tot / 1e9) / 1e3, 3)
    sq = {k2: v for k2, v in a.items() if k2.startswith("SQ_")}
    if sq:
        e["sq"] = sq
        if sq.get("SQ_BUSY_CYCLES") and sq.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles summed over waves; per-wave VALU-issue share:
            if sq.get("SQ_WAVE_CYCLES"):
                e["valu_active_share_of_wave_cycles"] = round(sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], 3)
            if sq.get("SQ_WAIT_INST_ANY") and sq.get("SQ_WAVE_CYCLES"):
                e["wait_inst_share_of_wave_cycles"] = round(sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 3)
    if "fetch_GB" in e and "write_GB" in e:
        e["fetch_bytes_per_launch"] = e["fetch_GB"] * 1e9 / calls
        e["write_bytes_per_launch"] = e["write_GB"] * 1e9 / calls
        e["total_bytes_per_launch"] = (e["fetch_GB"] + e["write_GB"]) * 1e9 / calls
    if a.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        fl = a["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
        e["mfma_TFLOP_per_s"] = round(fl / (tot / 1e9) / 1e12, 2)
        e["mfma_frac_of_78.6"] = round(e["mfma_TFLOP_per_s"] / 78.6, 3)
    out["kernels"][k] = e
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import source_hashes  # noqa: E402

out["sources"] = source_hashes()  # the device sources these passes ran (tests/test_profiles_fresh.py)
json.dump(out, open(out_path, "w"), indent=1)
for k, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
    print(k[:44], e)
