"""Per-launch durations of the progressive stage-1 kernels from a tools/prog_profile.sh run:  python tools/prog_trace_summary.py r02p"""
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/prog_{tag}/stats/runc/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "progressive" in r["Kernel_Name"] or "destuff" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[-n:]:
    print(f'{r["Kernel_Name"][:58]:58s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:9.3f} ms')
for pat in ("pmc_sq", "pmc_sq2"):
    fs = glob.glob(f"gpurun_out/prog_{tag}/{pat}/runc/*counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        if "progressive" in r["Kernel_Name"]:
            agg[(int(r["Dispatch_Id"]), r["Kernel_Name"][:40])][r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(agg)[-n:]:
        if agg[k].get("SQ_WAVES", 1) > 0 and max(agg[k].values()) > 1e7:
            print(k, {c: f"{v:.3g}" for c, v in agg[k].items()})
