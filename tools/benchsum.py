import json,sys
d=json.loads(sys.stdin.read())
print("value", d["value"], "ms/step", d["ms_per_step"], d["parity"])
for k in ("roofline","roofline_other_stage"):
    r=d[k]; print(r["kernel"][:40], "ms", r["avg_launch_ms"], "GB/s", r["achieved"], "frac", r["frac"])
