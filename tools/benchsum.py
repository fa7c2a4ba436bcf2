import json,sys
d=json.loads((open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()).strip().splitlines()[-1])   # a file, or the line on stdin
print("value", d["value"], "ms/step", d["ms_per_step"], d["parity"])
for k in ("roofline","roofline_other_stage"):
    r=d[k]; print(r["kernel"][:40], "ms", r["avg_launch_ms"], "GB/s", r["achieved"], "frac", r["frac"])
