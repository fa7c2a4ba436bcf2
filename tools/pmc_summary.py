"""Mean per kernel launch of every counter under gpurun_out/pmc_<tag>/set*/ (tools/pmc_probe.sh)."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/set*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(w in k for w in ("huffman", "reconstruct", "progressive", "destuff", "scan_markers", "sync", "vsegs", "fused", "recon", "copy")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in agg.items()}
print(json.dumps(out, indent=1))
