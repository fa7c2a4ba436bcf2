"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed summaries
under profiles/: <tag>_bench_n1.json, <tag>_kernel_stats_batch1024.csv, <tag>_hbm_traffic_batch1024.json,
<tag>_sq_counters_batch1024.json."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import sources_sha16       # noqa: E402  (bench.py quotes the traffic only for the sources it was taken from)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01x"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"

bench = (src / "bench.json").read_text().strip().splitlines()[-1]
json.loads(bench)
(dst / f"{tag}_bench_n1.json").write_text(bench + "\n")
stats = sorted(glob.glob(str(src / "stats" / "*" / "*kernel_stats.csv")), key=lambda f: -Path(f).stat().st_size)     # (the traced process's own: the largest)
if stats:
    shutil.copy(stats[0], dst / f"{tag}_kernel_stats_batch1024.csv")
    # the line the traced process itself printed (tools/profile_round.sh): the trace's averages are held against THIS line
    sl = src / "stats_bench_line.json"
    if sl.exists() and sl.read_text().strip():
        line = sl.read_text().strip().splitlines()[-1]
        d = json.loads(line)
        (dst / f"{tag}_kernel_stats_bench_line.json").write_text(line + "\n")
        rows = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(stats[0]))}
        dom = [k for k in rows if "fused" in k] or [k for k in rows if "reconstruct_fast" in k]
        if dom:
            r = rows[dom[0]]
            print(f"same process: {dom[0][:50]} rocprof average {float(r['AverageNs']) / 1e6:.3f} ms over {r['Calls']} calls (min {float(r['MinNs']) / 1e6:.3f}); "
                  f"the line: ms_per_step {d['ms_per_step']}, HIP-event launch {d['roofline']['avg_launch_ms']} ms, "
                  f"shader clock {d['roofline'].get('shader_clock_mhz')} MHz")


def per_kernel(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(src / pattern / "*" / "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if any(w in k for w in ("huffman", "reconstruct", "progressive", "destuff", "scan_markers", "sync", "vsegs", "planes", "fused", "count")):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


traffic = {"sources_sha16": sources_sha16(), "note": "rocprofv3 --pmc passes (one counter set per pass, tools/profile_round.sh) of `python3 bench.py --steps 1 "
                   "--warmup 1 --no-cpu-baseline` on MI355X, 1024 x 1080p 4:2:0 DRI=120 per launch; values are per kernel launch "
                   "(mean over the launches of the run). FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE counts "
                   "half the bytes of a wide coalesced streaming read on gfx950, so hbm_read_bytes_corrected = 2 * FETCH_SIZE * 1024 "
                   "for the stage-2 kernel (16-byte-per-lane reads); the stage-1 kernel reads per-lane dwords (uncalibrated width), "
                   "both readings are given.",
           "kernels": {}}
for pat in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_TCC_HIT_sum_TC"):
    for k, d in per_kernel(pat).items():
        traffic["kernels"].setdefault(k, {}).update(d)
for k, d in traffic["kernels"].items():
    if not all(c in d for c in ("FETCH_SIZE", "WRITE_SIZE")):
        continue
    d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    d["hbm_read_bytes_uncorrected"] = d["FETCH_SIZE"] * 1024
    d["hbm_read_bytes_corrected_x2"] = 2 * d["FETCH_SIZE"] * 1024
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    d["traffic_bytes_per_launch"] = d["hbm_read_bytes_corrected_x2"] + d["hbm_write_bytes"]
(dst / f"{tag}_hbm_traffic_batch1024.json").write_text(json.dumps(traffic, indent=1) + "\n")

sq = {"sources_sha16": sources_sha16(), "note": "SQ counters per kernel launch (mean), same command as the traffic file; SQ_* are summed over all SEs/CUs.",
      "kernels": {}}
for pat in ("pmc_SQ_WAVE_CYCLES", "pmc_SQ_LDS_BANK_CO"):
    for k, d in per_kernel(pat).items():
        sq["kernels"].setdefault(k, {}).update(d)
for k, d in sq["kernels"].items():
    if d.get("SQ_INSTS_VALU"):
        d["valu_cycles_per_inst"] = d.get("SQ_ACTIVE_INST_VALU", 0) / d["SQ_INSTS_VALU"]
        d["salu_per_valu"] = d.get("SQ_INSTS_SALU", 0) / d["SQ_INSTS_VALU"]
(dst / f"{tag}_sq_counters_batch1024.json").write_text(json.dumps(sq, indent=1) + "\n")
print(json.dumps({k: {c: round(v, 4) if isinstance(v, float) else v for c, v in d.items()} for k, d in traffic["kernels"].items()}, indent=1))
