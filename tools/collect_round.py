"""After `gpurun -- 'bash tools/final_round.sh <tag>'`: copy what the judge should see from gpurun_out/ (scratch) into profiles/
(tracked), under the names earlier rounds used.  tools/profile_summary.py <tag> does the rocprof part; this does the rest.
    python tools/collect_round.py r06e"""
import glob
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r06e"
src, dst = ROOT / "gpurun_out", ROOT / "profiles"
names = {
    f"config4_1gpu_share_{tag}.json": f"{tag}_config4_1gpu_1250_images.json",
    f"config4_2rank_sharegpu_{tag}.json": f"{tag}_config4_2ranks_on_one_gpu_2500_images.json",
    f"bench_gpus2_sharegpu_{tag}.json": f"{tag}_bench_gpus2_sharegpu.json",
    "layout_sweep.txt": f"{tag}_layout_sweep.txt",
    f"mixed_orders_{tag}.txt": f"{tag}_mixed_content_segment_orders.txt",
    f"generic_layouts_{tag}.txt": f"{tag}_generic_layouts.txt",
    f"prog_sweep_{tag}.txt": f"{tag}_progressive_sweep.txt",
    f"e2e_{tag}.txt": f"{tag}_public_api_e2e.txt",
    f"nodri_stats_{tag}.txt": f"{tag}_nodri_kernel_stats.txt",
    f"nodri_sizes_{tag}.txt": f"{tag}_nodri_batch_sizes.txt",
    f"single_file_{tag}.txt": f"{tag}_single_file_probe.txt",
    f"step_probe_{tag}.txt": f"{tag}_step_probe.txt",
    f"stress_fused_{tag}.txt": f"{tag}_stress_fused.txt",
    f"stress_parity_{tag}.txt": f"{tag}_stress_parity.txt",
    f"stress_progressive_{tag}.txt": f"{tag}_stress_progressive.txt",
    f"stress_crafted_{tag}.txt": f"{tag}_stress_crafted.txt",
}
for a, b in names.items():
    f = src / a
    if f.exists() and f.stat().st_size:
        text = "\n".join(l for l in f.read_text(errors="replace").splitlines() if "amdgpu.ids" not in l) + "\n"
        (dst / b).write_text(text)
        print("profiles/" + b)
    else:
        print("missing:", a)
stats = sorted(glob.glob(str(src / f"prog_{tag}" / "stats" / "*" / "*kernel_stats.csv")), key=lambda f: -Path(f).stat().st_size)
if stats:
    shutil.copy(stats[0], dst / f"{tag}_progressive_kernel_stats.csv")
    print(f"profiles/{tag}_progressive_kernel_stats.csv")
