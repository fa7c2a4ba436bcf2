"""Where does JpegDecoder(path) on ONE 1080p file spend its time?  (GPU box)  python tools/single_file_probe.py [ri]"""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
from tools import synth
from pyjpegdecoder_amd import JpegDecoder

ri = int(sys.argv[1]) if len(sys.argv) > 1 else 120
raw = synth.synth_jpeg(1, 1920, 1080, 85, "420", ri, 12.0)
with tempfile.TemporaryDirectory() as td:
    path = Path(td) / "one.jpg"
    path.write_bytes(raw)
    JpegDecoder(path)
    JpegDecoder(path)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        JpegDecoder(path)
        ts.append(time.perf_counter() - t0)
    print(f"ri={ri}: median {sorted(ts)[7] * 1e3:.2f} ms, min {min(ts) * 1e3:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        JpegDecoder(path)
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
    print("\n".join(l[:150] for l in out.getvalue().splitlines()[4:48]))
