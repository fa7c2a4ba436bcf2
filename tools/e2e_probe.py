"""Public-API wall time, host bytes in -> device pixels out (BatchDecoder.decode_device), 512 x 1080p files:
host segmentation / GPU segmentation with the Python host code / GPU segmentation with the native host front end.
Run on the GPU box:  python tools/e2e_probe.py [--profile]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import synth
from pyjpegdecoder_amd import BatchDecoder
if os.environ.get("MJ_PROBE_LIB"):      # A/B against another build of the library
    from pathlib import Path
    from pyjpegdecoder_amd import _binding as _B
    _B.LIB_PATH = Path(os.environ["MJ_PROBE_LIB"]).resolve()
W, H = 1920, 1080
blob, offs = synth.synth_batch(64, 0, W, H, 85, "420", 120)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(64)]
files = [raws[i % 64] for i in range(512)]
for label, kw in (("host segmentation, Python host code", dict(segment="host")),
                  ("GPU segmentation, Python host code", dict(segment="gpu", native_host=False)),
                  ("GPU segmentation, native host front end", dict(segment="gpu", native_host=True))):
    dec = BatchDecoder(0, **kw)
    dec.decode_device(files[:8])
    dec.decode_device(files)                 # steady state of a serving loop: staging buffer and allocator caches warm
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); out = dec.decode_device(files); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("%-42s decode_device 512 x 1080p: %.3f s = %.0f MP/s" % (label, best, 512 * W * H / 1e6 / best))
    if "--profile" in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); dec.decode_device(files); torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    dec.close()

# one call, as one plan or in overlapped parts (round 6: the default from 512 files on), 512 and 1024 files of 256 distinct ones
blob2, offs2 = synth.synth_batch(256, 1000, W, H, 85, "420", 120)
raws2 = [blob2[int(offs2[i]):int(offs2[i + 1])].tobytes() for i in range(256)]
dec = BatchDecoder(0, segment="gpu")
for n in (512, 1024):
    fs = [raws2[i % 256] for i in range(n)]
    for parts in (1, 2, 4, None):
        dec.decode_device(fs, parts=parts)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t = time.perf_counter(); out = dec.decode_device(fs, parts=parts); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        del out
        ts.sort()
        print("one call, %4d files, parts=%-4s  median %.4f s (min %.4f) = %.0f MP/s" % (n, parts, ts[2], ts[0], n * W * H / 1e6 / ts[2]))
dec.close()

# a serving loop: batches back to back, host work and upload of the next batch under the GPU work of the current one
dec = BatchDecoder(0, segment="gpu")
for _ in dec.decode_device_iter([files[:8], files, files]): pass
torch.cuda.synchronize()
nb = 24
for depth in (1, 2, 3):          # batches in flight (1: round 5's pipeline — the host waited for batch k before it assembled k + 2)
    for _ in dec.decode_device_iter((files for _ in range(3)), depth=depth): pass
    torch.cuda.synchronize()
    t = time.perf_counter()
    for out in dec.decode_device_iter((files for _ in range(nb)), depth=depth): pass
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / nb
    print("%-42s decode_device_iter, %d batches of 512 x 1080p, %d in flight: %.4f s per batch = %.0f MP/s" % ("pipelined, native host front end", nb, depth, dt, 512 * W * H / 1e6 / dt))
dec.close()
