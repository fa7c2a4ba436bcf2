import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools import synth
from pyjpegdecoder_amd import BatchDecoder
W, H = 1920, 1080
blob, offs = synth.synth_batch(64, 0, W, H, 85, "420", 120)
raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(64)]
files = [raws[i % 64] for i in range(512)]
for seg in ("host", "gpu"):
    dec = BatchDecoder(0, segment=seg)
    dec.decode_device(files[:8])
    torch.cuda.synchronize()
    t = time.perf_counter(); out = dec.decode_device(files); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(seg, "decode_device 512 x 1080p: %.3f s = %.0f MP/s" % (dt, 512 * W * H / 1e6 / dt))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); dec.decode_device(files); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
    dec.close()
