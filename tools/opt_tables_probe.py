"""Batches of files with per-file optimised Huffman tables (Pillow optimize=True): which stage-1 form the library picks
and what it costs, against the same pictures with the standard tables.  Run on the GPU box."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from tools import synth
from pyjpegdecoder_amd import BatchDecoder, _binding as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1920, 1080
NAMES = {0: "wave", 1: "lanes", 2: "sync", 3: "scans"}
for label, kw in (("standard tables, DRI", dict(restart_marker_rows=1)), ("optimised tables, DRI", dict(optimize=True, restart_marker_rows=1)),
                  ("standard tables, no DRI", dict()), ("optimised tables, no DRI", dict(optimize=True))):
    distinct = []
    for i in range(32):
        b = io.BytesIO(); Image.fromarray(synth.synth_rgb(i, W, H)).save(b, "JPEG", quality=85, subsampling=2, **kw); distinct.append(b.getvalue())
    files = [distinct[i % 32] for i in range(n)]
    for force in (None, "wave"):
        if force: B.set_option("MJ_HUFFMAN", force)
        else: B.set_option("MJ_HUFFMAN", None)
        dec = BatchDecoder(0)
        t = time.time(); prep, plan = dec.plan(files); tc = time.time() - t
        plan.execute(); plan.sync()
        s1, s2 = plan.time_stages(3)
        f = plan.stage1_form()
        print("%-26s %-12s tables %4d  form %-5s%s  plan create %6.1f ms  stage 1 %7.2f ms  stage 2 %5.2f ms" % (
            label, "(forced wave)" if force else "", prep.n_huff, NAMES[f & 15], "+wg" if f & 16 else "   ", tc * 1e3, s1, s2))
        plan.close(); dec.close()
B.set_option("MJ_HUFFMAN", None)
# medium files: more images per workgroup
for size in ((1280, 720), (800, 600), (640, 480)):
    for kw, label in ((dict(optimize=True, restart_marker_rows=1), "optimised, DRI"), (dict(optimize=True), "optimised, no DRI")):
        distinct = []
        for i in range(32):
            b = io.BytesIO(); Image.fromarray(synth.synth_rgb(i, *size)).save(b, "JPEG", quality=85, subsampling=2, **kw); distinct.append(b.getvalue())
        files = [distinct[i % 32] for i in range(1024)]
        for force in (None, "wave"):
            if force: B.set_option("MJ_HUFFMAN", force)
            else: B.set_option("MJ_HUFFMAN", None)
            dec = BatchDecoder(0)
            prep, plan = dec.plan(files)
            plan.execute(); plan.sync()
            s1, s2 = plan.time_stages(3)
            f = plan.stage1_form()
            print("1024 x %dx%d %-18s %-13s avg file %4d KB  form %-5s%s  stage 1 %7.2f ms  stage 2 %5.2f ms" % (
                size[0], size[1], label, "(forced wave)" if force else "", sum(map(len, distinct)) // 32 // 1024, NAMES[f & 15], "+wg" if f & 16 else "   ", s1, s2))
            plan.close(); dec.close()
B.set_option("MJ_HUFFMAN", None)
