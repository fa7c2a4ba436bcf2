import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools import synth
from oracle import oracle
from pyjpegdecoder_amd import BatchDecoder, _binding as B
blob, offs = synth.synth_batch(256, 0, 1920, 1080, 85, "420", 120)
raw = blob[int(offs[248]):int(offs[249])].tobytes()
ref = oracle.decode(raw)
dec = BatchDecoder(device=0, segment="host")
for form in (None, "wave", "lanes", "lanes11", "sync"):
    B.set_option("MJ_HUFFMAN", form)
    (img,), (seam,) = dec.decode([raw], return_seams=True)
    cd = np.argwhere(seam["coef"] != ref["coef"])
    print(form, "rgb ok", np.array_equal(img, ref["rgb"]), "coef diffs", len(cd), cd[:5].tolist())
    if len(cd):
        b = cd[0][0]
        print("  block", b, "mcu", b // 6, "row", (b // 6) // 120, "gpu", seam["coef"][b][:8].tolist(), "ref", ref["coef"][b][:8].tolist())
