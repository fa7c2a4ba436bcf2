import io, sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from tools import synth
from oracle import oracle
from pyjpegdecoder_amd import BatchDecoder, _binding as B
from pyjpegdecoder_amd.batch import prepare_batch
raws = []
for i in range(4):
    b = io.BytesIO(); Image.fromarray(synth.synth_rgb(i, 1920, 1080)).save(b, "JPEG", quality=85, subsampling=2, progressive=True); raws.append(b.getvalue())
print("sizes", [len(r) for r in raws])
dec = BatchDecoder(0)
t = time.time(); ref = oracle.decode(raws[0]); print("oracle prog 1080p s:", time.time() - t)
imgs = dec.decode(raws)
print("match", np.array_equal(imgs[0], ref["rgb"]))
for nb in [int(a) for a in sys.argv[1:]] or [256]:
    files = [raws[i % 4] for i in range(nb)]
    prep = prepare_batch(files)
    plan = B.Plan(dec.ctx, prep.to_c(), {"prep": prep, "n_images": len(files)})
    plan.execute(); plan.sync()
    t = time.time()
    for _ in range(3): plan.execute()
    plan.sync(); dt = (time.time() - t) / 3
    print("%d x 1080p progressive: %.1f ms/batch = %.0f MP/s" % (nb, dt * 1e3, nb * 2.0736 / dt))
    s1, s2 = plan.time_stages(2)
    print("stage1 ms", s1, "stage2 ms", s2)
    plan.close()
