"""Does WHERE the output (and the coefficient store) lies change the step?  Round 6 noticed, in a sweep that re-created the plan and
its output tensor per experiment, that every other experiment ran 8-9 % slower whatever it was — the two output blocks torch's
allocator alternated between.  One plan, one process: the same execute into output buffers at different addresses.
    python tools/placement_probe.py [--batch 1024] [--reps 20]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=10)
    args = ap.parse_args()
    import torch
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.batch import prepare_batch
    from tools import synth
    dev = torch.device("cuda", 0)
    blob, offs = synth.synth_batch(args.distinct, 0, 1920, 1080, 85, "420", 120)
    raws = [blob[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(args.distinct)]
    files = [raws[i % args.distinct] for i in range(args.batch)]
    prep = prepare_batch(files, B.MJ_LAYOUT_XMAJOR, 0)
    ctx = B.Context(0)
    d_blob = torch.from_numpy(prep.blob).to(dev)
    if not args.fused:
        B.set_option("MJ_FUSED", "0")
    plan = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": args.batch})
    n = plan.info.rgb_bytes
    stream = torch.cuda.current_stream().cuda_stream
    bufs = plan.device_buffers()
    print(f"coefficient store at {bufs['coef']:#x} (mod 2 MiB {bufs['coef'] % (2 << 20):#x}, mod 1 GiB {bufs['coef'] % (1 << 30):#x}); output {n} bytes")

    def timed(ptr):
        for _ in range(6):
            plan.execute(stream, ptr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            plan.execute(stream, ptr)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps * 1e3
    # warm the clocks
    big = torch.empty(n + (256 << 20), dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        plan.execute(stream, big.data_ptr())
    torch.cuda.synchronize()
    base = big.data_ptr()
    print(f"one block at {base:#x} (mod 2 MiB {base % (2 << 20):#x}, mod 1 GiB {base % (1 << 30):#x})")
    for off in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, 32 << 20, 64 << 20, 128 << 20, (128 << 20) + (1 << 20), 0):
        print(f"   output at block + {off:>11d}: {timed(base + off):.3f} ms per step", flush=True)
    # separate blocks, as an allocator would hand them out
    blocks = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(args.blocks)]
    for rnd in range(2):
        for i, b in enumerate(blocks):
            p = b.data_ptr()
            print(f"   block {i} at {p:#x} (mod 1 GiB {p % (1 << 30):#x}, distance to the coefficient store {(p - bufs['coef']) / (1 << 30):+.3f} GiB): {timed(p):.3f} ms per step", flush=True)
    # the plain copy between pairs of blocks (mj_device_copy_rate uses the context's own buffers: here torch's copy kernel, for the pattern only)
    for i, j in ((0, 1), (0, 2), (0, 5), (0, 9), (4, 5), (2, 7), (8, 9)):
        if j >= len(blocks):
            continue
        a, b = blocks[i][:1 << 31].view(torch.float32), blocks[j][:1 << 31].view(torch.float32)
        for _ in range(3):
            b.copy_(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            b.copy_(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"   copy 2 GiB block {i} -> block {j}: {2 * (1 << 31) / dt / 1e12:.2f} TB/s", flush=True)
    # a second plan: its coefficient store is another allocation
    plan2 = B.Plan(ctx, prep.to_c(d_blob.data_ptr()), {"prep": prep, "n_images": args.batch})
    b2 = plan2.device_buffers()
    print(f"second plan: coefficient store at {b2['coef']:#x}")
    plan_keep, plan = plan, plan2
    for i, b in enumerate(blocks):
        print(f"   second plan, block {i}: {timed(b.data_ptr()):.3f} ms per step", flush=True)
    plan2.close()
    plan_keep.close()
    ctx.close()


if __name__ == "__main__":
    main()
