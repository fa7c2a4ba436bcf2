"""The N>1 path of bench.py on CPU: two gloo ranks shard a list of images, decode their shard's headers
(host logic only — no GPU), and aggregate pixel counts and the MAX of their timings."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_names, load_golden
from pyjpegdecoder_amd.sharding import max_over_ranks, shard, sum_over_ranks


def test_shard_partitions_exactly():
    for n in (0, 1, 7, 8, 1024, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyjpegdecoder_amd import parse_jpeg
    names = golden_names()
    lo, hi = shard(len(names), rank, world)
    pixels = 0
    for n in names[lo:hi]:
        p = parse_jpeg(load_golden(n)[0])
        pixels += p.image_width * p.image_height
    total = sum_over_ranks(float(pixels))
    slowest = max_over_ranks(1.0 + rank)            # rank r "took" 1+r seconds
    dist.barrier()
    q.put((rank, lo, hi, pixels, total, slowest))
    dist.destroy_process_group()


def test_two_gloo_ranks_shard_and_aggregate():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = golden_names()
    (r0, lo0, hi0, px0, tot0, slow0), (r1, lo1, hi1, px1, tot1, slow1) = out
    assert (lo0, hi1) == (0, len(names)) and hi0 == lo1            # disjoint, complete
    assert tot0 == tot1 == px0 + px1                               # job-wide aggregate on every rank
    assert slow0 == slow1 == 2.0                                   # MAX over ranks


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher around it starts the N ranks itself (torch.distributed.run as a child of a
    process that never touched the GPU) and hands rank 0's stdout through — here with a stand-in script for the rank body."""
    import json
    import subprocess
    import sys
    import bench
    script = tmp_path / "rank_body.py"
    script.write_text("import json, os, sys\n"
                      "import torch.distributed as dist\n"
                      "dist.init_process_group('gloo')\n"
                      "dist.barrier()\n"
                      "if os.environ['RANK'] == '0':\n"
                      "    print(json.dumps({'world': int(os.environ['WORLD_SIZE']), 'argv': sys.argv[1:],\n"
                      "                      'addr': os.environ['MASTER_ADDR']}), flush=True)\n"
                      "dist.destroy_process_group()\n")
    code = ("import sys, bench; sys.exit(bench.launch_ranks(2, ['--gpus', '2', '--steps', '3'], True, script=%r))" % str(script))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(bench.ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"world": 2, "argv": ["--gpus", "2", "--steps", "3"], "addr": "127.0.0.1"}


def test_bench_refuses_more_ranks_than_gpus(monkeypatch):
    import subprocess
    import sys
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus() == 1
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = subprocess.run([sys.executable, str(bench.ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=120, cwd=str(bench.ROOT))
    assert out.returncode == 2 and "--share-gpu" in out.stderr and out.stdout.strip() == ""
