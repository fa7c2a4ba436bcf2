import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_index():
    return json.loads((GOLDEN / "files" / "index.json").read_text())


def golden_names(small_only=True):
    idx = golden_index()
    return sorted(n for n, m in idx.items() if not (small_only and "sampled" in m) and not n.startswith("prog_"))


@pytest.fixture(scope="session")
def gindex():
    return golden_index()


def load_golden(name):
    raw = (GOLDEN / "files" / f"{name}.jpg").read_bytes()
    vec = np.load(GOLDEN / "files" / f"{name}.npz")
    return raw, vec


@pytest.fixture
def tune():
    """Library switches (mj_set_option: stage-1 form, segment order, chunk sizes ...) for one test; back to the defaults afterwards."""
    from pyjpegdecoder_amd import _binding as B
    touched = []

    def _set(name, value):
        B.set_option(name, value)
        touched.append(name)
    yield _set
    for name in touched:
        B.set_option(name, None)


def oracle_rgb_all(raws, workers=None):
    """The oracle's pixels of every file in `raws`, on a pool of threads (the C oracle runs outside the GIL): what lets the
    at-size tests hold EVERY distinct image of a 1024-image batch to the oracle instead of a sample."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    if not raws:
        return []
    first = oracle.decode(raws[0])["rgb"]                     # (builds the oracle's tables once, before the threads start)
    workers = workers or max(1, min(16, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(workers) as pool:
        return [first] + list(pool.map(lambda r: oracle.decode(r)["rgb"], raws[1:]))
