"""The per-GPU image queue of BASELINE configs[3] (pyjpegdecoder_amd/queue.py) without a GPU: slot reuse, collection order,
status accounting and clean-up, with a stand-in for the device (memory, streams, plans)."""
import numpy as np
import pytest

from pyjpegdecoder_amd.queue import DeviceImageQueue


class _Prep:
    def __init__(self, files):
        self.blob = np.frombuffer(b"".join(files), dtype=np.uint8).copy()
        self.shapes = [(16, 8, 3)] * len(files)


class _Plan:
    def __init__(self, log, k, n, bad=False, fail_execute=False, code=2):
        self.log, self.k, self.n, self.bad, self.fail_execute, self.code = log, k, n, bad, fail_execute, code
        self.closed = False

    def execute(self, stream, out):
        if self.fail_execute:
            raise RuntimeError("launch failed")
        self.log.append(("execute", self.k, stream, out))

    def sync(self):
        self.log.append(("sync", self.k))

    def read(self, rgb=True):
        st = np.zeros(self.n, dtype=np.int32)
        if self.bad:
            st[-1] = self.code
        return {"status": st}

    def close(self):
        if not self.closed:
            self.log.append(("close", self.k))
        self.closed = True


class _Backend:
    def __init__(self, bad=(), fail=(), code=2):
        self.log, self.bad, self.fail, self.code = [], set(bad), set(fail), code
        self.n_up = self.n_out = self.n_streams = self.made = 0
        self.plans = []

    def upload(self, a):
        self.n_up += 1
        return a, 1000 + self.n_up

    def empty(self, nbytes):
        self.n_out += 1
        self.cap = nbytes
        return ("out", self.n_out), 5000 + self.n_out

    def stream(self):
        self.n_streams += 1
        return ("stream", self.n_streams), 70 + self.n_streams

    def make_plan(self, prep, ptr, n):
        k = ptr - 1001                      # the batch whose blob this is (a blob uploaded later: a batch decoded once more)
        p = _Plan(self.log, k, n, bad=k in self.bad, fail_execute=k in self.fail, code=self.code)
        self.plans.append(p)
        return p


def _prepare(files, layout, flags):
    return _Prep(files)


def _files(n):
    return [bytes([i % 251]) * 5 for i in range(n)]


def test_slots_are_reused_only_after_their_plan_was_collected():
    be = _Backend()
    q = DeviceImageQueue(None, _files(23), 4, 0, depth=3, backend=be, prepare=_prepare)
    assert len(q.batches) == 6 and [b[3] for b in q.batches] == [4, 4, 4, 4, 4, 3]
    assert be.n_up == 6 and be.n_out == 3 and be.n_streams == 3 and be.cap == 4 * 16 * 8 * 3
    q.run()
    assert q.collected == [0, 1, 2, 3, 4, 5] and q.bad == 0
    ex = {e[1]: i for i, e in enumerate(be.log) if e[0] == "execute"}
    cl = {e[1]: i for i, e in enumerate(be.log) if e[0] == "close"}
    sy = {e[1]: i for i, e in enumerate(be.log) if e[0] == "sync"}
    for k in range(6):
        _, _, stream, out = be.log[ex[k]]
        assert (stream, out) == (71 + k % 3, 5001 + k % 3)          # plan k: stream and output slot k % depth
        assert sy[k] < cl[k]
        if k >= 3:
            assert cl[k - 3] < ex[k]                                # slot k % depth is free before plan k is queued
    # depth plans are in flight: plan 2 is queued before plan 0 is collected
    assert ex[2] < sy[0]
    assert all(p.closed for p in be.plans)


def test_depth_is_capped_by_the_number_of_batches_and_partial_runs():
    be = _Backend()
    q = DeviceImageQueue(None, _files(8), 4, 0, depth=5, backend=be, prepare=_prepare)
    assert q.depth == 2 and be.n_out == 2
    q.run(first=1, count=1)
    assert q.collected == [1]
    assert [e for e in be.log if e[0] == "execute"] == [("execute", 1, 72, 5002)]


def test_status_counting_callbacks_and_cleanup_after_a_failure():
    seen = []
    be = _Backend(bad={1, 3})
    q = DeviceImageQueue(None, _files(16), 4, 0, depth=2, backend=be, prepare=_prepare,
                         on_collect=lambda k, slot, plan: seen.append((k, slot, plan.closed)))
    q.run()
    assert q.bad == 2 and seen == [(0, 0, False), (1, 1, False), (2, 0, False), (3, 1, False)]
    q.run()
    assert q.bad == 4                                               # accumulates over runs
    be2 = _Backend(fail={2})
    q2 = DeviceImageQueue(None, _files(16), 4, 0, depth=2, backend=be2, prepare=_prepare)
    with pytest.raises(RuntimeError):
        q2.run()
    assert all(p.closed for p in be2.plans)                         # nothing of the pass stays alive
    with pytest.raises(ValueError):
        DeviceImageQueue(None, _files(4), 0, 0, backend=_Backend(), prepare=_prepare)


def test_passes_overlap_when_asked_to_and_drain_collects_the_rest():
    """across_passes + run(wait=False): the next pass's plan is created (host work) before the previous pass's plan is waited
    for — config 4's share is ONE 1 250-image plan per pass, whose creation would otherwise sit between two passes' kernels."""
    be = _Backend()
    q = DeviceImageQueue(None, _files(4), 4, 0, depth=2, backend=be, prepare=_prepare, across_passes=True)
    assert len(q.batches) == 1 and q.depth == 2 and be.n_out == 2
    for _ in range(3):
        q.run(wait=False)
    ex = [i for i, e in enumerate(be.log) if e[0] == "execute"]
    sy = [i for i, e in enumerate(be.log) if e[0] == "sync"]
    assert len(ex) == 3 and len(sy) == 1                           # two plans in flight, the first pass collected for its slot
    assert ex[1] < sy[0] < ex[2]                                   # pass 1 was queued before pass 0 was waited for
    assert [be.log[i][2:] for i in ex] == [(71, 5001), (72, 5002), (71, 5001)]     # slots in turn
    assert q.slot_of(0) == 0                                       # the latest decode of batch 0 went to slot 0
    assert len(be.plans) == 3 and [p.closed for p in be.plans] == [True, False, False]
    q.drain()
    assert all(p.closed for p in be.plans) and q.collected == [0, 0, 0] and q.bad == 0
    q.run()                                                        # (and a waiting pass afterwards starts from slot 0 again)
    assert be.log[-3][0] == "execute" and be.log[-3][2:] == (71, 5001) and all(p.closed for p in be.plans)


def test_statuses_are_kept_unsettled_images_go_round_again_and_internal_errors_raise():
    from pyjpegdecoder_amd import _binding as B
    from pyjpegdecoder_amd.errors import BackendError
    be = _Backend(bad={1})
    q = DeviceImageQueue(None, _files(8), 4, 0, depth=2, backend=be, prepare=_prepare)
    q.run()
    assert q.bad == 1 and not q.statuses[0].any() and q.statuses[1].tolist() == [0, 0, 0, 2]      # which image of which batch
    # MJ_ST_UNCONVERGED: the batch is decoded once more with MJ_FLAG_NO_SYNC into the same slot; what that decode says stands
    flags_seen = []

    def prepare(files, layout, flags):
        flags_seen.append(flags)
        return _Prep(files)
    be = _Backend(bad={1}, code=B.MJ_ST_UNCONVERGED)
    q = DeviceImageQueue(None, _files(8), 4, 0, depth=2, backend=be, prepare=prepare)
    q.run()
    assert flags_seen == [0, 0, B.MJ_FLAG_NO_SYNC]
    redo = be.plans[-1]                                             # (its blob was uploaded third: the stand-in calls it batch 2)
    assert redo.closed and ("execute", redo.k, 72, 5002) in be.log  # batch 1's stream and slot
    assert q.bad == 0 and not q.statuses[1].any()
    # MJ_ST_INTERNAL is not a property of a file: it raises, and nothing stays alive
    be = _Backend(bad={0}, code=B.MJ_ST_INTERNAL)
    q = DeviceImageQueue(None, _files(8), 4, 0, depth=2, backend=be, prepare=_prepare)
    with pytest.raises(BackendError):
        q.run()
    assert all(p.closed for p in be.plans)


def test_tuned_output_keeps_the_fastest_pair_and_retunes_for_it():
    """pyjpegdecoder_amd.placement.tuned_output without a GPU: output buffers are tried in turn, the plan picks its stores against
    each, the fastest pair stays — and when that is not the last one tried, the plan picks its stores once more against it."""
    from pyjpegdecoder_amd.placement import tuned_output

    class P:
        def __init__(self, ms_by_ptr):
            self.ms_by_ptr, self.calls, self.best_ms = ms_by_ptr, [], 0.0

        def tune_placement(self, stream, ptr, candidates):
            self.calls.append(ptr)
            self.best_ms = self.ms_by_ptr[ptr]
            return [self.best_ms + 0.3] + [self.best_ms] * (candidates - 1), 1
    n = [0]

    def alloc(nbytes):
        n[0] += 1
        return ("buf", n[0]), 100 + n[0]
    plan = P({101: 6.30, 102: 5.80, 103: 5.79})             # (103 is within one per cent of 102: the earlier one stays)
    handle, ptr, rep = tuned_output(plan, 7, 1 << 20, alloc, out_candidates=3, store_candidates=4)
    assert (handle, ptr) == (("buf", 2), 102) and rep["kept"] == 1 and rep["output_candidates_ms_per_execute"] == [6.3, 5.8, 5.79]
    assert plan.calls == [101, 102, 103, 102] and rep["ms_per_execute"] == 5.8          # re-tuned against the one that stays
    plan = P({101: 5.7})
    n[0] = 0
    _, ptr, rep = tuned_output(plan, 7, 1 << 20, alloc, out_candidates=1)
    assert ptr == 101 and plan.calls == [101] and rep["kept"] == 0
    plan = P({101: 0.0})                                     # not a fused plan: the first buffer, nothing else allocated
    m = [0]

    def alloc2(nbytes):
        m[0] += 1
        return None, 100 + m[0]
    _, ptr, rep = tuned_output(plan, 7, 64, alloc2, out_candidates=3)
    assert ptr == 101 and m[0] == 1 and rep["kept"] == 0
