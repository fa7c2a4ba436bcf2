"""Per-GPU image queue (BASELINE configs[3]): one rank's share of a job, decoded plan after plan with a few
plans in flight.

The reference decodes one file per ``JpegDecoder(path)`` call (jpeg_decoder.py:56-110); a job of thousands of files on a
node of GPUs is sharded by image (``sharding.shard``) and each rank feeds ITS GPU from this queue — no collective on the
data path.  The share's files are assembled and uploaded once (inputs resident in HBM), cut into batches of
``batch_size`` images; ``run()`` makes one pass: plan k is created and queued on stream ``k % depth`` into output slot
``k % depth`` — the plan is created first (host work beside the kernels in flight), then, before slot ``k % depth`` is used
again, plan ``k - depth`` is collected (synchronised, its per-image status read, destroyed: where a consumer takes the
pixels from HBM).  Statuses are kept per batch (``statuses``): images whose synchronisation rounds had not settled are
decoded again through the serial walk, an internal error raises, corrupt files are counted and named.

Why a queue and not one plan per batch back to back: stage 1 lasts as long as ONE restart segment's serial walk however
few segments a plan holds, so a plan of a few hundred images leaves most of the chip idle, and plan creation
(descriptor uploads, buffer clears) is host work that would otherwise sit between two plans' kernels.

Everything that touches the device goes through a small backend object (default: torch for memory and streams,
``_binding.Plan`` for the plans), so that the slot / collection logic is testable without a GPU (tests/test_queue.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence


class TorchBackend:
    """Device memory and streams from torch, plans from libmijpeg.so (through ``_binding``)."""

    def __init__(self, ctx, device=None):
        import torch
        from . import _binding as B
        self.torch, self.B, self.ctx = torch, B, ctx
        self.dev = torch.device("cuda", ctx.device if device is None else device)

    def upload(self, host_array):
        t = self.torch.from_numpy(host_array).to(self.dev)
        return t, t.data_ptr()

    def empty(self, nbytes: int):
        t = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.dev)
        return t, t.data_ptr()

    def stream(self):
        s = self.torch.cuda.Stream(device=self.dev)
        return s, s.cuda_stream

    def make_plan(self, prep, blob_ptr: int, n_images: int):
        return self.B.Plan(self.ctx, prep.to_c(blob_ptr), {"prep": prep, "n_images": n_images})

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)


class DeviceImageQueue:
    """``files``: this rank's share (a list of file bytes).  ``depth`` plans are in flight at once, each on its own stream
    and with its own output buffer.

    ``tune_placement=n`` (n > 1): the first plan that goes to each output slot tries n coefficient stores against it and keeps
    the fastest (``placement``); later plans of the slot inherit it through the context's buffer cache.

    ``across_passes=True`` keeps ``depth`` slots even when the share is fewer batches than that, so that with
    ``run(wait=False)`` the plan of the next pass (the next job's files, in a service) is created — host work — while the
    current pass's kernels run; ``drain()`` collects what is still in flight."""

    def __init__(self, ctx, files: Sequence[bytes], batch_size: int, layout: int, depth: int = 2, device=None, backend=None,
                 prepare: Optional[Callable] = None, on_collect: Optional[Callable] = None, across_passes: bool = False,
                 tune_placement: int = 0, collect_first: bool = False):
        if batch_size < 1 or depth < 1:
            raise ValueError("batch_size and depth must be at least 1")
        if prepare is None:
            from .batch import prepare_batch as prepare
        self.backend = backend if backend is not None else TorchBackend(ctx, device)
        self.layout = layout
        self._prepare = prepare
        self.batches = []                 # (prep, device blob handle, device blob pointer, images)
        self._files = []                  # per batch, its files (only looked at again when a batch has to be decoded another way)
        for i in range(0, len(files), batch_size):
            part = files[i:i + batch_size]
            prep = prepare(part, layout, 0)
            handle, ptr = self.backend.upload(prep.blob)
            self.batches.append((prep, handle, ptr, len(part)))
            self._files.append(part)
        cap = max((sum(w * h * nc for (w, h, nc) in b[0].shapes) for b in self.batches), default=0)
        self.depth = max(1, depth if across_passes else min(depth, len(self.batches)))
        self._slot_bytes = cap
        self.out = [self.backend.empty(cap) for _ in range(self.depth)]           # (handle, pointer) per slot
        self.streams = [self.backend.stream() for _ in range(self.depth)]        # (handle, raw stream) per slot
        self.n_images = len(files)
        self.bad = 0                      # plans with an image whose status was not MJ_ST_OK, over every run() so far
        self.statuses = {}                # batch number -> per-image status array of its latest decode (MJ_ST_*; all zero = fine)
        self.collected: List[int] = []    # batch numbers in the order they were collected (the latest run())
        # called as on_collect(batch number, slot, plan) when a plan's pixels are in HBM, before the plan is destroyed;
        # statuses[batch number] then says which of its images (if any) are not valid
        self.on_collect = on_collect
        # Placement (mj_plan_tune_placement): where a plan's coefficient store lies relative to its output slot decides, by the
        # luck of two allocations, whether its fused launch runs 8-9 % slower.  The context recycles the stores from plan to plan
        # (three of them go round two slots: a slot's next plan is created before its previous one is destroyed), so the FIRST plan
        # that goes to a slot tries `tune_placement` stores — and up to three buffers for the slot itself — (a few timed decodes
        # each, once: placement.tuned_output); the losers go back to the device, so what keeps going round are stores that ran
        # well against one of the slots.
        self.tune_placement = tune_placement
        # collect_first (an A/B switch, off): collect — and destroy — the plan whose slot the next plan takes BEFORE that plan is
        # created, so that the new plan takes over exactly its buffers (the context's cache hands out the most recently released
        # block of a size) and the tuned pairing of coefficient store and output slot stays for good.  Created first (the default),
        # a slot's new plan gets the store the OTHER slot's plan released and three stores go round two slots.  Measured, 1 250
        # images per pass, same box, twice each: 7.43-7.50 ms per pass collected first against 7.20-7.24 created first (tuned
        # executes 7.0-7.05 ms by HIP events either way): the fixed pairing buys nothing and the later launch costs 0.25 ms.
        self.collect_first = collect_first
        self._slot_tuned = [False] * self.depth
        self.placement = {}               # slot -> placement.tuned_output's report (output buffers and stores tried, what stayed)
        self._flying = []                 # (batch number, plan), oldest first — across run() calls with wait=False
        self._seq = 0                     # plans submitted since the queue last ran dry: plan number `seq` uses slot seq % depth
        self._slot = {}                   # batch number -> slot of its latest submission

    def slot_of(self, k: int) -> int:
        """The slot batch k's latest decode went to (k % depth for a pass that started with nothing in flight)."""
        return self._slot.get(k, k % self.depth)

    def out_tensor(self, slot: int):
        """The output buffer of a slot (the backend's handle: a ``torch.uint8`` tensor with the default backend)."""
        return self.out[slot][0]

    def _again_without_sync(self, k: int):
        """Batch k once more through the serial walk (MJ_FLAG_NO_SYNC), into the same slot: some image's synchronisation
        rounds had not settled (files without restart markers; BatchDecoder does the same per image).  Returns the statuses."""
        from . import _binding as B
        prep = self._prepare(self._files[k], self.layout, B.MJ_FLAG_NO_SYNC)
        handle, ptr = self.backend.upload(prep.blob)
        plan = self.backend.make_plan(prep, ptr, len(self._files[k]))
        try:
            slot = self.slot_of(k)
            plan.execute(self.streams[slot][1], self.out[slot][1])
            plan.sync()
            return plan.read(rgb=False)["status"]
        finally:
            plan.close()
            del handle

    def _collect(self, k: int, plan):
        try:
            plan.sync()
            status = plan.read(rgb=False)["status"]
            if status.any():
                from . import _binding as B
                if (status == B.MJ_ST_UNCONVERGED).any():
                    status = self._again_without_sync(k)
                if (status == B.MJ_ST_INTERNAL).any():          # not the files' fault: nothing of this batch can be trusted
                    from .errors import BackendError
                    raise BackendError(f"batch {k}: a fused launch gave up waiting for its decoder wavefronts (internal error)")
            self.statuses[k] = status
            self.bad += int(status.any())
            self.collected.append(k)
            if self.on_collect is not None:
                self.on_collect(k, self.slot_of(k), plan)
        finally:
            plan.close()

    def run(self, first: int = 0, count: Optional[int] = None, wait: bool = True):
        """One pass over the share (or over ``count`` batches from batch ``first``).  ``wait=True``: returns when every
        batch's pixels are in HBM; batch k's pixels are in ``out_tensor(slot_of(k))`` until ``depth`` more plans have been
        submitted.  ``wait=False``: returns with up to ``depth`` plans still in flight — the next ``run()`` goes on from
        there (its first plan is created while they run), ``drain()`` collects them."""
        if not self._flying:
            self.collected = []
            self._seq = first
        last = len(self.batches) if count is None else min(len(self.batches), first + count)
        try:
            for k in range(first, last):
                prep, _, ptr, n = self.batches[k]
                # if the slot is still in use by the plan submitted `depth` plans ago: that plan's collection first (its buffers go
                # to the plan created next: collect_first above), then the new plan (host work, beside the kernels in flight), then
                # the launch
                if self.collect_first and len(self._flying) == self.depth:
                    self._collect(*self._flying.pop(0))
                plan = self.backend.make_plan(prep, ptr, n)
                try:
                    if len(self._flying) == self.depth:
                        self._collect(*self._flying.pop(0))
                except BaseException:
                    plan.close()
                    raise
                slot = self._seq % self.depth
                self._seq += 1
                self._slot[k] = slot
                self._flying.append((k, plan))
                if self.tune_placement > 1 and not self._slot_tuned[slot] and hasattr(plan, "tune_placement"):
                    self._slot_tuned[slot] = True
                    from .placement import tuned_output
                    have = [self.out[slot]]

                    def alloc(nbytes, have=have):           # the slot's own buffer first, then others of its size
                        return have.pop() if have else self.backend.empty(self._slot_bytes)
                    handle, ptr, report = tuned_output(plan, self.streams[slot][1], self._slot_bytes, alloc, out_candidates=3,
                                                       store_candidates=self.tune_placement)
                    self.out[slot] = (handle, ptr)
                    self.placement[slot] = report
                plan.execute(self.streams[slot][1], self.out[slot][1])
            if wait:
                self.drain()
        except BaseException:
            for _, plan in self._flying:  # (an exception on the way: nothing of this pass stays alive)
                plan.close()
            self._flying = []
            raise

    def drain(self):
        """Collect every plan still in flight (after ``run(wait=False)``)."""
        try:
            while self._flying:
                self._collect(*self._flying.pop(0))
        except BaseException:
            for _, plan in self._flying:
                plan.close()
            self._flying = []
            raise
