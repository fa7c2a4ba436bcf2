"""Placement tuning for plans that are executed many times into one output buffer (a decode service's output slot, a benchmark's
step).

On MI355X the fused launch (stages 1 and 2 in one launch: ``MJ_FORM_FUSED``) runs in one of two classes 5-9 % apart depending on
where — physically — the plan's coefficient store, its stage-0 stream buffer and the output buffer lie relative to each other
(``profiles/r06_placement.txt``).  Nothing in the virtual addresses says which; a few timed executes do.  ``mj_plan_tune_placement``
picks the plan's own buffers against a given output buffer; a caller that also owns the output can try a few of those too — this
helper does both and returns the output buffer to use.
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def tuned_output(plan, stream: int, nbytes: int, alloc: Callable[[int], Tuple[object, int]], out_candidates: int = 3,
                 store_candidates: int = 4):
    """Allocate up to ``out_candidates`` output buffers with ``alloc(nbytes) -> (handle, device pointer)``, let the plan pick its
    stores against each (``store_candidates`` each), keep the pair with the fastest execute.  Returns (handle, pointer, report);
    the other buffers are dropped.  ``report``: per output candidate the ms per execute the plan reached, and which was kept.
    A plan that is not fused is left alone (one buffer is allocated and returned)."""
    tried: List[Tuple[object, int, float, list]] = []
    best = -1
    for _ in range(max(1, out_candidates)):
        handle, ptr = alloc(nbytes)
        ms_list, _ = plan.tune_placement(stream, ptr, store_candidates)
        ms = float(getattr(plan, "best_ms", 0.0))
        tried.append((handle, ptr, ms, [round(x, 3) for x in ms_list if x > 0]))
        if ms <= 0.0:                 # not a fused plan: nothing to choose between
            best = 0
            break
        if best < 0 or ms < tried[best][2] * 0.99:
            best = len(tried) - 1
    if len(tried) > 1 and best != len(tried) - 1:
        # the plan's buffers were last picked against the last candidate: once more against the one that stays
        plan.tune_placement(stream, tried[best][1], store_candidates)
    report = {"output_candidates_ms_per_execute": [round(t[2], 3) for t in tried], "kept": best,
              "stores_tried_per_output_ms": [t[3] for t in tried],
              "ms_per_execute": round(float(getattr(plan, "best_ms", 0.0)), 3)}
    handle, ptr = tried[best][0], tried[best][1]
    tried.clear()                     # (the losers' handles go with it)
    return handle, ptr, report
