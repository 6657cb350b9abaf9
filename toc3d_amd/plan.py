"""Where the launches of one frame go: straight onto HIP streams (eager) or into a recorded launch plan.

The backbone / neck host code names concurrency by *lane* (lane 0 = the caller's stream; the view groups and the scorer's
side work run on further lanes) and orders lanes with ``wait(a, b)`` = "lane a's next launch runs after everything issued so
far on lane b".  Two executors implement that vocabulary:

* :class:`EagerExec`  -- lanes are torch HIP streams, ``wait`` is an event record + stream wait; every C-ABI call launches at once.
* :class:`RecordExec` -- lanes are lanes of a ``toc3d_plan_t`` (``include/toc3d.h``); every C-ABI call made inside
  ``with exec.lane(i)`` is *recorded*, and the finished :class:`LaunchPlan` replays the whole frame with one C call per frame
  (``toc3d_plan_run``) -- the Python loop over blocks (the reference's ``toc3d_eva_vit.py:263-291``) runs once per configuration,
  not once per frame.
"""
from __future__ import annotations

import contextlib
from typing import List

import torch

from . import lib

MODES = {"eager": None, "plan": 0, "graph": 1}     # launch_mode -> toc3d_plan_end mode
MAX_LANES = 64                                     # csrc/plan.cpp MAX_LANES: frames that need more lanes launch eagerly


class EagerExec:
    def __init__(self, n_lanes: int, pool: List[torch.cuda.Stream]):
        while len(pool) < n_lanes - 1:
            pool.append(torch.cuda.Stream())
        self.streams = [torch.cuda.current_stream()] + pool[: n_lanes - 1]
        self.n_lanes = n_lanes

    def lane(self, i: int):
        return torch.cuda.stream(self.streams[i])

    def wait(self, waiting: int, on: int):
        if waiting == on:
            return
        ev = torch.cuda.Event()
        ev.record(self.streams[on])
        self.streams[waiting].wait_event(ev)


class LaunchPlan:
    """Owner of a ``toc3d_plan_t``."""

    def __init__(self):
        import ctypes
        h = ctypes.c_void_p()
        lib.load()
        lib.call("toc3d_plan_create", ctypes.addressof(h))
        self.handle = h.value
        self.keep = []                 # tensors the recorded launches point into

    def run(self):
        lib.call("toc3d_plan_run", self.handle, torch.cuda.current_stream().cuda_stream)

    @property
    def num_launches(self) -> int:
        return int(lib.load().toc3d_plan_num_launches(self.handle))

    # A plan's recorded launches carry baked device pointers (workspaces, packed weights) of the module that recorded it: a copy
    # of the handle would replay on the ORIGINAL module's buffers and free the C object twice.  Plans are therefore never copied
    # or pickled; the owning modules drop them in __deepcopy__ / __getstate__ and re-record on their next forward.
    def __deepcopy__(self, memo):
        raise TypeError("LaunchPlan is not copyable: it points into the buffers of the module that recorded it (copy the module; it re-records)")

    __copy__ = lambda self: LaunchPlan.__deepcopy__(self, None)     # noqa: E731

    def __reduce__(self):
        raise TypeError("LaunchPlan cannot be pickled: it owns a native toc3d_plan_t with baked device pointers")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.load().toc3d_plan_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class RecordExec:
    """Records into ``plan``; use as ``with RecordExec(plan, n_lanes, mode) as ex: ...`` around the frame's launch sequence."""

    def __init__(self, plan: LaunchPlan, n_lanes: int, mode: int):
        self.plan, self.n_lanes, self.mode = plan, n_lanes, mode

    def __enter__(self):
        if lib.recording():
            raise RuntimeError("a launch plan is already being recorded on this thread")
        lib.call("toc3d_plan_begin", self.plan.handle)
        lib.set_rec_lane(0)
        return self

    def __exit__(self, et, ev, tb):
        lib.set_rec_lane(None)
        if et is not None:
            # leave recording mode on the C side as well; the plan is unusable and the caller drops it
            try:
                lib.call("toc3d_plan_end", self.plan.handle, self.mode)
            except RuntimeError:
                pass
            return False
        for l in range(1, self.n_lanes):
            self.wait(0, l)                                   # a frame ends joined into lane 0
        lib.call("toc3d_plan_end", self.plan.handle, self.mode)
        return False

    @contextlib.contextmanager
    def lane(self, i: int):
        prev = lib.rec_lane()
        lib.set_rec_lane(i)
        try:
            yield
        finally:
            lib.set_rec_lane(prev)

    def wait(self, waiting: int, on: int):
        lib.call("toc3d_plan_wait", self.plan.handle, waiting, on)


def run_frame(state: dict, launch_mode: str, n_lanes: int, frame_fn, pool: List[torch.cuda.Stream]):
    """Run ``frame_fn(executor)`` -- the launch sequence of one frame -- in ``launch_mode``: eagerly on HIP streams, or recorded once
    into a launch plan (second call with this ``state``; the first one runs eagerly, which is also when GEMM tiles are autotuned) and
    from then on replayed with one C call per frame.  ``state`` is the caller's per-(shape, variant) dict."""
    mode = MODES[launch_mode]
    if mode is None or n_lanes > MAX_LANES:
        frame_fn(EagerExec(n_lanes, pool))
        return
    if state.get("cplan") is not None and state.get("mode") == mode:
        state["cplan"].run()
        return
    if not state.get("warm"):
        frame_fn(EagerExec(n_lanes, pool))
        state["warm"] = True
        return
    cplan = LaunchPlan()
    with RecordExec(cplan, n_lanes, mode) as ex:
        frame_fn(ex)
    state["cplan"], state["mode"] = cplan, mode
    cplan.run()
