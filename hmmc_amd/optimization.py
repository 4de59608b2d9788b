"""BertAdam with the reference's constructor / step() / get_lr() / state layout
(reference modules/optimization.py:26-168), executed as TWO multi-tensor HIP launches per step
(per-parameter gradient norms, then the fused clip + moment + weight update) instead of ~10 tiny
launches for each of the 350-560 parameters.  Arithmetic follows the reference's tensor expressions op by
op in each parameter's own dtype (fp16 parameters keep fp16 moments), no bias correction, step counter
starting at 0 so the first update uses lr 0 under warm-up.
"""
from __future__ import annotations

import ctypes
import os
import math
import weakref

import numpy as np
import torch
from torch.optim import Optimizer
from torch.optim.optimizer import required

from ._lib import call, load, ptr


_NO_SHARED_NORMS = os.environ.get("HMMC_NO_SHARED_NORMS", "0") == "1"      # A/B runs and tests


def warmup_cosine(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {"warmup_cosine": warmup_cosine, "warmup_constant": warmup_constant, "warmup_linear": warmup_linear}


class _TensorTable:
    """Device-side pointer/chunk tables for the multi-tensor kernels.  The tables are built once per set of tensors; from then
    on a step only refreshes the column of gradient pointers when the allocator handed out different ones (with several
    streams in play it does, every step).  That refresh goes through a small ring of pinned staging rows that are allocated
    once and reused behind an event each: `pin_memory()` per upload made the pinned-memory cache grow by 18-30 MiB per step
    whenever the host ran ahead of the GPU (scratch/leak_check.py)."""

    RING = 8

    def __init__(self, device):
        self.device = device
        self.key = None
        self.chunk_elems = load().hmmc_mt_chunk_elems()
        self._ring, self._events, self._slot = None, None, 0

    def _refresh_column(self, col, values):
        T = len(values)
        if self._ring is None or self._ring.shape[1] != T:
            self._ring = torch.empty((self.RING, T), dtype=torch.int64).pin_memory()
            self._events = [None] * self.RING
        k = self._slot
        self._slot = (k + 1) % self.RING
        if self._events[k] is not None:
            self._events[k].synchronize()            # the copy that last read this row has run (RING steps ago)
        self._ring[k].numpy()[:] = values
        self.tab[:, col].copy_(self._ring[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._events[k] = ev

    def build(self, rows):
        """rows: int64 array [n, k] or list of (ptr0, ptr1, ptr2, ptr3, numel, dtype_flag[, group])."""
        rows = np.asarray(rows, dtype=np.int64)
        if self.key is not None and rows.shape == self.key.shape:
            if np.array_equal(rows, self.key):
                return self
            diff = np.flatnonzero((rows != self.key).any(axis=0))
            if len(diff) == 1 and diff[0] < 4:         # one pointer column moved (the gradients): refresh it in place
                self._refresh_column(int(diff[0]), rows[:, diff[0]])
                self.key = rows.copy()
                return self
        # full (re)build: rare (first steps, a parameter set that changed); plain blocking uploads
        arr = np.zeros((len(rows), 8), dtype=np.int64)
        arr[:, :rows.shape[1]] = rows
        nch = (arr[:, 4] + self.chunk_elems - 1) // self.chunk_elems
        tidx = np.repeat(np.arange(len(rows)), nch)
        cidx = np.concatenate([np.arange(n) for n in nch]) if len(rows) else np.zeros(0, dtype=np.int64)
        chunks = np.stack([tidx, cidx], axis=1).astype(np.int32)
        arr[:, 7] = np.cumsum(nch) - nch                # index of the tensor's first chunk (the fixed-order norm reduction)
        self.tab = torch.from_numpy(arr).to(self.device)
        self.chunk = torch.from_numpy(chunks).to(self.device)
        self.nchunks = int(chunks.shape[0])
        self.T = len(rows)
        self.sumsq = torch.zeros(len(rows) + self.nchunks, dtype=torch.float32, device=self.device)   # [T] norms + per-chunk partials
        self.key = rows.copy()
        return self


def _dtype_flag(t):
    if t.dtype == torch.float16:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise TypeError(f"unsupported parameter dtype {t.dtype}")


_clip_tables = {}
_pending_norms = {}          # device -> what the last clip_grad_norm_ left for the optimizer (see there)


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) semantics (main_task_retrieval.py:291) in three
    launches over all gradients.  Returns the total norm (0-dim device tensor, no host sync)."""
    plist = [p for p in parameters if p.grad is not None]
    grads = [p.grad for p in plist]
    if not grads:
        return torch.zeros(())
    dev = grads[0].device
    tbl = _clip_tables.setdefault(str(dev), _TensorTable(dev))
    # per step only the gradient pointers can change (and in steady state the caching allocator hands the same ones back):
    # the static columns are kept with the table and the device copy is refreshed only when a pointer moved
    # The size and dtype columns are rebuilt from the gradients on EVERY call (a key of count and total size alone would
    # let a different gradient set with the same totals - the same model after .float(), a second model - reuse stale sizes
    # and dtypes: out-of-bounds reads); build() compares the whole table and uploads only what changed.
    n = len(grads)
    rows = np.zeros((n, 6), dtype=np.int64)
    rows[:, 1] = np.fromiter((g.data_ptr() for g in grads), dtype=np.int64, count=n)
    rows[:, 4] = np.fromiter((g.numel() for g in grads), dtype=np.int64, count=n)
    rows[:, 5] = np.fromiter((_dtype_flag(g) for g in grads), dtype=np.int64, count=n)
    if not all(g.is_contiguous() for g in grads):
        raise ValueError("gradients must be contiguous")
    tbl.build(rows)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    # The scaling pass also leaves every gradient's squared norm as it stands afterwards: BertAdam's per-parameter clip needs
    # exactly that next, and would otherwise read every gradient once more (optimization.py:140-150 of the reference calls
    # clip_grad_norm_ per parameter inside step()).  The hand-over (_pending_norms) names the gradients by address and version
    # counter, is consumed by the first step() that sees it and ignored by one whose gradients differ in either.
    if getattr(tbl, "sumsq_after", None) is None or tbl.sumsq_after.numel() != tbl.sumsq.numel():
        tbl.sumsq_after = torch.zeros_like(tbl.sumsq)
    call("hmmc_mt_clip_grad_norm_keep", ptr(tbl.tab), ptr(tbl.chunk), tbl.nchunks, ptr(tbl.sumsq), tbl.T, float(max_norm), ptr(out),
         ptr(tbl.sumsq_after))
    # The gradients are named by tensor OBJECT (weak references), not by address: the library's raw-pointer kernels never bump
    # `_version` and the caching allocator hands the same addresses back every step, so address + version alone would let a
    # later, different gradient at the same place (clip, no step, zero_grad(set_to_none), backward, step without clip) pass.
    _pending_norms[str(dev)] = {"pids": np.fromiter((id(p) for p in plist), dtype=np.int64, count=n), "ptrs": rows[:, 1].copy(),
                                "versions": np.fromiter((g._version for g in grads), dtype=np.int64, count=n),
                                "grads": [weakref.ref(g) for g in grads],
                                "norms": tbl.sumsq_after, "stream": torch.cuda.current_stream(dev)}
    return out[1]


def _drop_pending_norms(device=None):
    """Forget what the last clip_grad_norm_ left for the optimizer (a step that does not consume it, zero_grad)."""
    if device is None:
        _pending_norms.clear()
    else:
        _pending_norms.pop(str(device), None)


class BertAdam(Optimizer):
    """Implements BERT version of Adam algorithm with weight decay fix (same arguments as the reference)."""

    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        self._table = None
        self._fast = None

    # ---- steady-state fast path.  step() above walks 350-560 parameters in Python every call (state lookups, schedule, tuple
    # rows: 2 ms of host time, which at small per-GPU batches is what the step waits for).  Once a full step has shown that
    # every parameter has a gradient and the parameters of a group share their step count, the static part of the table (weights,
    # moments, sizes, group index) is kept and a step only reads the gradient pointers and evaluates the schedule per group.
    def _plan_fast_path(self, n_rows, steps_per_group):
        """steps_per_group: {group index: set of the step counts its parameters had in the slow step just taken}.
        The plan covers the parameters that HAD a gradient in that step; the others (the pre-training model's t_projector, built and
        never used, reference modules/modeling.py:113-114; the momentum encoders, requires_grad False) are remembered as `idle` and
        the fast path holds only while exactly they have none - round 5: the pre-training step never took the fast path before."""
        self._fast = None
        groups = [g for g in self.param_groups if any(p.grad is not None for p in g["params"])]
        params = [p for g in groups for p in g["params"] if p.grad is not None]
        idle = [p for g in self.param_groups for p in g["params"] if p.grad is None]
        if n_rows != len(params) or not params or len(groups) > 32 or any(len(s) != 1 for s in steps_per_group.values()):
            return                                   # a group's parameters differ in step count
        static = np.zeros((len(params), 7), dtype=np.int64)
        row = 0
        for gi, g in enumerate(groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                static[row] = (p.data_ptr(), 0, st["next_m"].data_ptr(), st["next_v"].data_ptr(), p.numel(), _dtype_flag(p), gi)
                row += 1
        self._fast = {"groups": groups, "params": params, "idle": idle, "static": static, "states": [self.state[p] for p in params],
                      "table": _TensorTable(params[0].device), "n_total": sum(len(g["params"]) for g in self.param_groups)}

    def _fast_step(self):
        fp = getattr(self, "_fast", None)
        if fp is None:
            return False
        grads = [p.grad for p in fp["params"]]
        ps, static = fp["params"], fp["static"]
        if (any(g is None for g in grads) or any(p.grad is not None for p in fp["idle"])
                or sum(len(g["params"]) for g in self.param_groups) != fp["n_total"]):
            self._fast = None
            return False
        # parameters moved / a moment tensor was replaced: every pointer is compared, as one array each
        n = len(ps)
        states = fp["states"]
        if not (np.array_equal(np.fromiter((p.data_ptr() for p in ps), dtype=np.int64, count=n), static[:, 0])
                and np.array_equal(np.fromiter((st["next_m"].data_ptr() for st in states), dtype=np.int64, count=n), static[:, 2])
                and np.array_equal(np.fromiter((st["next_v"].data_ptr() for st in states), dtype=np.int64, count=n), static[:, 3])):
            self._fast = None
            return False
        hp = []
        for g in fp["groups"]:
            steps = {self.state[p]["step"] for p in g["params"] if p.grad is not None}
            if len(steps) != 1:
                self._fast = None
                return False
            step = steps.pop()
            lr_s = g["lr"] * SCHEDULES[g["schedule"]](step / g["t_total"], g["warmup"]) if g["t_total"] != -1 else g["lr"]
            hp += [lr_s, g["weight_decay"], g["b1"], g["b2"], g["e"], g["max_grad_norm"], 1 - g["b1"], 1 - g["b2"]]
        rows = fp["static"].copy()
        rows[:, 1] = np.fromiter((g.data_ptr() for g in grads), dtype=np.int64, count=len(grads))
        tbl = fp["table"].build(rows)
        hp_host = (ctypes.c_float * len(hp))(*hp)
        index = self._norms_from_clip(fp, rows[:, 1], grads, tbl.device)
        if index is not None:
            norms, idx = index
            call("hmmc_mt_bertadam_ext", ptr(tbl.tab), hp_host, len(fp["groups"]), ptr(tbl.chunk), tbl.nchunks, tbl.T, ptr(norms), ptr(idx))
        else:
            call("hmmc_mt_bertadam", ptr(tbl.tab), hp_host, len(fp["groups"]), ptr(tbl.chunk), tbl.nchunks, ptr(tbl.sumsq), tbl.T)
        for st in fp["states"]:
            st["step"] += 1
        return True

    @staticmethod
    def _norms_from_clip(fp, grad_ptrs, grads, device):
        """(norms, index) when the clip_grad_norm_ call just before this step left the squared norms of exactly these gradients
        (the same parameters' gradients at the same addresses, version counters unchanged since, same stream): index[t] =
        position of this table's tensor t in that call's list.  None otherwise - the optimizer then forms the norms itself.
        The index depends on the two parameter ORDERS only, so it is built (and uploaded) once, not per step: gradient
        addresses change from step to step.  HMMC_NO_SHARED_NORMS=1 switches the hand-over off."""
        rec = _pending_norms.pop(str(device), None)            # single use: a later step() never sees a stale hand-over
        if rec is None or _NO_SHARED_NORMS or rec["stream"] != torch.cuda.current_stream(device):
            return None
        cached = fp.get("clip_index")
        if cached is None or not np.array_equal(cached[0], rec["pids"]):
            pos = {int(a): i for i, a in enumerate(rec["pids"])}
            host = np.fromiter((pos.get(id(p), -1) for p in fp["params"]), dtype=np.int64, count=len(fp["params"]))
            if len(pos) != len(rec["pids"]) or (host < 0).any():
                return None                                    # a parameter the clip did not see
            cached = (rec["pids"].copy(), host, torch.from_numpy(host.astype(np.int32)).to(device))
            fp["clip_index"] = cached
        host = cached[1]
        if not (np.array_equal(rec["ptrs"][host], grad_ptrs) and
                np.array_equal(np.fromiter((g._version for g in grads), dtype=np.int64, count=len(grads)), rec["versions"][host])):
            return None                                        # another gradient tensor, or one written to after the clip
        refs = rec["grads"]
        if not all(refs[j]() is g for j, g in zip(host.tolist(), grads)):
            return None                                        # a different tensor object at the same address (see clip_grad_norm_)
        return rec["norms"], cached[2]

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if p.is_cuda:
                    _drop_pending_norms(p.device)
                    break
        return super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict):
        self._fast, self._table = None, None                  # the cached tables point at the old moment tensors
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        self._fast, self._table = None, None
        return super().add_param_group(param_group)

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                if group["t_total"] != -1:
                    lr_scheduled = group["lr"] * SCHEDULES[group["schedule"]](state["step"] / group["t_total"], group["warmup"])
                else:
                    lr_scheduled = group["lr"]
                lr.append(lr_scheduled)
        return lr

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._fast_step():
            return loss
        _drop_pending_norms()                               # the slow path forms its own norms: a hand-over is never left behind
        rows, frows, group_steps = [], {}, {}
        dev = None
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if not (p.is_contiguous() and p.grad.is_contiguous()) or p.grad.dtype != p.dtype:
                    raise ValueError("BertAdam (HIP) needs contiguous parameters and same-dtype gradients")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["next_m"] = torch.zeros_like(p.data)
                    state["next_v"] = torch.zeros_like(p.data)
                if group["t_total"] != -1:
                    lr_s = group["lr"] * SCHEDULES[group["schedule"]](state["step"] / group["t_total"], group["warmup"])
                else:
                    lr_s = group["lr"]
                dev = p.device
                # one hyper-parameter row per DISTINCT (group, step count), wherever its parameters sit in the list (a
                # parameter that skipped steps - grad None for a while, partial optimizer state - has its own row)
                hp = (lr_s, group["weight_decay"], group["b1"], group["b2"], group["e"], group["max_grad_norm"],
                      1 - group["b1"], 1 - group["b2"])
                gi_row = frows.setdefault(hp, len(frows))
                group_steps.setdefault(gi, set()).add(state["step"])
                rows.append((p.data_ptr(), p.grad.data_ptr(), state["next_m"].data_ptr(), state["next_v"].data_ptr(),
                             p.numel(), _dtype_flag(p), gi_row))
                state["step"] += 1
        if not rows:
            return loss
        if dev.type != "cuda":
            raise RuntimeError("hmmc_amd.BertAdam runs on the GPU only (no CPU fallback)")
        self._plan_fast_path(len(rows), group_steps)
        if self._table is None:
            self._table = {}
        hps = list(frows)                                   # insertion order = row index
        MAXG = 32                                           # hyper-parameter rows one launch takes (kernel MAX_GROUPS)
        for part in range((len(hps) + MAXG - 1) // MAXG):   # normally one launch; more only with > 32 distinct rows
            lo = part * MAXG
            sel = rows if len(hps) <= MAXG else [r[:6] + (r[6] - lo,) for r in rows if lo <= r[6] < lo + MAXG]
            tbl = self._table.setdefault(part, _TensorTable(dev)).build(sel)
            hp_part = hps[lo:lo + MAXG]
            hp_host = (ctypes.c_float * (8 * len(hp_part)))(*[x for row in hp_part for x in row])
            call("hmmc_mt_bertadam", ptr(tbl.tab), hp_host, len(hp_part), ptr(tbl.chunk), tbl.nchunks, ptr(tbl.sumsq), tbl.T)
        return loss
