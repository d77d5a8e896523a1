"""A small static tape for the hand-written HIP kernels.

The reference relies on ``torch.autograd`` over ~2000 ATen ops per step.  Here one top-level module call
(``CineMA.forward``) is ONE ``torch.autograd.Function`` node: its forward runs the kernel sequence while recording
backward closures on a :class:`Tape`; its backward replays them in reverse.  All activations are 2-D row-major
matrices (rows = tokens / voxels in raster order, channels last); fp32 for the residual stream and losses, bf16
for everything that feeds an MFMA GEMM.  Every op below is a sequence of ``cinema_amd.hip`` launches -- there is no
ATen compute on this path except index bookkeeping on tiny integer tensors.
"""

from __future__ import annotations

import math
import contextlib
import os
import weakref
from collections import deque
from typing import Callable

import torch

from cinema_amd import hip as K

BF16, F32 = torch.bfloat16, torch.float32


class Var:
    """An activation on the tape: ``data`` plus its (lazily created) gradient.

    ``grad`` is fp32 for fp32 data and bf16 for bf16 data - except that a producer whose gradient has exactly ONE reader which takes bf16 anyway (a GEMM operand)
    may hand a bf16 gradient to fp32 data; ``add_grad`` widens it to fp32 the moment a second contribution arrives.  ``grad16`` optionally carries a bf16 copy of
    an fp32 gradient (written for free by the LayerNorm backward) so that dgrad/wgrad GEMMs need no extra cast pass.
    """

    __slots__ = ("data", "grad", "grad16", "needs_grad", "fp8", "grad8", "grad8_site", "fp8t", "grad8_bias", "grad8_bias_done", "grad_any")

    def __init__(self, data: torch.Tensor, needs_grad: bool = True) -> None:
        self.data = data
        self.grad: torch.Tensor | None = None
        self.grad16: torch.Tensor | None = None
        self.needs_grad = needs_grad
        self.fp8: tuple | None = None  # (e4m3 copy of data, per-row scales [rows] or per-tensor scale [1]) when the producer emitted one (op_layernorm(fp8=True))
        self.grad8: tuple | None = None  # (e4m3 copy of the complete gradient, per-tensor scale [1]): written by the LayerNorm backward that produces the gradient
        self.grad8_site = None  # Q8Site: set by the op that wants that copy (its weight- and data-gradient GEMMs read it)
        self.fp8t = None  # (e4m3 copy, per-tensor scale [1]) of a tensor several ops read (the decoder's shared keys): made once by the first of them
        self.grad8_bias = None  # the bias Parameter of the projection that produced this tensor: the LayerNorm backward that writes grad8 also sums the columns of the gradient
        self.grad8_bias_done = False  # ... and says so here (the projection's own backward then adds no bias gradient)
        self.grad_any = False  # the producer's backward reads the gradient in either dtype (op_assemble: a gather + column sums): a bf16 gradient needs no fp32 copy

    def add_grad(self, g: torch.Tensor, g16: torch.Tensor | None = None) -> None:
        if not self.needs_grad:
            return
        if self.grad is None:
            self.grad, self.grad16 = g, g16
        else:
            if self.grad.dtype == BF16 and self.data.dtype == F32:
                # the first contribution was handed over in bf16 (op_mse, Segment(grad_bf16=True), op_droppath_add, op_cast_bf16 with grad_any: single-consumer
                # shortcuts of the glue trims); a SECOND consumer exists after all: widen once, accumulate in fp32 from here on
                self.grad = K.cast(self.grad, F32)
            K.row_copy(self.grad.view(-1, self.grad.shape[-1]), g.view(-1, g.shape[-1]), accumulate=True)
            self.grad16 = None
            self.grad8 = None
            if self.grad8_bias_done:
                # a LayerNorm backward summed the columns of what it believed to be the complete gradient into the producing projection's bias gradient
                # (op_layernorm); a later contribution is not in that sum - the block wiring of this build never does this, so fail loudly instead of
                # training with a partial bias gradient
                raise RuntimeError("Var.add_grad: a gradient contribution arrived after a LayerNorm backward had already summed this tensor's columns as a bias gradient")

    def grad_bf16(self) -> torch.Tensor:
        """bf16 view of the gradient (GEMM operand)."""
        if self.grad is None:
            raise RuntimeError("gradient requested before it was produced")
        if self.grad.dtype == BF16:
            return self.grad
        if self.grad16 is None:
            self.grad16 = K.cast(self.grad, BF16)
        return self.grad16


def zeros(shape, dtype: torch.dtype = F32, device=None) -> torch.Tensor:  # noqa: ANN001
    """Zero tensor whose fill is a launch of the HIP library on GPU (part of a recorded step); torch.zeros elsewhere (host-logic tests)."""
    if torch.device(device).type == "cuda":
        return K.zeros(shape, dtype, device)
    return torch.zeros(shape, dtype=dtype, device=device)


# ---- recorded steps (cinema_amd/replay.py) ---------------------------------------------------------------------
# A recorded step re-issues the HIP launches of one eager step verbatim.  Device work done by torch (ATen) between them is not in that
# list, so the model code marks it: host() for index tensors that depend on this step's random masks (re-run on every replay, results
# copied into the tensors the recorded launches point at) and const() for tensors that depend on shapes only (computed once, cached).
_CONST: dict = {}
REC_CALL = None    # while recording: the (tape, vars, params) of the step's top-level call, whose backward the recorder runs itself


class _Allowed:
    """Marks torch ops as accounted for while a recording is audited (see cinema_amd.replay)."""

    def __enter__(self) -> None:
        if K.RECORD is not None:
            from cinema_amd import replay
            self.ctx = replay.allowed_aten()
            self.ctx.__enter__()
        else:
            self.ctx = None

    def __exit__(self, *exc) -> None:  # noqa: ANN002
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def const(key: tuple, fn: Callable) -> torch.Tensor:
    """Shape-only device tensor (index tables): computed once per key and kept for the life of the process."""
    t = _CONST.get(key)
    if t is None:
        with _Allowed():
            t = _CONST[key] = K.persistent(fn)  # never from a recording's private pool: its content is not rebuilt by the replayed launches
    return t


def host(fn: Callable) -> tuple:
    """``fn() -> tuple of tensors`` computed with torch ops from this step's masks.  Eager: just call it.  While recording: the results
    become static tensors and the call is appended to the recording as a host entry that recomputes them in place on every replay."""
    if K.RECORD is None:
        return fn()
    with _Allowed():
        static = tuple(o.clone() for o in fn())
    # a view stream, possibly: the replayed torch ops must be queued where the recorded consumers wait
    st = torch.cuda.current_stream() if static and static[0].is_cuda else None

    def again() -> None:
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            for dst, src in zip(static, fn()):
                dst.copy_(src)

    K.RECORD.append((None, again))
    return static


def recording() -> bool:
    return K.RECORD is not None


class PVar:
    """A parameter on the tape.  ``grad`` (fp32, in the *kernel* layout) is created zeroed on first use."""

    __slots__ = ("param", "grad", "to_param_layout", "direct")

    def __init__(self, param: torch.nn.Parameter) -> None:
        self.param = param
        self.grad: torch.Tensor | None = None
        self.to_param_layout: Callable | None = None
        self.direct = False

    def grad_buffer(self, shape: tuple, to_param_layout: Callable | None = None) -> torch.Tensor:
        if self.grad is None:
            flat = getattr(self.param, "_cinema_flat_grad", None)
            if flat is not None and to_param_layout is None and flat.data_ptr() == getattr(self.param.grad, "data_ptr", lambda: 0)():
                # the optimiser owns a flat gradient buffer (cinema_amd.optim.FlatModel): accumulate straight into it
                self.grad, self.direct = flat.view(shape), True
            else:
                self.grad = zeros(shape, F32, self.param.device)
            self.to_param_layout = to_param_layout
        return self.grad

    def final_grad(self) -> torch.Tensor | None:
        if self.grad is None or self.direct:
            return None
        g = self.to_param_layout(self.grad) if self.to_param_layout is not None else self.grad
        return g.reshape(self.param.shape)


class WeightCache:
    """bf16 (and re-laid-out) shadows of the fp32 master parameters, refreshed when a parameter changes.

    Keyed by (parameter identity, kind); validated by ``param._version``, the storage pointer and a global epoch that
    optimisers writing through raw pointers (``cinema_amd.optim.FusedAdamW``) bump after each step.
    """

    def __init__(self) -> None:
        self.epoch = 0
        self._store: dict = {}

    def invalidate(self) -> None:
        self.epoch += 1

    def get(self, params: tuple, kind: str, build: Callable) -> torch.Tensor:
        key = (tuple(id(p) for p in params), kind)
        stamp = (self.epoch, tuple((p._version, p.data_ptr()) for p in params))  # noqa: SLF001
        hit = self._store.get(key)
        # the weak references pin the entry to the parameter OBJECTS: a dead parameter's id(), storage address and version can all be
        # handed to a new one (seen as a rare stale-shadow error when many models are built in one process, e.g. the test suite)
        if hit is not None and hit[0] == stamp and all(r() is p for r, p in zip(hit[2], params)):
            return hit[1]
        val = build()
        if len(self._store) > 4096:  # entries of dead models
            self._store = {k: v for k, v in self._store.items() if all(r() is not None for r in v[2])}
        self._store[key] = (stamp, val, tuple(weakref.ref(p) for p in params))
        return val


WEIGHTS = WeightCache()

# Data-parallel overlap: when set (cinema_amd.ddp.GradientSynchronizer), called in the BACKWARD pass as hook(tape, params) at the
# point where every gradient contribution of ``params`` has been launched (see mark_params).
PARAMS_DONE_HOOK: Callable | None = None


# Weight-gradient GEMMs on a second HIP stream (default; CINEMA_SIDE_WGRAD=0 keeps one stream): nothing in the backward chain waits for them, so they can fill the
# compute units left idle by the tails of the kernels on the main stream.  Each launch waits for an event recorded on the main stream
# (its dy / x operands are complete); Tape.backward() and the all-reduce hook join the side stream before anyone reads the gradients.
SIDE_WGRAD = bool(int(os.environ.get("CINEMA_SIDE_WGRAD", "1")))
_SIDE_STREAMS: dict = {}


def side_stream(i: int = 0) -> "torch.cuda.Stream":
    dev = torch._C._cuda_getDevice()
    st = _SIDE_STREAMS.get((dev, i))
    if st is None:
        st = _SIDE_STREAMS[(dev, i)] = torch.cuda.Stream(device=dev)
        if i == 0:
            _SIDE_STREAMS[dev] = st  # ("this device has a weight-gradient stream")
    return st


def _ranks_share_a_device() -> bool:
    """More local ranks than visible devices (torchrun sets LOCAL_WORLD_SIZE): several processes then drive ONE GPU, and three streams per process oversubscribe
    its hardware queues (measured: the two-rank shared-GPU check took 162 s instead of 9 s with the third stream)."""
    try:
        local = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        return local > 1 and torch.cuda.is_available() and local > torch.cuda.device_count()
    except (ValueError, RuntimeError):
        return False


# A SECOND weight-gradient stream for the single (not grouped) weight gradients - the conv stems': an event timeline of the replayed step without the profiler
# (tools/phase_events.py, profiles/r05_o_phase_events.txt) shows the main stream finishing its chain 0.9 ms before the end of the backward pass and then waiting
# for the weight-gradient stream to work off ~20 small split-K GEMM + reduce pairs one after the other on an otherwise empty chip; dealt alternately to two
# streams they run two at a time.  CINEMA_SIDE_STREAMS=1: one stream.
SIDE_STREAMS = max(1, min(2, int(os.environ.get("CINEMA_SIDE_STREAMS", "1" if _ranks_share_a_device() else "2"))))


class side_streams_limit:  # noqa: N801
    """``with side_streams_limit(1):`` - at most n weight-gradient streams for the launches issued (or recorded) inside, unless CINEMA_SIDE_STREAMS is set explicitly.
    The ConvUNetR step (BASELINE config 4) is 0.35 ms SLOWER with the second stream (46.86 / 46.92 vs 47.24 / 47.25 ms, profiles/r05_p_side_streams_ab.txt): its
    implicit-convolution weight gradients already fill the chip beside the main stream, a third runnable queue only adds contention."""

    def __init__(self, n: int) -> None:
        self.n, self.saved = n, None

    def __enter__(self) -> None:
        global SIDE_STREAMS
        self.saved = SIDE_STREAMS
        if "CINEMA_SIDE_STREAMS" not in os.environ:
            SIDE_STREAMS = max(1, min(SIDE_STREAMS, self.n))

    def __exit__(self, *exc) -> None:  # noqa: ANN002
        global SIDE_STREAMS
        SIDE_STREAMS = self.saved


_SIDE_ALT = [0]   # stream index of the last single weight-gradient launch (what depends on that launch goes to the same stream)
# Accumulated destination (data pointer) -> the weight-gradient stream that last added into it during this backward pass.  A weight used more than once per step
# (dec_linear: once per view) gets several read-modify-write launches on ONE buffer; dealt to different streams they ran at the same time and lost updates
# (found by tests/test_ddp_gpu.py: ~1 run in 8 had dec_linear.weight's gradient off by 20 %, tools/ddp_first_step_probe.py) - they follow each other on one stream.
_DST_STREAM: dict = {}


_SIDE_KEEP: deque = deque()  # (completion event, operands) of weight-gradient launches still (possibly) running on the side stream

# The lane group of the long-axis stems on a THIRD stream (default; CINEMA_LAX_STREAM=0: on the main stream, after the short-axis stem): the stems are chains of
# ~45 forward / ~95 backward small launches per view (16-50 us each for the short-axis view, 5-15 us for the zipped long-axis group) that leave most compute
# units idle, and the views share nothing - so the long-axis chain runs BESIDE the short-axis one: forked before the first stem launch, joined before the token
# assembly (forward) / at the end of the backward pass.  The buffers its lanes allocate (under torch's current stream) are held until that join.
# (default on; switched off by itself when several local ranks share one device - the product layout is one process per GPU)
LAX_STREAM = bool(int(os.environ.get("CINEMA_LAX_STREAM", "0" if _ranks_share_a_device() else "1")))
_LAX_STREAMS: dict = {}
_LAX_KEEP: list = []


def lax_stream() -> "torch.cuda.Stream":
    dev = torch._C._cuda_getDevice()
    st = _LAX_STREAMS.get(dev)
    if st is None:
        st = _LAX_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


def join_lax_stream() -> None:
    """The current stream waits for everything issued to the long-axis stream so far."""
    # (not while the current stream is being captured into a HIP graph: run_in_lanes keeps the long-axis stream out of a capture, so there is nothing
    # to join - and an event recorded on a stream OUTSIDE the capture must not be waited on from inside it)
    if _LAX_STREAMS and torch._C._cuda_getDevice() in _LAX_STREAMS and not torch._C._cuda_isCurrentStreamCapturing():
        K.stream_fork(lax_stream().cuda_stream, K._stream())


def join_side_stream(release: bool = False) -> None:
    """Make the current stream wait for the side stream.  ``release`` (end of the backward pass): the operands kept alive for the side
    stream may go back to the allocator - their next user is ordered after this wait."""
    if SIDE_WGRAD and _SIDE_STREAMS:
        K.stream_fork(side_stream().cuda_stream, K._stream())
        if (torch._C._cuda_getDevice(), 1) in _SIDE_STREAMS:
            K.stream_fork(side_stream(1).cuda_stream, K._stream())
    if release:
        _SIDE_KEEP.clear()
        _LAX_KEEP.clear()
        _DST_STREAM.clear()


def mark_params(tape: "Tape", params: list) -> None:
    """Record, at the START of a module's forward ops, that ``params`` are used by these ops only: in the reversed backward order the
    marker runs after all of their gradient kernels, which is where their gradient all-reduce may start."""
    hook = PARAMS_DONE_HOOK
    if hook is not None and tape.train:
        def run_hook() -> None:
            # the block's weight gradients may still be running on the side stream: issue the collective FROM the side stream (after
            # everything queued on the main stream so far), so that RCCL orders itself behind both without stalling the main stream
            if SIDE_WGRAD and torch.cuda.is_available():
                main, side = torch.cuda.current_stream(), side_stream()
                ev = torch.cuda.Event()
                ev.record(main)
                if (torch._C._cuda_getDevice(), 1) in _SIDE_STREAMS:
                    K.stream_fork(side_stream(1).cuda_stream, side.cuda_stream)  # (weight gradients dealt to the second stream)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    hook(tape, params)
            else:
                hook(tape, params)

        def start() -> None:
            if K.RECORD is not None:  # recorded step: the collective is a torch call, so it enters the launch list as a host entry
                K.RECORD.append((None, run_hook))
            run_hook()

        def fire() -> None:
            # deferred LayerNorm parameter gradients of the ops launched so far (this block's included) must be in the flat buffer before any
            # of its ranges is handed to a collective: reduce the pending partials now (one extra launch per block, data-parallel runs only)
            flush_ln(tape)
            if tape.pending_wgrads or tape.pending_wgrads8:
                # this block's weight-gradient group is held back for a launch shared with the next block (GROUP_FLUSH_MIN): its collective starts
                # behind that launch (flush_wgrads runs the held hooks) - one block later, with most of the backward pass still ahead to hide it
                tape.held_hooks.append(start)
                return
            start()

        tape.record(fire)


class Tape:
    """Backward closures in forward order + the parameter registry of one top-level call."""

    def __init__(self, params: dict | None = None, train: bool = True) -> None:
        self.ops: list = []
        self.pending_wgrads: list = []  # (dy, x, dst, bias_grad) deferred to the enclosing weight-gradient group
        self.pending_wgrads8: list = []  # the same for weight gradients on e4m3 operands (wgrad8_problem)
        self.pending_ln: list = []      # LayerNorm parameter-gradient partials: one batched reduce at the end of the backward pass
        self.held_hooks: list = []      # gradient-exchange hooks of blocks whose weight-gradient group has not been launched yet (mark_params)
        self.grouping = False
        self.pvars: dict = {}
        self.train = train

    def record(self, fn: Callable) -> None:
        if self.train:
            self.ops.append(fn)

    def pvar(self, param: torch.nn.Parameter | None) -> PVar | None:
        if param is None:
            return None
        pv = self.pvars.get(id(param))
        if pv is None:
            pv = self.pvars[id(param)] = PVar(param)
        return pv

    def backward(self) -> None:
        debug = os.environ.get("CINEMA_TAPE_DEBUG") == "1"
        # whatever the weight-gradient stream still has queued from before this step (TrainStep's zero_grad fill) precedes the first gradient this pass writes
        if SIDE_WGRAD and _SIDE_STREAMS and self.ops and torch.cuda.is_available() and not torch._C._cuda_isCurrentStreamCapturing():
            K.stream_fork(side_stream().cuda_stream, K._stream())
        try:
            for i, fn in enumerate(reversed(self.ops)):
                fn()
                if debug:  # localise an asynchronous kernel fault to one backward closure
                    torch.cuda.synchronize()
                    print(f"[tape] bwd {len(self.ops) - 1 - i:4d} {fn.__qualname__} ok", flush=True)
        except BaseException:
            # a backward lane group is opened and closed by two separate closures: a failure in between would leave the library recording (and never
            # issuing) every later launch of the process
            if K.LANE is not None:
                K.lanes_abort()
            raise
        join_lax_stream()  # the long-axis stems' backward (lane_group(stream=...)): their LayerNorm partials are reduced below, on this stream
        flush_wgrads(self)
        flush_ln(self)  # 86 LayerNorms per step: one launch per 48 instead of one each
        join_side_stream(release=True)
        self.ops = []


class lane_group:  # noqa: N801
    """Forward AND backward lane group (``hip.lanes``) over n independent, identically shaped op sequences recorded on ``tape``::

        with lane_group(tape, 3) as g:
            for i, view in enumerate(views):
                g.select(i)
                ...ops of this view...

    The launches of the n sequences go out zipped (one wide launch per position) when the block ends; the backward closures recorded inside
    run later under a mirrored group.  The sequences must not share parameters, gradient buffers or activations."""

    def __init__(self, tape: "Tape", n: int, stream: int | None = None) -> None:
        self.tape, self.n = tape, n
        self.fwd = K.lanes(n)
        self.bwd = None
        # ``stream`` (raw handle): the group's forward AND backward launches go to that stream.  The caller forks it from the current stream before the forward
        # group and joins it afterwards; the backward group forks when it opens, Tape.backward() joins.  Only when the lanes are really deferred (their
        # allocations are then tracked and can be held until the join).
        self.stream = stream if self.fwd.active else None
        self._ovr = None

    def __enter__(self) -> "lane_group":
        self.tape.record(self._bwd_close)  # runs LAST in the reversed backward order
        if self.stream is not None:
            self._ovr = K.on_stream(self.stream)
            self._ovr.__enter__()
        self.fwd.__enter__()
        return self

    def select(self, lane: int) -> None:
        self.fwd.select(lane)
        self.tape.record(lambda: self.bwd.select(lane - 1) if (lane > 0 and self.bwd is not None) else None)  # runs after lane `lane`'s backward ops

    def __exit__(self, *exc) -> None:  # noqa: ANN002
        if self.stream is not None:
            K.LANE_KEEP_SINK = _LAX_KEEP
        try:
            self.fwd.__exit__(*exc)
        finally:
            if self.stream is not None:
                K.LANE_KEEP_SINK = None
                self._ovr.__exit__()
                self._ovr = None
        if exc[0] is None:
            self.tape.record(self._bwd_open)  # runs FIRST in the backward pass

    def _bwd_open(self) -> None:
        self.bwd = K.lanes(self.n)
        if self.stream is not None and self.bwd.active:
            K.stream_fork(K._stream(), self.stream)  # the gradients entering the group were produced on the current stream
            self._ovr = K.on_stream(self.stream)
            self._ovr.__enter__()
        self.bwd.__enter__()
        self.bwd.select(self.n - 1)

    def _bwd_close(self) -> None:
        if self.bwd is not None:
            if self._ovr is not None:
                K.LANE_KEEP_SINK = _LAX_KEEP
            try:
                self.bwd.__exit__(None, None, None)
            finally:
                self.bwd = None
                if self._ovr is not None:
                    K.LANE_KEEP_SINK = None
                    self._ovr.__exit__()
                    self._ovr = None


def run_in_lanes(tape: "Tape", items: list, key: Callable, body: Callable, enabled: bool = True, beside: bool = False) -> None:
    """``body(item)`` for every item, in order; runs of 2-4 consecutive items with equal ``key(item)`` (identical launch geometry, nothing shared)
    are issued as one lane group.  ``beside``: the lane groups go to the long-axis stream (LAX_STREAM), beside the single items on the current stream - the
    caller guarantees that the items read nothing another item of this call writes; the current stream has joined when this returns."""
    runs, i = [], 0
    while i < len(items):
        j = i + 1
        while enabled and j < len(items) and j - i < 4 and key(items[j]) == key(items[i]):
            j += 1
        runs.append((i, j))
        i = j
    lax = None
    if (beside and LAX_STREAM and enabled and K.LANE is None and K.LANES_ENABLED and any(j - i >= 2 for i, j in runs) and any(j - i == 1 for i, j in runs)
            and not torch._C._cuda_isCurrentStreamCapturing()):
        lax = lax_stream().cuda_stream
        tape.record(join_lax_stream)  # backward: runs AFTER the items' closures - whatever consumes their input gradients next is on the current stream
        K.stream_fork(K._stream(), lax)  # BEFORE the first single item's launches: the group must not queue behind them
    for i, j in runs:
        if j - i >= 2:
            with lane_group(tape, j - i, stream=lax) as grp:
                for lane, it in enumerate(items[i:j]):
                    grp.select(lane)
                    body(it)
        else:
            body(items[i])
    if lax is not None:
        K.stream_fork(lax, K._stream())
        _LAX_KEEP.clear()  # the next user of these buffers is ordered behind the join


def flush_ln(tape: "Tape") -> None:
    """Add the LayerNorm parameter-gradient partials collected so far into dgamma / dbeta (one batched launch)."""
    items, tape.pending_ln = tape.pending_ln, []
    K.ln_param_reduce_batched(items)


# --------------------------------------------------------------------------------------------------------------
# weight shadows
# --------------------------------------------------------------------------------------------------------------
def _flat_shadow(p: torch.nn.Parameter) -> torch.Tensor | None:
    """The optimiser-maintained bf16 copy of ``p`` (cinema_amd.optim.FlatModel), if it is still in sync with the master."""
    sh = getattr(p, "_cinema_shadow", None)
    if sh is not None and getattr(p, "_cinema_shadow_version", -1) == p._version:  # noqa: SLF001
        return sh
    return None


def _adjacent(a: torch.Tensor, b: torch.Tensor) -> bool:
    """b starts where a ends INSIDE ONE storage (two separate allocations can be address-adjacent by chance: a strided view
    across them is out of bounds for the first storage - an intermittent failure of model tests without a FlatModel)."""
    return (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr())


def w_plain(weight: torch.nn.Parameter) -> torch.Tensor:
    """(out, in[,1,1,1]) fp32 -> bf16 [out, in]."""
    sh = _flat_shadow(weight)
    if sh is not None:
        return sh.view(weight.shape[0], -1)
    return WEIGHTS.get((weight,), "plain", lambda: K.cast(weight.detach().reshape(weight.shape[0], -1), BF16))


# fp8 forward GEMMs (BASELINE config 5, "fp8 MFMA path"): the q / kv / proj / fc1 / fc2 projections of the transformer blocks multiply e4m3 operands
# (per-tensor current scaling: activations quantised right before the GEMM, weights once per optimiser step) on the MX-scaled MFMA; every backward
# GEMM stays bf16 on the bf16 activations / weights that are kept anyway.  Off by default (the BASELINE metric is quoted in bf16).
FP8_FORWARD = bool(int(os.environ.get("CINEMA_FP8", "0")))


# The DATA-gradient GEMMs of those projections in e4m3 as well (follows FP8_FORWARD unless CINEMA_FP8_DGRAD=0): dX = dY W with dY quantised per row
# (one launch per gradient tensor) and W from the TRANSPOSED e4m3 shadows (all 2-D weights in one launch per optimiser step).  Weight gradients stay
# bf16: their reduction runs over the tokens, so neither per-row activation scales nor the token-major layouts fit the e4m3 MFMA without transposed,
# re-scaled copies of every activation - measured not to pay at these sequence lengths.
FP8_DGRAD = bool(int(os.environ.get("CINEMA_FP8_DGRAD", "1")))


# WEIGHT gradients on e4m3 operands as well (follows FP8_FORWARD unless CINEMA_FP8_WGRAD=0).  dW = dY^T X reduces over the TOKENS, so per-token scales do not
# factor out: the 8-bit copies of the activations and gradients of the transformer MLP / attention output projection carry ONE scale per tensor, taken from
# the previous step's maximum (delayed scaling), and are written by the kernels that produce the tensors - LayerNorm forward (the normed rows), fc1's GELU
# epilogue, fc2's data-gradient epilogue (x GELU'), the LayerNorm backward that completes a residual-stream gradient - or, for attention outputs, by one
# stand-alone pass; the same copies feed the e4m3 forward and data-gradient GEMMs (no per-row quantisation launches for them).  The weight-gradient kernel
# reads the row-major copies directly (ds_read_b64_tr_b8), see csrc/gemm256.hip form 3.  A site has no scale in its first step: that step runs the per-row /
# bf16 forms and records maxima (``Fp8Sites.update`` at the end of every optimisation step turns them into scales).
FP8_WGRAD = bool(int(os.environ.get("CINEMA_FP8_WGRAD", "1")))
# In the steady state the fp8 path skips the bf16 tensors whose only readers are e4m3 GEMMs (GELU output, its gradient, the bf16 copy of a residual-stream gradient).
FP8_MARGIN = float(os.environ.get("CINEMA_FP8_MARGIN", "1.25"))  # scale = margin * amax / 448: headroom for a maximum that grows from one step to the next


class Fp8Sites:
    """Device arrays of the delayed-scaling sites (maxima [cap][4096 slots], scales [cap], inverse scales [cap]) and the registry of live sites.

    A site belongs to a PARAMETER OBJECT (the weight whose GEMM reads the 8-bit copy) and a tensor position: the site objects hang on the parameter
    (``param._cinema_q8[(device, kind)]``), a ``weakref.finalize`` on the parameter hands the array slot back when the parameter dies.  (Until round 5
    the registry was keyed by ``id(parameter)``: a new model whose parameter re-used a dead parameter's id inherited a READY site with a foreign scale -
    no calibration step, saturated or flushed first steps - and a process that built a few models ran out of the 1024 slots.)"""

    CAP = 1024
    SLOTS = 4096  # CINEMA_Q8_SLOTS

    def __init__(self, device: torch.device) -> None:
        self.device = device
        self.amax = K.persistent(lambda: torch.zeros(self.CAP * self.SLOTS, dtype=torch.int32, device=device))
        self.scale = K.persistent(lambda: torch.ones(self.CAP, dtype=F32, device=device))
        self.inv = K.persistent(lambda: torch.ones(self.CAP, dtype=F32, device=device))
        self.live: dict = {}     # slot -> Q8Site of a live owner
        self.free: list = []     # slots of dead owners, re-used before the high-water mark grows
        self.n_alloc = 0         # high-water mark: slots [0, n_alloc) are swept by update()
        self.updates = 0

    def _release(self, slots: list) -> None:
        for i in slots:
            if self.live.pop(i, None) is not None:
                self.free.append(i)

    class _Named:  # owner object of a site asked for by a plain key (tests / tools): lives as long as the registry
        pass

    def site(self, owner: object, kind: str | None = None) -> K.Q8Site:
        if kind is None:  # ``site(key)``: a stand-alone site under a hashable key
            named = self.__dict__.setdefault("_named", {})
            owner, kind = named.setdefault(owner, Fp8Sites._Named()), "named"
        sites = owner.__dict__.get("_cinema_q8")
        if sites is None:
            sites = owner.__dict__["_cinema_q8"] = {}
        key = (self.device.index, kind)
        st = sites.get(key)
        if st is not None and st.owner is self:
            return st
        if self.free:
            i = self.free.pop()
            self.amax[i * self.SLOTS:(i + 1) * self.SLOTS].zero_()  # (swept every update; a maximum the dead owner recorded in THIS step must not leak)
            self.scale[i:i + 1].fill_(1.0)
            self.inv[i:i + 1].fill_(1.0)
        else:
            i = self.n_alloc
            if i >= self.CAP:
                raise RuntimeError(f"Fp8Sites: more than {self.CAP} live delayed-scaling sites on {self.device}")
            self.n_alloc += 1
        st = sites[key] = self.live[i] = K.Q8Site(self.scale[i:i + 1], self.inv[i:i + 1], self.amax[i * self.SLOTS:(i + 1) * self.SLOTS], self, self.updates)
        # one slot list and one finalizer per (owner, REGISTRY): a list shared between registries (the same parameter used on a second device, or under a second
        # Fp8Sites instance) would release that registry's indices into this one when the owner dies, freeing a slot that belongs to a live parameter here
        by_reg = owner.__dict__.setdefault("_cinema_q8_slots", {})
        slots = by_reg.get(id(self))
        if slots is None:
            slots = by_reg[id(self)] = []
            weakref.finalize(owner, Fp8Sites._release, self, slots)
        slots.append(i)
        return st

    def update(self) -> None:
        """Maxima recorded since the last call -> scales (one launch); sites created before this call have a scale afterwards."""
        if self.n_alloc:
            K.fp8_sites_update(self.amax, self.scale, self.inv, self.n_alloc, FP8_MARGIN)
        self.updates += 1

    def all_ready(self) -> bool:
        return bool(self.live) and all(s.ready for s in self.live.values())


_FP8_SITES: dict = {}


def fp8_sites(device: torch.device) -> Fp8Sites:
    st = _FP8_SITES.get(device.index)
    if st is None:
        st = _FP8_SITES[device.index] = Fp8Sites(device)
    return st


def fp8_site(t: torch.Tensor, owner: object, kind: str):  # noqa: ANN201
    """The delayed-scaling site of tensor position ``kind`` at parameter ``owner`` (None when the fp8 weight-gradient path is off / on the CPU)."""
    if not (FP8_FORWARD and FP8_WGRAD and t.is_cuda):
        return None
    return fp8_sites(t.device).site(owner, kind)


def fp8_step_end() -> None:
    """End of an optimisation step (``TrainStep`` calls it after the backward pass): recorded maxima become the next step's scales."""
    for st in _FP8_SITES.values():
        st.update()


def fp8_calibrating() -> bool:
    """True while a recorded / replayed step must not be taken yet: the fp8 weight-gradient path is on and some site has no scale (first step)."""
    if not (FP8_FORWARD and FP8_WGRAD):
        return False
    return not _FP8_SITES or any(not st.all_ready() for st in _FP8_SITES.values())


def _tensor_scaled(q8: tuple | None) -> bool:
    return q8 is not None and q8[1].numel() == 1


def wgrad8_problem(tape: Tape, dy8: tuple, x8: tuple, dst: torch.Tensor, dy16: torch.Tensor, bias_grad: torch.Tensor | None) -> None:
    """dst[n, k] += dY^T X on the e4m3 copies (per-tensor scales); bias_grad[n] += column sums of the bf16 dY.  Deferred to the enclosing weight-gradient
    group like the bf16 problems (one persistent launch per block for all of them)."""
    item = (dy8[0], dy8[1], x8[0], x8[1], dst, dy16, bias_grad)
    if getattr(tape, "grouping", False):
        tape.pending_wgrads8.append(item)
    else:
        _wgrad8_launch([item])


def _wgrad8_launch(items: list) -> None:
    def run() -> None:
        K.gemm_fp8_wgrad_grouped([it[:5] for it in items])
        for it in items:
            if it[6] is not None:
                K.colsum(it[5], it[6])
    _wgrad_launch(run, *[t for it in items for t in (it[0], it[2], it[5]) if t is not None],
                  keys=tuple(t.data_ptr() for it in items for t in (it[4], it[6]) if t is not None))


def w_fp8_t(weight: torch.nn.Parameter) -> tuple | None:
    """(uint8 [in, out] transposed e4m3 shadow, fp32 [1] scale) of a 2-D Linear weight, or None."""
    flat = getattr(weight, "_cinema_flat", None)
    if flat is None or weight.dim() != 2:
        return None
    off = flat.offsets.get(id(weight))
    if off is None or off[0] % 8:
        return None
    hit = flat.fp8_shadow(weight, transposed=True)
    return None if hit is None else (hit[0].view(weight.shape[1], weight.shape[0]), hit[1])


def dgrad(dy16: torch.Tensor, weight: torch.nn.Parameter, w16: torch.Tensor, *, gelu_in: torch.Tensor | None = None, row_mask: torch.Tensor | None = None,
          fp8: bool = False, dy8: tuple | None = None, out_f32_residual: torch.Tensor | None = None, gelu_deriv: bool = False, out8: tuple | None = None,
          colsum_partials: torch.Tensor | None = None) -> torch.Tensor:
    """dX = dY W (x GELU'(gelu_in), or x gelu_in itself when it already holds the derivative: ``gelu_deriv``): bf16 MFMA GEMM, or - ``fp8`` and the shapes allow it - the e4m3 GEMM on per-row quantised dY (``dy8`` = an already
    quantised (bytes, row scales) pair of the same rows, e.g. a column slice of a fused gradient) and the transposed weight shadow.
    ``out_f32_residual``: fp8 path only, adds an fp32 tensor and returns fp32 (two weights fed by column blocks of one gradient)."""
    if fp8 and FP8_FORWARD and FP8_DGRAD and row_mask is None and (dy8 is not None or (dy16.is_cuda and dy16.is_contiguous())) and weight.shape[0] % 16 == 0 and weight.shape[1] % 8 == 0:
        wt = w_fp8_t(weight)
        if wt is not None:
            a8, sa = dy8 if dy8 is not None else K.quantize_fp8_rows(dy16)
            return K.gemm_fp8(a8, sa, wt[0], wt[1], gelu_in=gelu_in, residual=out_f32_residual, out_dtype=F32 if out_f32_residual is not None else BF16,
                              gelu_deriv=gelu_deriv, out8=out8 if out_f32_residual is None else None, colsum_partials=colsum_partials)
    if out_f32_residual is not None:
        raise RuntimeError("dgrad: the fp32-residual form exists on the fp8 path only")
    if dy16 is None:  # an 8-bit-only gradient on a path that turned out to need bf16
        dy16 = K.dequantize_fp8(dy8)
    return K.gemm(dy16, w16, a_kmajor=True, b_kmajor=False, gelu_in=gelu_in, row_mask=row_mask, gelu_deriv=gelu_deriv,
                  out8=out8 if (out8 is not None and row_mask is None and dy16.is_cuda and not K.FORCE_GENERIC) else None, colsum_partials=colsum_partials)


def w_fp8(weight: torch.nn.Parameter) -> tuple:
    """(uint8 [out, in] e4m3 shadow, fp32 [1] scale) of a Linear weight."""
    flat = getattr(weight, "_cinema_flat", None)
    hit = flat.fp8_shadow(weight) if flat is not None else None
    if hit is not None:
        return hit[0].view(weight.shape[0], -1), hit[1]
    return WEIGHTS.get((weight,), "fp8", lambda: K.quantize_fp8(w_plain(weight).contiguous()))


def _fp8_ok(x: torch.Tensor, *weights: torch.nn.Parameter) -> bool:
    return (FP8_FORWARD and x.is_cuda and x.dtype == BF16 and x.is_contiguous() and x.shape[1] % 16 == 0 and x.numel() % 8 == 0
            and all(w.shape[0] % 8 == 0 and math.prod(w.shape[1:]) % 16 == 0 for w in weights))


def a_fp8(x: Var) -> tuple:
    """(e4m3 rows, per-row scales) of a bf16 activation: the producer's copy when it made one (LayerNorm), else one per-row quantisation launch."""
    return x.fp8 if x.fp8 is not None else K.quantize_fp8_rows(x.data)


def _hip_layout(fn: Callable, jmap: torch.Tensor | None) -> Callable:
    """Tag a to-parameter-layout function with what the HIP re-layout kernel needs to add the gradient straight into the flat buffer."""
    fn.hip_relayout = (jmap,)
    return fn


def w_patch(weight: torch.nn.Parameter) -> torch.Tensor:
    """k==s conv weight (out, c, *k) -> bf16 [out, (*k, c)] matching the patch-gather feature order (one re-layout kernel)."""
    return WEIGHTS.get((weight,), "patch", lambda: K.patch_weight_rows(weight.detach()))


def patch_grad_to_param(weight: torch.nn.Parameter) -> Callable:
    shape = weight.shape

    def conv(g: torch.Tensor) -> torch.Tensor:
        nd = len(shape) - 2
        g = g.reshape(shape[0], *shape[2:], shape[1])
        return g.permute(0, nd + 1, *range(1, nd + 1)).contiguous()

    return _hip_layout(conv, None)


def w_patch_perm(weight: torch.nn.Parameter, inv_pos: torch.Tensor) -> torch.Tensor:
    """Like :func:`w_patch` for rows whose patch voxels are stored in a permuted order: feature block q of a row holds
    the voxel with raster index ``inv_pos[q]`` (the visible-voxel stem keeps coarser-stage children contiguous)."""
    return WEIGHTS.get((weight,), "patch_perm", lambda: K.patch_weight_rows(weight.detach(), jmap=inv_pos))


def patch_grad_to_param_perm(weight: torch.nn.Parameter, pos: torch.Tensor, inv_pos: torch.Tensor) -> Callable:
    shape = weight.shape
    base = patch_grad_to_param(weight)

    def conv(g: torch.Tensor) -> torch.Tensor:
        g3 = g.reshape(shape[0], -1, shape[1])
        return base(g3[:, pos.long(), :].contiguous())  # raster voxel u sits at row block pos[u]

    return _hip_layout(conv, inv_pos)


def w_conv_same(weight: torch.nn.Parameter) -> torch.Tensor:
    """Dense conv weight (out, c, *k) -> bf16 [out, ld] in the im2col feature order (*k, c), zero-padded to ld = ceil(taps*c / 8) * 8."""
    return WEIGHTS.get((weight,), "conv_same", lambda: K.patch_weight_rows(weight.detach(), pad_to=8))


def conv_same_grad_to_param(weight: torch.nn.Parameter) -> Callable:
    shape = weight.shape
    base = patch_grad_to_param(weight)

    def conv(g: torch.Tensor) -> torch.Tensor:
        f = math.prod(shape[1:])
        return base(g[:, :f].contiguous())

    return _hip_layout(conv, None)


def w_cat(weights: tuple) -> torch.Tensor:
    """Row-concatenation of Linear weights as one bf16 [sum(out), in] operand; a zero-copy view when the flat shadows are adjacent."""
    shs = [_flat_shadow(w) for w in weights]
    if all(s is not None for s in shs) and all(_adjacent(a, b) for a, b in zip(shs, shs[1:])):
        k = weights[0].shape[1]
        return shs[0].as_strided((sum(w.shape[0] for w in weights), k), (k, 1))
    return WEIGHTS.get(tuple(weights), "cat", lambda: torch.cat([K.cast(w.detach().reshape(w.shape[0], -1), BF16) for w in weights], dim=0))


def b_cat(biases: tuple) -> torch.Tensor:
    ds = [b.detach() for b in biases]
    if all(_adjacent(a, b) for a, b in zip(ds, ds[1:])):
        return ds[0].as_strided((sum(b.numel() for b in ds),), (1,))
    return WEIGHTS.get(tuple(biases), "bcat", lambda: torch.cat(ds, dim=0))


def trainable_params(module: torch.nn.Module) -> list:
    """The module's trainable parameters without walking the module tree on every step (the tree is static; requires_grad is not)."""
    pl = module.__dict__.get("_cinema_param_list")
    if pl is None:
        pl = module.__dict__["_cinema_param_list"] = list(module.parameters())
    return [p for p in pl if p.requires_grad]


SPLITK_SLOTS = 256
SPLITK_MIN_ROWS = 512


def _split_k(m_red: int, n_out: int, k_out: int) -> int:
    """k-slices of a weight gradient on the 128x128 kernel (the stem / fusion / head GEMMs: small outputs, reductions of 9 k - 147 k rows).  Half a round of
    workgroup slots and at least 512 rows per slice - measured with the slab reduce behind it (tools/bench_skinny_wgrad.py): dW[64x64] over 147456 rows
    29.4 us at 512 slices, 24.3 at 256; dW[256x64] 39.6 -> 32.4; dW[128x128] over 36864 rows 19.5 at 144, 18.4 at 64; dW[128x512] 30.1 -> 27.4."""
    tiles = ((n_out + 127) // 128) * ((k_out + 127) // 128)
    want = max(1, SPLITK_SLOTS // tiles)
    return max(1, min(want, (m_red + SPLITK_MIN_ROWS - 1) // SPLITK_MIN_ROWS))


CONV_SPLITK_SLOTS = 256


def _split_k_conv(m_red: int, n_out: int, k_out: int) -> int:
    """k-slices of the implicit-convolution weight gradient (CINEMA_CONV_SPLITK_SLOTS workgroup slots).  ALONE these GEMMs want a full round of the 128x128
    kernel's slots (512: the second resident workgroup covers the gathered operand's DMA latency - 1113 -> 604 us at the 32-channel level, 622 -> 340, 500 -> 301,
    545 -> 304 us at the others, tools/bench_conv.py, profiles/r04_p_conv_split.txt; the single-stream step 53.8 -> 49.7 ms).  In the product step they run on the
    weight-gradient stream BESIDE the main stream's kernels, which already take the slots a half round leaves free: 47.5 ms with 256 slots, 47.9 with 512
    (profiles/r04_q_seg_split_ab.txt) - so the half round stays."""
    tiles = ((n_out + 127) // 128) * ((k_out + 127) // 128)
    want = max(1, CONV_SPLITK_SLOTS // tiles)
    return max(1, min(want, (m_red + SPLITK_MIN_ROWS - 1) // SPLITK_MIN_ROWS))


def _wgrad_launch(fn: Callable, *operands: torch.Tensor, alt: int = 0, keys: tuple = ()) -> None:
    """Run a weight-gradient launch on the side stream (after everything queued on the main stream so far) or inline.  ``operands``
    are the activation / gradient tensors the launch reads: they were allocated on the main stream, so they are kept alive until the
    backward pass joins the side stream (a closure may drop its last reference right away and the allocator would reuse the memory)."""
    if SIDE_WGRAD and operands[0].is_cuda:
        # alt = 1: a single weight gradient - dealt alternately to the two streams; alt = 2: follows the last such launch (its re-layout); 0: stream 0 (groups)
        if alt == 1 and SIDE_STREAMS > 1 and K.LANE is None:
            _SIDE_ALT[0] ^= 1
        idx = _SIDE_ALT[0] if (alt and SIDE_STREAMS > 1 and K.LANE is None) else 0
        if SIDE_STREAMS > 1:
            # ``keys``: the buffers this launch ADDS into.  One written earlier in this pass from the other stream: a single GEMM moves over to that stream;
            # anything else (a re-layout tied to its GEMM's stream, a group) makes its stream wait for the other one first
            prev = {_DST_STREAM[k] for k in keys if k in _DST_STREAM}
            if alt == 1 and len(prev) == 1 and K.LANE is None:
                idx = _SIDE_ALT[0] = prev.pop()
            for other in prev:
                if other != idx:
                    K.stream_fork(side_stream(other).cuda_stream, side_stream(idx).cuda_stream)
            for k in keys:
                _DST_STREAM[k] = idx
        side = side_stream(idx).cuda_stream
        K.stream_fork(K._stream(), side)
        with K.on_stream(side):  # raw redirection: no torch stream context, no event objects (this runs ~200x per step)
            fn()
        if K.RECORD is not None or K.LANE is not None or torch._C._cuda_isCurrentStreamCapturing():  # recorded / captured step (or a lane group, whose
            # launches go out later): no completion queries at replay time, so the
            # allocator must not reuse an operand before the join (which is itself part of the recording)
            _SIDE_KEEP.append((None, operands))
            return
        _SIDE_KEEP.append((K.marker_record(side), operands))  # cheaper than record_stream (allocator events on every free of these blocks)
        while _SIDE_KEEP and (len(_SIDE_KEEP) > 2048 or _SIDE_KEEP[0][0] is None or K.marker_done(_SIDE_KEEP[0][0])):
            if len(_SIDE_KEEP) > 2048:                # finished launches give their operands back (holding everything to the end of the
                torch.cuda.synchronize()              # backward pass kept ~4 GB more live and cost 2 ms/step of cache locality); the
            _SIDE_KEEP.popleft()                      # marker ring holds 4096 tickets
        return
    fn()


# The weight gradients of one transformer block in ONE launch (cinema_gemm_bf16_grouped): alone, each has 36-144 output tiles for 512
# workgroup slots and is cut into k-slices with fp32 slabs and a reduce launch; together the encoder block's four have 432 whole-K tiles.
# Measured (tools/wgrad_group_ab.py): in isolation 223 us instead of 308 us for the four launches - but the STEP gets slower (32.85 vs
# 32.39 ms): the group can only be issued at the end of the block's backward and its 432 workgroups each run for 160 us, so the side
# stream no longer fills the main stream's idle slots early and the main stream's kernels wait for slots behind long tiles.  Opt-in.
# Groups that would leave the slots mostly empty (decoder blocks: 192 tiles) keep the per-GEMM split-K path either way.
GROUP_WGRAD = 2  # 1: whole-K 128x128 tiles (cinema_gemm_bf16_grouped), 2: persistent 256x256 kernel, k-slices finished in the launch (tests compare the two)
_GROUP_MIN_TILES = 384
# A persistent launch is held back until GROUP_FLUSH_MIN problems OR GROUP_FLUSH_GFLOP of work are pending.  Round 3 (main-loop form 1): per block 28.67,
# two encoder blocks 28.81 ms/step.  Under form 2 (round 4) two ViT-Base blocks per launch (8 problems, 310-344 GFLOP) are 0.15 ms FASTER in every round of two A/Bs (26.17 -> 26.04, 26.91 -> 26.75 ms,
# profiles/r04_ba_knobs.txt, r04_bb_flush_min.txt), three or more slower (27.4-27.9); a ViT-Large block (319 GFLOP) must go out alone (config 5: 58.9 -> 60.1 ms with two per launch,
# profiles/r04_bc_flush_min_cfg45.txt) - hence the work threshold.  The same schedule runs under a gradient exchange: the collective of a block whose group is held
# back starts behind the shared launch (mark_params / flush_wgrads).  GROUP_FLUSH_MIN = 1 is one launch per block.
GROUP_FLUSH_MIN = 8
GROUP_FLUSH_GFLOP = 300.0
P256_MAX_PROBLEMS = 12


def _pending_gflop(tape: Tape) -> float:
    return (sum(2.0 * dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in tape.pending_wgrads)
            + sum(2.0 * it[0].shape[0] * it[0].shape[1] * it[2].shape[1] for it in tape.pending_wgrads8)) * 1e-9


def wgrad_group(tape: Tape) -> None:
    """Bracket the forward ops of a module whose weight gradients should be launched together: call at the START of its forward ops (the
    flush runs at the end of its backward) ... and :func:`wgrad_group_end` at the end of them (grouping switches on when the backward enters)."""
    def flush() -> None:
        # the persistent kernel balances better and writes fewer partial tiles with more problems per launch: without a gradient exchange waiting for
        # this block's range, the launch is held back until GROUP_FLUSH_MIN problems (two transformer blocks) are pending
        if (GROUP_WGRAD != 2 or len(tape.pending_wgrads) + len(tape.pending_wgrads8) >= GROUP_FLUSH_MIN or _pending_gflop(tape) >= GROUP_FLUSH_GFLOP):
            flush_wgrads(tape)
        tape.grouping = False

    tape.record(flush)


def wgrad_group_end(tape: Tape) -> None:
    def begin() -> None:
        tape.grouping = bool(GROUP_WGRAD) and not K.FORCE_GENERIC

    tape.record(begin)


def _wgrad_single(dy16: torch.Tensor, x16: torch.Tensor, dst: torch.Tensor, bias_grad: torch.Tensor | None) -> None:
    n, k = dy16.shape[1], x16.shape[1]
    _wgrad_launch(lambda: K.gemm(dy16, x16, a_kmajor=False, b_kmajor=False, out=dst, accumulate=True, split_k=_split_k(dy16.shape[0], n, k),
                                 a_rowsum=bias_grad), dy16, x16, alt=1, keys=(dst.data_ptr(),) if bias_grad is None else (dst.data_ptr(), bias_grad.data_ptr()))


def flush_wgrads(tape: Tape) -> None:
    _flush_wgrads(tape)
    hooks, tape.held_hooks = tape.held_hooks, []
    for h in hooks:  # the gradient exchange of the blocks whose weight gradients just went out
        h()


def _flush_wgrads(tape: Tape) -> None:
    probs8, tape.pending_wgrads8 = tape.pending_wgrads8, []
    for i in range(0, len(probs8), P256_MAX_PROBLEMS):
        _wgrad8_launch(probs8[i:i + P256_MAX_PROBLEMS])
    probs, tape.pending_wgrads = tape.pending_wgrads, []
    if not probs:
        return
    tiles = sum(((dy.shape[1] + 127) // 128) * ((x.shape[1] + 127) // 128) for dy, x, _, _ in probs)
    p256 = GROUP_WGRAD == 2
    if not p256 and (len(probs) < 2 or tiles < _GROUP_MIN_TILES or len({dy.shape[0] for dy, _, _, _ in probs}) > 1):
        for pr in probs:
            _wgrad_single(*pr)
        return
    step = P256_MAX_PROBLEMS if p256 else 8
    for i in range(0, len(probs), step):
        chunk = probs[i:i + step]
        _wgrad_launch(lambda c=chunk: K.gemm_wgrad_grouped(c, p256=p256), *[t for dy, x, _, _ in chunk for t in (dy, x)],
                      keys=tuple(t.data_ptr() for _, _, d, b in chunk for t in (d, b) if t is not None))


def wgrad_problem(tape: Tape, dy16: torch.Tensor, x16: torch.Tensor, dst: torch.Tensor, bias_grad: torch.Tensor | None) -> None:
    """dst[n,k] += dy^T x ; bias_grad[n] += colsum(dy): launched now (split-K) or deferred to the enclosing weight-gradient group."""
    n, k = dy16.shape[1], x16.shape[1]
    ok = (n % 8 == 0 and k % 8 == 0 and dy16.stride(1) == 1 and x16.stride(1) == 1 and dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0
          and dst.stride(0) % 8 == 0 and dy16.data_ptr() % 16 == 0 and x16.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0)
    if getattr(tape, "grouping", False) and ok:
        tape.pending_wgrads.append((dy16, x16, dst, bias_grad))
    else:
        _wgrad_single(dy16, x16, dst, bias_grad)


def wgrad(tape: Tape, dy16: torch.Tensor, x16: torch.Tensor, wv: PVar, bv: PVar | None, wshape: tuple, to_param_layout: Callable | None = None,
          row_offset: int = 0, total_rows: int | None = None) -> None:
    """dW[n,k] += dy^T x ; db[n] += colsum(dy).  ``row_offset`` targets a row block of a fused (cat) weight."""
    n, k = dy16.shape[1], x16.shape[1]
    full = wv.grad_buffer(wshape if total_rows is None else (total_rows, k), to_param_layout)
    dst = full.view(-1, k)[row_offset:row_offset + n]
    bias_grad = None if bv is None else bv.grad_buffer((n,) if total_rows is None else (total_rows,))[row_offset:row_offset + n]
    wgrad_problem(tape, dy16, x16, dst, bias_grad)
    # A weight whose gradient is computed in the KERNEL's layout (k == s patch convolutions, transposed convolutions) is added into the flat gradient buffer by
    # one re-layout launch.  Until round 5 all of them ran at the very end of the backward pass, behind the join of the streams - 16 serial launches, 0.24 ms
    # of a step with nothing else left to run (profiles/r05_l_phase_timeline.txt); now each follows its own weight-gradient launch on the weight-gradient
    # stream and is hidden behind the rest of the backward pass.  The kernel-layout buffer is dropped afterwards (a second use of the weight starts a new one).
    tagged = getattr(wv.to_param_layout, "hip_relayout", None)
    p = wv.param
    flat = getattr(p, "_cinema_flat_grad", None)
    if (tagged is not None and total_rows is None and row_offset == 0 and not wv.direct and flat is not None and dst.is_cuda
            and not getattr(tape, "grouping", False) and flat.data_ptr() == getattr(p.grad, "data_ptr", lambda: 0)()):
        rows = wv.grad
        if tagged[0] == "convt":
            _wgrad_launch(lambda: K.convt_weight_grad_accumulate(rows.contiguous(), flat.view(p.shape)), rows, alt=2, keys=(flat.data_ptr(),))
        else:
            _wgrad_launch(lambda: K.patch_weight_grad_accumulate(rows.view(p.shape[0], -1), flat.view(p.shape), tagged[0]), rows, alt=2, keys=(flat.data_ptr(),))
        wv.grad = None


ATTN_CAPTURE: list | None = None  # set to a list: every self-attention backward appends its operands (dev tooling only)
# training keeps the second bf16 half of every attention output, O = o + o_lo, for delta = rowsum(dO O) of the backward pass (csrc/attention.hip attn_bwd_dq_mfma:
# delta from the bf16 output alone put 13 % of error into dQ of the late ViT-Large blocks); 0: one half (A/B, tools/attn_dq_error.py)
ATTN_O_LO = True  # False: delta from the bf16 output alone (tools/attn_dq_error.py and the depth-parity test compare the two)


FUSED_STEM = bool(int(os.environ.get("CINEMA_FUSED_STEM", "1")))  # 0: the MaskedConvBlock as separate LayerNorm / GEMM launches (A/B, and the form every other channel count takes)


# --------------------------------------------------------------------------------------------------------------
# torch.autograd bridge: one node per top-level call
# --------------------------------------------------------------------------------------------------------------
class _TapedCall(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, n_inputs, params, grad_mode, *tensors):  # noqa: ANN001, ANN205
        inputs = tensors[:n_inputs]
        train = grad_mode and any(t.requires_grad for t in tensors if isinstance(t, torch.Tensor))
        tape = Tape(train=train)
        in_vars = [Var(t, needs_grad=bool(t.requires_grad)) if isinstance(t, torch.Tensor) else t for t in inputs]
        out_vars, extras = runner(tape, *in_vars)
        ctx.tape, ctx.in_vars, ctx.out_vars, ctx.params, ctx.n_extras = tape, in_vars, out_vars, params, len(extras)
        outs = tuple(v.data for v in out_vars) + tuple(extras)
        ctx.mark_non_differentiable(*[e for e in extras if isinstance(e, torch.Tensor)])
        return outs

    @staticmethod
    def backward(ctx, *grads):  # noqa: ANN001, ANN205
        return _TapedCall._backward(ctx, *grads)

    @staticmethod
    def _backward(ctx, *grads):  # noqa: ANN001, ANN205
        for v, g in zip(ctx.out_vars, grads):
            if g is not None:
                v.grad = g.contiguous().to(v.data.dtype).reshape(v.data.shape)
        ctx.tape.backward()
        in_grads = [v.grad.reshape(v.data.shape) if (isinstance(v, Var) and v.needs_grad and v.grad is not None) else None for v in ctx.in_vars]
        p_grads = []
        for p in ctx.params:
            pv = ctx.tape.pvars.get(id(p))
            flat = getattr(p, "_cinema_flat_grad", None)
            in_flat = flat is not None and flat.data_ptr() == getattr(p.grad, "data_ptr", lambda: 0)()
            tagged = getattr(pv.to_param_layout, "hip_relayout", None) if pv is not None else None
            if tagged is not None and in_flat and pv.grad is not None and not pv.direct and p.requires_grad:
                if tagged[0] == "convt":  # transposed-conv rows [(kv, c_out), c_in] -> (c_in, c_out, *k)
                    K.convt_weight_grad_accumulate(pv.grad.contiguous(), flat.view(p.shape))
                else:
                    K.patch_weight_grad_accumulate(pv.grad.view(p.shape[0], -1), flat.view(p.shape), tagged[0])  # re-layout + add, one kernel
                p_grads.append(None)
                continue
            g = pv.final_grad() if (pv is not None and p.requires_grad) else None
            if g is not None and flat is not None and flat.data_ptr() == getattr(p.grad, "data_ptr", lambda: 0)():
                # the optimiser's flat buffer: add here, on the current stream, instead of through autograd's AccumulateGrad node (which
                # runs on the parameter's creation stream - a cross-stream launch that breaks HIP-graph capture of the step)
                n = g.shape[-1] if g.dim() > 1 else g.numel()
                K.row_copy(flat.view(-1, n), g.contiguous().view(-1, n), accumulate=True)
                g = None
            p_grads.append(g)
        ctx.tape = None
        return (None, None, None, None, *in_grads, *p_grads)


class _DirectCall:
    """The context of a top-level call run WITHOUT autograd (recorded steps): same fields as the autograd context of _TapedCall."""

    def __init__(self, runner: Callable, inputs: list, params: list) -> None:
        self.tape = Tape(train=True)
        self.in_vars = [Var(t, needs_grad=False) if isinstance(t, torch.Tensor) else t for t in inputs]
        self.out_vars, extras = runner(self.tape, *self.in_vars)
        self.params, self.n_extras = params, len(extras)
        self.outputs = tuple(v.data for v in self.out_vars) + tuple(extras)

    def backward(self, *grads: torch.Tensor) -> None:
        left = [g for g in _TapedCall._backward(self, *grads)[4:] if g is not None]  # noqa: SLF001
        if left:  # a gradient that did not go into the optimiser's flat buffer would need autograd's accumulation
            raise RuntimeError("recorded steps need every trainable parameter in a FlatModel gradient buffer")


def taped_call(runner: Callable, inputs: list, params: list) -> tuple:
    """Run ``runner(tape, *input_vars) -> (out_vars, extra_tensors)`` as a single autograd node.

    Returns the output tensors (differentiable) followed by the extras (non-differentiable).  While a step is being recorded
    (cinema_amd/replay.py) the call runs without autograd - the engine would run the backward pass on its own thread, outside the
    recording's memory pool - and the recorder calls ``REC_CALL.backward`` itself.
    """
    global REC_CALL  # noqa: PLW0603
    params = [p for p in params if p is not None]
    if K.RECORD is not None:
        REC_CALL = _DirectCall(runner, inputs, params)
        return REC_CALL.outputs
    return _TapedCall.apply(runner, len(inputs), params, torch.is_grad_enabled(), *inputs, *params)


# ---- the ops, one module per part of the model, re-exported: callers write tape.op_* --------------------------------------------------------------------
from cinema_amd.tape.ops_block import *  # noqa: E402, F401, F403
from cinema_amd.tape.ops_conv import *  # noqa: E402, F401, F403
from cinema_amd.tape.ops_rows import *  # noqa: E402, F401, F403
from cinema_amd.tape.ops_options import *  # noqa: E402, F401, F403
