"""Tensor-level launchers for the HIP operators (thin: shape/stride plumbing only).

Ownership follows the reference's kernels (kernels/geglu.py:31,
kernels/groupnorm.py:136-138, kernels/linear.py:193): the callee allocates and
returns a fresh tensor from torch's caching allocator, inputs are never
mutated, parameters are borrowed from the live module at call time.  Every
launch goes to torch's current stream, so the ops are capturable.
"""
from __future__ import annotations

import threading
from typing import Optional

import torch

from . import _C
from ._C import BackendError

# ---- execution context: the mutable host state of one compiled module ------------------------------
# split-K scratch for st_linear / st_conv2d: a fixed-size slab, allocated once per context and never
# replaced (captured graphs keep its address); problems that would need more run unsplit.  Concurrent
# launches must not share one (the arrival counters), so every compiled module / stream has its own.
GEMM_WORKSPACE_BYTES = 192 << 20


class WeightPlan:
    """Next-weights hints.  Every denoise step launches the same GEMMs over the same weights in the same
    order, and every one of them finds its weights cold (5 GB of weights stream through per step).  The
    first pass after `begin()` records the order; later passes tell each launch which weights the launch
    after it reads (the `next_weights` argument of st_linear / st_ln_linear / st_conv2d), and that launch
    touches them.  Entries hold the tensors themselves, so a hinted buffer cannot be freed under a
    captured graph that still touches it."""

    def __init__(self):
        self.entries = []            # weight tensors in launch order (strong references)
        self.pos = 0
        self.state = "off"           # off | record | replay

    def begin(self) -> None:
        if self.state == "off":
            self.state = "record"
            self.entries.clear()
        elif self.state == "record" and self.pos > 0:
            self.state = "replay"
        self.pos = 0

    def reset(self) -> None:
        self.state, self.pos = "off", 0
        self.entries.clear()

    def next_after(self, w: torch.Tensor):
        """Called by the GEMM launchers with the weights they are about to read; returns the tensor the
        following launch reads (or None)."""
        if self.state == "off":
            return None
        if self.state == "record":
            self.entries.append(w)
            self.pos += 1
            return None
        i = self.pos
        cur = self.entries[i] if i < len(self.entries) else None
        if cur is None or cur.data_ptr() != w.data_ptr() or cur.numel() != w.numel():
            self.reset()             # the launch sequence changed: stop hinting
            return None
        self.pos += 1
        return self.entries[(i + 1) % len(self.entries)]


class DerivedWeights:
    """A weight buffer computed from module parameters (row-concatenated q|k|v, gamma-folded LayerNorm
    projections).  `refresh()` recomputes it IN PLACE when a source parameter changed, so addresses held
    by captured graphs stay valid and in-place weight updates (LoRA merges) become visible."""

    def __init__(self, sources, compute):
        self.sources = list(sources)
        self.compute = compute
        self.versions = self._versions()
        self.stamp = self._stamp()
        self.value = compute()

    def _versions(self):
        return [t._version for t in self.sources]

    def _stamp(self):
        return tuple((t.data_ptr(), t.dtype) for t in self.sources)

    def refresh(self, full: bool = True) -> bool:
        """`full` also compares storage addresses (`param.data = ...` swaps); the quick form only looks at
        the in-place version counters (copy_/add_/load_state_dict), which is what a per-replay check can afford."""
        ver = self._versions()
        st = self._stamp() if full else self.stamp
        if ver == self.versions and st == self.stamp:
            return False
        self.versions = ver
        new = self.compute()
        for dst, src in zip(self.value, new):
            if dst is not None:
                dst.copy_(src)
        self.stamp = st
        return True


class ExecContext:
    """Host state owned by ONE compiled module (or DenoiseLoop): its split-K workspace, its weight plan
    and its derived weight buffers.  `with ctx:` makes it current for the calling thread.  Launchers called
    outside any context share one default context per device (no weight plan): callers that launch from
    several streams at once give each stream its own ExecContext."""

    def __init__(self, hints: bool = True):
        self.plan = WeightPlan() if hints else None
        self.hinting = False         # True only inside step(): one-off passes (context / time tables) stay out of the plan
        self._ws = {}
        self.derived = {}
        self.fp8 = None              # Fp8Scales of the fp8 plan (delayed per-tensor scaling), created on first use

    def gemm_workspace(self, device: torch.device) -> torch.Tensor:
        key = (device.type, device.index)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.zeros(GEMM_WORKSPACE_BYTES, dtype=torch.uint8, device=device)      # the arrival counters must start at zero
            self._ws[key] = ws
        return ws

    def derived_weights(self, key, sources, compute) -> "DerivedWeights":
        """The buffer derived from `sources` (parameters), built on first use.  Outside graph capture a
        stale buffer is refreshed on the spot; under a captured graph the owner calls refresh_derived()."""
        d = self.derived.get(key)
        if d is None or len(d.sources) != len(sources) or any(a is not b for a, b in zip(d.sources, sources)):
            d = self.derived[key] = DerivedWeights(sources, compute)      # first use (or the key's modules were replaced)
        elif sources[0].device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            d.refresh()
        return d

    def refresh_derived(self, full: bool = False) -> int:
        """Re-derive (in place) every buffer whose source parameters changed; returns how many did.  Captured
        graphs read the derived buffers by address and never re-run the wrappers that build them, so their owners
        call this before a replay (GraphedCallable: quick check every call; DenoiseLoop / hooks: full check per prompt)."""
        return sum(1 for d in self.derived.values() if d.refresh(full))

    def step(self) -> "_StepScope":
        """`with ctx.step():` around every pass of a loop that repeats the same launches in the same order
        (one UNet evaluation): makes the context current and lets each launch warm the next one's weights."""
        return _StepScope(self)

    def __enter__(self):
        stack = _tls.__dict__.setdefault("stack", [])
        stack.append(self)
        return self

    def __exit__(self, *exc):
        _tls.stack.pop()
        # the notes on producers' split images (strict mode) hold output tensors: none may outlive the pass that made them
        # (under a graph capture they belong to the graph's private pool)
        self.__dict__.pop("recent_splits", None)
        return False


FP8_MAX_TENSORS = 512       # e4m3 activation tensors one compiled module may track (fixed: captured graphs hold the addresses)
FP8_AMAX_SLOTS = 256         # partial maxima per tensor (csrc/common.h ST_FP8_AMAX_SLOTS)
FP8_MARGIN = 2.0             # head room of a scale over the previous step's maximum (values beyond it saturate at +-448)


class Fp8Scales:
    """Delayed per-tensor scaling state of one compiled module: for every e4m3 activation tensor of the fp8 plan a scale
    (what one e4m3 unit is worth), its inverse, and the partial maxima this step's producer leaves behind.  `update()` is the
    ONE launch per step that turns the maxima into the next step's scales; `site(key)` hands out the tensor's index."""

    def __init__(self, device):
        self.scale = torch.full((FP8_MAX_TENSORS,), 1.0 / 16, dtype=torch.float32, device=device)       # until a first pass has measured
        self.inv_scale = torch.full((FP8_MAX_TENSORS,), 16.0, dtype=torch.float32, device=device)
        self.amax = torch.zeros((FP8_MAX_TENSORS, FP8_AMAX_SLOTS), dtype=torch.int32, device=device)
        self.sites = {}
        self.calibrated = False      # False until one whole forward has left its maxima (the first one runs twice)

    def site(self, key) -> int:
        i = self.sites.get(key)
        if i is None:
            if len(self.sites) >= FP8_MAX_TENSORS:
                raise BackendError("fp8 plan: more e4m3 activation tensors than FP8_MAX_TENSORS")
            i = self.sites[key] = len(self.sites)
        return i

    def reset(self) -> None:
        """Forget what earlier evaluations measured: the scales go back to their initial value and the next evaluation
        measures again (optimization.recalibrate_fp8).  Called where the activation ranges may have nothing in common with
        the last ones seen: a new trajectory (the last step of one and the first of the next sit at opposite ends of the
        sigma schedule), a new prompt, new weights.  In place: captured graphs keep the addresses."""
        self.scale.fill_(1.0 / 16)
        self.inv_scale.fill_(16.0)
        self.amax.zero_()
        self.calibrated = False

    def update(self) -> None:
        if self.sites:
            _C.check(_C.load().st_fp8_update_scales(self.scale.data_ptr(), self.inv_scale.data_ptr(), self.amax.data_ptr(),
                                                    len(self.sites), FP8_MARGIN, _C.stream_ptr()), "fp8_update_scales")


class Fp8Act:
    """An activation tensor as e4m3 bytes with ONE scale: q (M, K) uint8, the index of its scale in the context's Fp8Scales."""
    __slots__ = ("q", "index", "shape")

    def __init__(self, q: torch.Tensor, index: int, shape):
        self.q, self.index, self.shape = q, index, tuple(shape)


def fp8_scales(device) -> Fp8Scales:
    ctx = current_context(device)
    if ctx.fp8 is None:
        ctx.fp8 = Fp8Scales(device)
    return ctx.fp8


class _StepScope:
    def __init__(self, ctx: ExecContext):
        self.ctx = ctx

    def __enter__(self):
        self.ctx.__enter__()
        self.was = self.ctx.hinting
        self.ctx.hinting = True
        if self.ctx.plan is not None and not self.was:
            self.ctx.plan.begin()
        return self.ctx

    def __exit__(self, *exc):
        self.ctx.hinting = self.was
        return self.ctx.__exit__(*exc)


_tls = threading.local()
_default_ctx = {}
_default_lock = threading.Lock()


def current_context(device: Optional[torch.device] = None) -> ExecContext:
    stack = getattr(_tls, "stack", None)
    if stack:
        return stack[-1]
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    key = (dev.type, dev.index)
    with _default_lock:
        ctx = _default_ctx.get(key)
        if ctx is None:
            ctx = _default_ctx[key] = ExecContext(hints=False)
    return ctx


# What a touch is worth (round 5; whole-step A/Bs on one box, tools/ab_step.py with product:nohints / product:hintcap<n>,
# profiles/r05_hints_ab.txt): the touching launch waits for its touches before it exits - bytes at HBM speed, 26 MB of a GEGLU
# projection's weights = 6-7 us of its tail - and saves the next launch a cold start of 1-2 us: hints pay for SMALL weight matrices
# only.  bs=1 bf16: no hints +2.4 %, every matrix hinted (rounds 3-4) 0, up to 4 / 10 / 12 / 14 / 20 MB -1.0 / -1.4 / -1.3 / -0.8 / -0.9 %;
# bs=2: up to 4 MB -0.8 %, 12 MB -0.55 %, none +0.4 %; bs=4: up to 2 / 4 / 8 MB -2.6 / -2.8 / -2.1 %, none -2.4 %, 12 MB -1.8 %.
# The rule: up to 10 MB from launches of at most 1,024 rows (the 1280 level at batch 1), up to 4 MB from larger ones.
# Larger matrices get a STRIDED touch: the first 2^HINT_LEAD_SHIFT 128-byte lines of every row - the K tiles the next launch's
# prologue asks for - through bits 40-61 of the byte count (include/stabletriton_amd.h, `next_weights_bytes`); None: no touch.
HINT_SMALL_ROWS = 1024
HINT_MAX_BYTES_SMALL_ROWS = 10 << 20
HINT_MAX_BYTES = 4 << 20
HINT_MAX_BYTES_FP8 = 8 << 20
HINT_LEAD_SHIFT = 1


def _next_weights(w: torch.Tensor, rows: int = 0):
    """(pointer, bytes) of the weights the launch after this one reads, from the current context's plan - or (None, 0) where a
    touch would cost this launch (`rows` rows of output) more than it saves the next one (the rule above)."""
    ctx = current_context(w.device)
    nxt = ctx.plan.next_after(w) if (ctx.plan is not None and ctx.hinting) else None
    if nxt is None:
        return None, 0
    nbytes = nxt.numel() * nxt.element_size()
    # (e4m3 matrices: the fp8 launches are shorter and a cold start weighs more - up to 8 MB from the larger launches:
    #  fp8 mode at bs=2 18.60 ms against 18.96 with 4 MB and 18.67 with every matrix hinted, bs=4 30.72 / 30.82 / 30.64)
    big = HINT_MAX_BYTES_FP8 if nxt.element_size() == 1 else HINT_MAX_BYTES
    if nbytes > (HINT_MAX_BYTES_SMALL_ROWS if 0 < rows <= HINT_SMALL_ROWS else big):
        row_bytes = nbytes // nxt.shape[0] if nxt.dim() >= 2 else 0
        if HINT_LEAD_SHIFT is None or row_bytes % 128 or (row_bytes >> 7) < (2 << HINT_LEAD_SHIFT) or nbytes >= (1 << 40):
            return None, 0
        return nxt.data_ptr(), nbytes | ((row_bytes >> 7) << 40) | (HINT_LEAD_SHIFT << 60)
    return nxt.data_ptr(), nbytes


def _gemm_workspace(device: torch.device) -> torch.Tensor:
    return current_context(device).gemm_workspace(device)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows2d(x: torch.Tensor):
    """View x (..., K) as M rows with a uniform row stride; returns (tensor, M, ld)."""
    K = x.shape[-1]
    if x.dim() == 2 and x.stride(1) == 1:
        return x, x.shape[0], x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x, x.numel() // K, K


def _is_nhwc(x: torch.Tensor) -> bool:
    # a tensor that is contiguous in both senses (C == 1 or H*W == 1) takes the NCHW kernel
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


# ----------------------------------------------------------------------------- split fp32 operands (strict mode)
# fp32 tensors meet the matrix pipe as "split" images (csrc/split.h, ST_F32S): every value as two IEEE halves, 22 significant
# bits, three 16-bit MFMAs per product with fp32 accumulation - about five times the rate of the exact fp32 MFMA the strict
# mode used to run on, inside the same 1e-3 gates.  STRICT_SPLIT = False takes the exact-fp32 kernels instead (tests compare).
STRICT_SPLIT = True
SPLIT_K = 32                  # values per 128-byte segment of a split row


class SplitAct:
    """A matrix operand as a split image: `s` holds the (rows, K) image in a float32-typed tensor of that shape (4 bytes per
    value, NOT readable as floats), `shape` the logical shape of the tensor it was made from."""
    __slots__ = ("s", "shape")

    def __init__(self, s: torch.Tensor, shape):
        self.s, self.shape = s, tuple(shape)


def split_rows(x2: torch.Tensor, shape=None) -> SplitAct:
    """(rows, K) fp32 with unit column stride -> its split image (one launch, 4 B in / 4 B out per value)."""
    _C.require_device(x2)
    rows, K = x2.shape
    if x2.dtype != torch.float32 or K % SPLIT_K or x2.stride(1) != 1:
        raise BackendError(f"split_rows: (rows, K) float32 with K % {SPLIT_K} == 0 expected, got {tuple(x2.shape)} {x2.dtype}")
    out = torch.empty((rows, K), dtype=torch.float32, device=x2.device)
    _C.check(_C.load().st_split_f32(x2.data_ptr(), out.data_ptr(), rows, K, x2.stride(0), _C.stream_ptr()), "split_f32")
    return SplitAct(out, x2.shape if shape is None else shape)


def split_usable(dtype: torch.dtype, K: int) -> bool:
    return STRICT_SPLIT and dtype == torch.float32 and K % SPLIT_K == 0


# Split images written by PRODUCERS (st_arm_split_output): an operator whose fp32 output usually feeds a GEMM-shaped consumer
# (GroupNorm(+SiLU) -> conv / proj_in, attention -> to_out, the GEGLU projection -> ff.net.2, a GEMM that emits LayerNorm
# partials -> the LayerNorm-folded projection) leaves the image beside its output and notes it here; the consumer finds it by
# the output's memory (address, rows, row length) and skips its own st_split_f32 launch.  An entry HOLDS the output tensor,
# so its memory cannot be handed to another tensor while the note exists, and the list keeps only the last few outputs
# (consumers follow their producers within a handful of launches), and a note is only honoured while the output's version
# counter is the one it had when the image was written (an in-place update in between: the consumer splits for itself): a
# miss costs one launch, never a wrong operand.
_RECENT_SPLITS = 6
EMIT_SPLIT = True             # False: every consumer splits its own input (tests compare)


def _arm_split(out: torch.Tensor, rows: int, cols: int):
    """Arm the next launch to write the split image of `out` (rows x cols fp32, dense rows); returns the image tensor or None."""
    if not (EMIT_SPLIT and split_usable(out.dtype, cols)):
        return None
    img = torch.empty((rows, cols), dtype=torch.float32, device=out.device)
    _C.check(_C.load().st_arm_split_output(img.data_ptr(), rows, cols), "arm_split_output")
    return img


class _Armed:
    """`with _Armed(_arm_split(...)) as img:` around the launch that is to write the image: an exception raised between the arm and
    the launch (argument conversion, a rejected shape) disarms, so the arm cannot outlive the frame that owns `img` and make a
    LATER launch of the same shape write into memory the allocator has handed to somebody else."""
    __slots__ = ("img",)

    def __init__(self, img):
        self.img = img

    def __enter__(self):
        return self.img

    def __exit__(self, et, ev, tb):
        if et is not None and self.img is not None:
            _C.load().st_arm_split_output(None, 0, 0)
        return False


def _split_notes(device, create: bool = False):
    """The notes of the current context; the shared default context (no plan, never left, used by any thread) keeps none of its
    own: eager strict-mode calls outside a compiled module note per THREAD (bounded: the last few outputs)."""
    ctx = current_context(device)
    holder = ctx.__dict__ if ctx.plan is not None else _tls.__dict__
    if create:
        return holder.setdefault("recent_splits", [])
    return holder.get("recent_splits", ())


def _note_split(out: torch.Tensor, img: Optional[torch.Tensor], rows: int, cols: int) -> None:
    if img is None:
        return
    if current_context(out.device).plan is None and torch.cuda.is_current_stream_capturing():
        return          # no context of its own: the notes must not keep tensors of somebody's graph pool alive
    lst = _split_notes(out.device, create=True)
    lst.append((out, out.data_ptr(), rows, cols, img, out._version))      # (the version: an in-place update of the output later makes the image stale)
    del lst[:-_RECENT_SPLITS]


def _image_columns(t: torch.Tensor, rows: int, ld: int):
    """(image, first column) when `t` - rows x C values at row stride ld - is a column range of a recently noted output
    whose rows are ld values long (the K / V slices of the fused q|k|v projection), else None."""
    if not EMIT_SPLIT:
        return None
    for ent in reversed(_split_notes(t.device)):
        off = t.data_ptr() - ent[1]
        if ent[2] == rows and ent[3] == ld and 0 <= off < 4 * ld and off % 128 == 0 and t._version == ent[5]:
            return ent[4], off // 4
    return None


def _split_of(x: torch.Tensor, rows: int, cols: int, ld: int) -> torch.Tensor:
    """The split image of the (rows, cols) fp32 matrix at x's address (row stride ld): a producer's, if one was noted, else made now."""
    if ld == cols:
        for ent in reversed(_split_notes(x.device)):
            if ent[1] == x.data_ptr() and ent[2] == rows and ent[3] == cols and x._version == ent[5]:
                return ent[4]
    x2 = x if (x.dim() == 2 and x.shape[1] == cols) else x.as_strided((rows, cols), (ld, 1))
    return split_rows(x2).s


@torch.no_grad()
def _split_weight(owner: torch.Tensor, as_rows=None, want_rowsum: bool = False):
    """Split image of a weight, kept by the current execution context and re-derived in place when `owner` changes.
    `as_rows(owner)` gives the (N, K) row-major view to split (default: the tensor itself); with want_rowsum also
    c[n] = sum_k of the values the image holds (what a folded LayerNorm subtracts)."""

    def compute():
        w2 = owner.detach() if as_rows is None else as_rows(owner.detach())
        if w2.dim() != 2 or not w2.is_contiguous():
            raise BackendError("split weight: a row-major (N, K) view is needed")
        img = split_rows(w2).s
        if not want_rowsum:
            return (img,)
        h = img.view(torch.float16).view(w2.shape[0], w2.shape[1] // SPLIT_K, 2, SPLIT_K).double()
        c = (h[:, :, 0, :].sum(dim=(1, 2)) + h[:, :, 1, :].sum(dim=(1, 2)) / 2048.0).float().contiguous()
        return (img, c)

    ctx = current_context(owner.device)
    if ctx.plan is None:          # the shared default context keeps nothing alive: a launch outside a compiled module splits on the spot
        return compute()
    return ctx.derived_weights(("split", id(owner), want_rowsum), [owner], compute).value


def _conv_weight_rows(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, R, S) channels_last = memory [Cout][R][S][Cin] -> the (Cout * R * S, Cin) view of it"""
    return w.permute(0, 2, 3, 1).reshape(-1, w.shape[1])


# ----------------------------------------------------------------------------- norms
def group_norm(x: torch.Tensor, num_groups: int, weight: torch.Tensor, bias: torch.Tensor,
               eps: float, silu: bool) -> torch.Tensor:
    """(N,C,*) GroupNorm(+SiLU); NCHW-contiguous or channels_last in, same layout out."""
    _C.require_device(x, weight, bias)
    lib = _C.load()
    if x.dim() < 3:
        raise BackendError(f"group_norm expects (N, C, *) input, got {tuple(x.shape)}")
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    if _is_nhwc(x):
        layout = _C.ST_NHWC
    else:
        layout = _C.ST_NCHW
        if not x.is_contiguous():
            x = x.contiguous()
    y = torch.empty_like(x)                     # preserve_format keeps NHWC strides
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if bias.dtype == x.dtype else bias.to(x.dtype)
    # scratch for the partial statistics: allocated per call from torch's caching allocator (stream-ordered; under
    # graph capture it belongs to the graph's private pool, so replays never alias a buffer somebody else owns)
    ws = torch.empty(lib.st_group_norm_workspace_bytes(N, Cc, HW, num_groups), dtype=torch.uint8, device=x.device)
    with _Armed(_arm_split(y, N * HW, Cc) if layout == _C.ST_NHWC else None) as img:      # strict mode: the conv / proj_in behind it reads the split image
        _C.check(lib.st_group_norm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, Cc, HW, num_groups,
                                   float(eps), int(bool(silu)), layout, _C.dtype_code(x.dtype), ws.data_ptr(),
                                   _C.stream_ptr()), "group_norm")
    _note_split(y, img, N * HW, Cc)
    return y


class ColStats:
    """GroupNorm partials a GEMM / conv left beside its output: (tiles, C) float2 = per tile row of the launch and per
    channel (sum, sum of squares); `rows` = rows per tile row."""
    __slots__ = ("buf", "rows", "channels")

    def __init__(self, buf: torch.Tensor, rows: int, channels: int):
        self.buf, self.rows, self.channels = buf, rows, channels


COLSTATS_MIN_ROWS = 64         # smallest tile height of any GEMM configuration: sizes the partial buffers


def _colstats_buffer(M: int, N: int, device):
    import ctypes
    tiles = (M + COLSTATS_MIN_ROWS - 1) // COLSTATS_MIN_ROWS
    return torch.empty((tiles, N, 2), dtype=torch.float32, device=device), tiles, ctypes.c_int(0)


def group_norm_from_stats(x: torch.Tensor, sources, num_groups: int, weight: torch.Tensor, bias: torch.Tensor,
                          eps: float, silu: bool) -> torch.Tensor:
    """GroupNorm(+SiLU) of a channels_last x whose statistics its producer(s) emitted (`sources`: one ColStats, or two for
    a channel concatenation; None entries = that producer could not emit): no statistics pass over x.  Falls back to
    `group_norm` when a source is missing."""
    if any(s is None for s in sources) or not _is_nhwc(x) or not 1 <= len(sources) <= 2:
        return group_norm(x, num_groups, weight, bias, eps, silu)
    _C.require_device(x, weight, bias)
    lib = _C.load()
    N, Cc = x.shape[0], x.shape[1]
    HW = x.numel() // (N * Cc)
    if sum(s.channels for s in sources) != Cc or any(HW % s.rows for s in sources):
        return group_norm(x, num_groups, weight, bias, eps, silu)
    y = torch.empty_like(x)
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if bias.dtype == x.dtype else bias.to(x.dtype)
    ws = torch.empty(lib.st_group_norm_workspace_bytes(N, Cc, HW, num_groups), dtype=torch.uint8, device=x.device)
    s0 = sources[0]
    s1 = sources[1] if len(sources) == 2 else None
    with _Armed(_arm_split(y, N * HW, Cc)) as img:
        _C.check(lib.st_group_norm_from_stats(x.data_ptr(), w.data_ptr(),
                        b.data_ptr(), y.data_ptr(), N, Cc, HW, num_groups, float(eps), int(bool(silu)), _C.dtype_code(x.dtype),
                        s0.buf.data_ptr(), s0.channels, s0.rows, None if s1 is None else s1.buf.data_ptr(),
                        0 if s1 is None else s1.channels, 0 if s1 is None else s1.rows, ws.data_ptr(), _C.stream_ptr()),
                 "group_norm_from_stats")
    _note_split(y, img, N * HW, Cc)
    return y


def group_norm_from_stats_cat(x0: torch.Tensor, x1: torch.Tensor, sources, num_groups: int, weight: torch.Tensor, bias: torch.Tensor,
                              eps: float, silu: bool) -> torch.Tensor:
    """GroupNorm(+SiLU) of torch.cat([x0, x1], 1) (channels_last) without writing the concatenated tensor: both halves bring
    the statistics of their producers (`sources`: two ColStats).  Anything the two-source kernel does not take (a missing
    source, channel counts that are not whole 16-byte vectors, NCHW inputs) goes through torch.cat and `group_norm_from_stats`.
    Bit-identical to that path."""
    vec = 4 if x0.dtype == torch.float32 else 8
    ok = (len(sources) == 2 and all(s is not None for s in sources) and _is_nhwc(x0) and _is_nhwc(x1) and x0.dtype == x1.dtype
          and x0.shape[0] == x1.shape[0] and x0.shape[2:] == x1.shape[2:] and x0.shape[1] % vec == 0 and x1.shape[1] % vec == 0)
    if ok:
        N, C0, C1 = x0.shape[0], x0.shape[1], x1.shape[1]
        HW = x0.numel() // (N * C0)
        ok = sources[0].channels == C0 and sources[1].channels == C1 and not any(HW % s.rows for s in sources) and (C0 + C1) % num_groups == 0
    if not ok:
        return group_norm_from_stats(torch.cat([x0, x1], dim=1), sources, num_groups, weight, bias, eps, silu)
    _C.require_device(x0, x1, weight, bias)
    lib = _C.load()
    Cc = C0 + C1
    y = torch.empty((N, Cc) + tuple(x0.shape[2:]), dtype=x0.dtype, device=x0.device, memory_format=torch.channels_last)
    w = weight if weight.dtype == x0.dtype else weight.to(x0.dtype)
    b = bias if bias.dtype == x0.dtype else bias.to(x0.dtype)
    ws = torch.empty(lib.st_group_norm_workspace_bytes(N, Cc, HW, num_groups), dtype=torch.uint8, device=x0.device)
    s0, s1 = sources
    with _Armed(_arm_split(y, N * HW, Cc)) as img:
        _C.check(lib.st_group_norm_from_stats_cat(x0.data_ptr(), x1.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, Cc, HW, num_groups,
                                                  float(eps), int(bool(silu)), _C.dtype_code(x0.dtype), s0.buf.data_ptr(), s0.channels, s0.rows,
                                                  s1.buf.data_ptr(), s1.channels, s1.rows, ws.data_ptr(), _C.stream_ptr()),
                 "group_norm_from_stats_cat")
    _note_split(y, img, N * HW, Cc)
    return y


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    _C.require_device(x, weight, bias)
    lib = _C.load()
    Cc = x.shape[-1]
    if weight.numel() != Cc:
        raise BackendError("layer_norm: only normalisation over the last dimension is supported")
    xc = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(xc)
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if bias.dtype == x.dtype else bias.to(x.dtype)
    _C.check(lib.st_layer_norm(xc.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), xc.numel() // Cc, Cc,
                               float(eps), _C.dtype_code(x.dtype), _C.stream_ptr()), "layer_norm")
    return y


def geglu(state: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """state * gelu_erf(gate); accepts the two strided halves of one projection."""
    _C.require_device(state, gate)
    lib = _C.load()
    if state.shape != gate.shape or state.dtype != gate.dtype:
        raise BackendError("geglu: state and gate must have the same shape and dtype")
    F = state.shape[-1]

    def rows(t):
        if t.stride(-1) == 1 and (t.dim() == 1 or all(
                t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2))):
            return t, (t.stride(-2) if t.dim() >= 2 else F)
        t = t.contiguous()
        return t, F

    s, lds = rows(state)
    g, ldg = rows(gate)
    out = torch.empty(state.shape, dtype=state.dtype, device=state.device)
    _C.check(lib.st_geglu(s.data_ptr(), g.data_ptr(), out.data_ptr(), state.numel() // F, F, lds, ldg, F,
                          _C.dtype_code(state.dtype), _C.stream_ptr()), "geglu")
    return out


# ----------------------------------------------------------------------------- linear
STATS_MAX_CHUNKS = 64      # N tiles a row-statistics buffer can hold (N <= 64 * 64)


class RowStats:
    """LayerNorm partials of a GEMM output: (M, chunks) float2 = per-row (sum, sum of squares) per N tile."""
    __slots__ = ("buf", "chunks")

    def __init__(self, buf: torch.Tensor, chunks: int):
        self.buf, self.chunks = buf, chunks


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, silu: bool = False,
           geglu: bool = False, residual: Optional[torch.Tensor] = None, emit_stats: bool = False, emit_colstats: bool = False,
           emit_q8=None):
    """epilogue(x @ weight.T): +bias, then SiLU or GEGLU (weight has 2N rows), then +residual.
    With emit_stats the GEMM also writes the LayerNorm partials of its output rows and the call
    returns (out, RowStats).  With emit_colstats (x is (B, T, K): T tokens per image) it writes the GroupNorm partials
    of its output columns and the call returns (out, ColStats or None).  With emit_q8 (a site key of the fp8 plan) the
    epilogue also leaves an e4m3 copy of the output under that tensor's delayed scale; the Fp8Act is appended to the result."""
    _C.require_device(x, weight, bias, residual)
    lib = _C.load()
    K = x.shape[-1]
    if weight.dim() != 2 or weight.shape[1] != K:
        raise BackendError(f"linear: weight {tuple(weight.shape)} does not match input K={K}")
    if weight.dtype != x.dtype:
        raise BackendError(f"linear: weight dtype {weight.dtype} != input dtype {x.dtype} "
                           "(convert the module once, the wrappers never mutate parameters)")
    w = weight if weight.is_contiguous() else weight.contiguous()
    x2, M, lda = _rows2d(x)
    N = w.shape[0] // 2 if geglu else w.shape[0]
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    code = _C.dtype_code(x.dtype)
    if split_usable(x.dtype, K) and emit_q8 is None and weight.is_contiguous():
        # strict mode: both matrix operands as split images (the weight's is kept by the context), everything else fp32
        x2, lda, w, code = _split_of(x2, M, K, lda), K, _split_weight(weight)[0], _C.ST_F32S
    epi = 0
    if bias is not None:
        epi |= _C.EPI_BIAS
        if bias.dtype != x.dtype:
            bias = bias.to(x.dtype)
    if silu:
        epi |= _C.EPI_SILU
    if geglu:
        epi |= _C.EPI_GEGLU
    ldr = 0
    if residual is not None:
        if residual.shape != out.shape or residual.dtype != x.dtype:
            raise BackendError("linear: residual must match the output shape and dtype")
        residual, _, ldr = _rows2d(residual)
        epi |= _C.EPI_RESIDUAL
    gws = _gemm_workspace(x.device)
    stats = chunks = None
    if emit_stats:
        import ctypes
        cap = min(STATS_MAX_CHUNKS, (N + 63) // 64)
        stats = torch.empty((M, cap, 2), dtype=torch.float32, device=x.device)
        chunks = ctypes.c_int(0)
    nxt_p, nxt_b = _next_weights(w, M)
    cbuf = ctiles = crows = None
    rows_per_image = 0
    if emit_colstats:
        import ctypes
        if x.dim() == 3:
            rows_per_image = x.shape[1]
            cbuf, ctiles, crows = _colstats_buffer(M, N, x.device)
        # (any other rank: tokens were flattened, the image boundaries are unknown - no partials, the consumer
        # GroupNorm takes its own statistics pass, as for every other producer that cannot emit them)
    act8 = None
    # strict mode: an output that a LayerNorm-folded projection (emit_stats) or the feed-forward output projection (geglu) reads
    # next leaves its split image too
    with _Armed(_arm_split(out, M, N) if (code == _C.ST_F32S and (emit_stats or geglu)) else None) as img:
        if emit_q8 is None:
            _C.check(lib.st_linear(x2.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), None, out.data_ptr(), M, N, K,
                            lda, N, ldr, rows_per_image, epi, code, gws.data_ptr(), gws.numel(),
                            _ptr(stats), 0 if stats is None else stats.shape[1],
                            None if chunks is None else ctypes.byref(chunks), _ptr(cbuf), ctiles or 0,
                            None if crows is None else ctypes.byref(crows), nxt_p, nxt_b, _C.stream_ptr()), "linear")
            _note_split(out, img, M, N)
        else:
            sc = fp8_scales(x.device)
            idx = sc.site(emit_q8)
            q8 = torch.empty((M, N), dtype=torch.uint8, device=x.device)
            act8 = Fp8Act(q8, idx, out.shape)
            _C.check(lib.st_linear_emit8(x2.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), None, out.data_ptr(), M, N, K,
                            lda, N, ldr, rows_per_image, epi, _C.dtype_code(x.dtype), gws.data_ptr(), gws.numel(),
                            _ptr(stats), 0 if stats is None else stats.shape[1],
                            None if chunks is None else ctypes.byref(chunks), _ptr(cbuf), ctiles or 0,
                            None if crows is None else ctypes.byref(crows), q8.data_ptr(), N, sc.inv_scale[idx:].data_ptr(), sc.amax[idx:].data_ptr(),
                            nxt_p, nxt_b, _C.stream_ptr()), "linear_emit8")
    if act8 is not None:
        if emit_colstats:
            raise BackendError("linear: emit_q8 with emit_colstats is not supported")
        if emit_stats:
            if chunks.value <= 0:
                raise BackendError("linear: this shape cannot emit LayerNorm row statistics (K must be a multiple of the K tile)")
            dense = stats.view(-1)[:M * chunks.value * 2].view(M, chunks.value, 2)
            return out, RowStats(dense, chunks.value), act8
        return out, act8
    if emit_colstats:
        return out, (ColStats(cbuf, crows.value, N) if crows is not None and crows.value > 0 else None)
    if emit_stats:
        if chunks.value <= 0:
            raise BackendError("linear: this shape cannot emit LayerNorm row statistics (K must be a multiple of the K tile)")
        dense = stats.view(-1)[:M * chunks.value * 2].view(M, chunks.value, 2)      # the kernel packs rows at `chunks` stride
        return out, RowStats(dense, chunks.value)
    return out


def ln_linear(x: torch.Tensor, stats: "RowStats", w_folded: torch.Tensor, c: torch.Tensor, d: torch.Tensor, eps: float, *,
              geglu: bool = False, emit_split: bool = False) -> torch.Tensor:
    """LayerNorm(x) @ W.T (+bias) as one GEMM on gamma-folded weights with a rank-1 correction
    (see st_ln_linear in the header); `w_folded`, `c`, `d` come from `fold_layer_norm`, `stats`
    from the `linear(..., emit_stats=True)` call that produced x."""
    _C.require_device(x, w_folded, c, d, stats.buf)
    lib = _C.load()
    K = x.shape[-1]
    if w_folded.shape[1] != K or w_folded.dtype != x.dtype or c.dtype != torch.float32 or d.dtype != torch.float32:
        raise BackendError("ln_linear: folded operands do not match the input")
    x2, M, lda = _rows2d(x)
    N = w_folded.shape[0] // 2 if geglu else w_folded.shape[0]
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    code = _C.dtype_code(x.dtype)
    if split_usable(x.dtype, K) and w_folded.is_contiguous():
        # (c must be the row sums of the values the split image holds: the fold subtracts mean * c from their products)
        w_folded, c = _split_weight(w_folded, want_rowsum=True)
        x2, lda, code = _split_of(x2, M, K, lda), K, _C.ST_F32S
    nxt_p, nxt_b = _next_weights(w_folded, M)
    # strict mode: the GEGLU projection's output (read by ff.net.2) and, on request, the q|k|v projection's (its K and V columns
    # are attention operands) leave their split images
    with _Armed(_arm_split(out, M, N) if (code == _C.ST_F32S and (geglu or emit_split)) else None) as img:
        _C.check(lib.st_ln_linear(x2.data_ptr(), stats.buf.data_ptr(), stats.chunks, w_folded.data_ptr(),
                        c.data_ptr(), d.data_ptr(), out.data_ptr(), M, N, K,
                        lda, N, float(eps), _C.EPI_GEGLU if geglu else 0, code, nxt_p, nxt_b, _C.stream_ptr()), "ln_linear")
    _note_split(out, img, M, N)
    return out


def xattn_fusable(x: torch.Tensor, k: torch.Tensor, heads: int) -> bool:
    """Shapes `ln_linear_xattn` takes: bf16 / fp16, a short context (the 16-row attention kernel's range) and query blocks of
    128 rows that do not straddle batch entries."""
    rows = x.shape[-2] if x.dim() >= 3 else x.shape[0]
    return (x.dtype in (torch.bfloat16, torch.float16) and k.dim() == 3 and k.shape[1] < 256 and rows % 128 == 0 and k.shape[-1] == heads * 64
            and k.stride(-1) == 1)


def xattn_fusion_pays(x: torch.Tensor, heads: int) -> bool:
    """The fused launch projects the queries in 128 x 64 tiles (one head per tile).  That is the projection's own best tile
    while the launch is one round of tiles (SDXL's 1024-token level at batch 1: 160); beyond that the wider tiles the
    dispatch would pick win back more than the saved launch (measured: 24.9 against 20.4 us at the 4096-token level,
    -1 % per step at batch 2 and 4), so the two-launch route is taken."""
    return (x.numel() // x.shape[-1] // 128) * heads <= 256


def ln_linear_xattn(x: torch.Tensor, stats: "RowStats", w_folded: torch.Tensor, c: torch.Tensor, d: torch.Tensor, eps: float,
                    k: torch.Tensor, v: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """attention(LayerNorm(x) @ Wq.T (+bias), k, v) over a short context as ONE launch (st_ln_linear_xattn): the query
    projection's epilogue runs the attention core on the tile it has just computed.  Bit-identical to
    `attention(ln_linear(x, ...), k, v, heads, scale)`."""
    _C.require_device(x, w_folded, c, d, stats.buf, k, v)
    lib = _C.load()
    K = x.shape[-1]
    N = w_folded.shape[0]
    if w_folded.shape[1] != K or w_folded.dtype != x.dtype or c.dtype != torch.float32 or d.dtype != torch.float32:
        raise BackendError("ln_linear_xattn: folded operands do not match the input")
    if x.dim() != 3 or k.dim() != 3 or v.shape != k.shape or k.shape[0] != x.shape[0] or N != heads * 64 or k.shape[-1] != N:
        raise BackendError(f"ln_linear_xattn: shapes x={tuple(x.shape)} k={tuple(k.shape)} v={tuple(v.shape)} heads={heads}")
    if not xattn_fusable(x, k, heads) or v.stride(-1) != 1 or k.stride(0) != k.shape[1] * k.stride(1) or v.stride(0) != v.shape[1] * v.stride(1):
        raise BackendError("ln_linear_xattn: layout not supported (bf16 / fp16, context < 256 tokens, 128 | rows per batch, dense batches)")
    x2, M, lda = _rows2d(x)
    out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    nxt_p, nxt_b = _next_weights(w_folded, M)
    S = k.shape[1]
    _C.check(lib.st_ln_linear_xattn(x2.data_ptr(), stats.buf.data_ptr(), stats.chunks, w_folded.data_ptr(),
                    c.data_ptr(), d.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), M, N, K, lda, N, float(eps),
                    x.shape[1], S, heads, k.stride(1), v.stride(1), float(scale), _C.dtype_code(x.dtype), nxt_p, nxt_b, _C.stream_ptr()), "ln_linear_xattn")
    return out


@torch.no_grad()
def fold_layer_norm(gamma: torch.Tensor, beta: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """(W * diag(gamma) in the weight dtype, c = row sums of that rounded matrix, d = W beta + bias), fp32 c/d."""
    wf = (weight.float() * gamma.float()[None, :]).to(weight.dtype).contiguous()
    c = wf.float().sum(dim=1).contiguous()
    d = weight.float() @ beta.float()
    if bias is not None:
        d = d + bias.float()
    return wf, c, d.contiguous()


# ----------------------------------------------------------------------------- fp8 projections
FP8_MAX = 448.0           # largest finite OCP e4m3 value


class Fp8Rows:
    """An activation matrix quantised row by row: e4m3 bytes (M, K) + fp32 scale per row."""
    __slots__ = ("q", "scale", "shape")

    def __init__(self, q: torch.Tensor, scale: torch.Tensor, shape):
        self.q, self.scale, self.shape = q, scale, tuple(shape)


@torch.no_grad()
def quantize_weight_fp8(weight: torch.Tensor):
    """(N, K) weight -> (e4m3 bytes as uint8 (N, K), fp32 scale per output channel).  Host-side, once per weight."""
    w = weight.detach().float()
    scale = (w.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX).contiguous()
    q = (w / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    return q, scale


def quantize_fp8(x: torch.Tensor, layernorm=None) -> Fp8Rows:
    """Row-wise e4m3 quantisation of x (..., K); with `layernorm=(gamma, beta, eps)` of LayerNorm(x), in one pass."""
    _C.require_device(x)
    lib = _C.load()
    K = x.shape[-1]
    x2, M, ldx = _rows2d(x)
    q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    scale = torch.empty((M,), dtype=torch.float32, device=x.device)
    if layernorm is None:
        _C.check(lib.st_quantize_fp8(x2.data_ptr(), ldx,
                        q.data_ptr(), scale.data_ptr(), M, K, _C.dtype_code(x.dtype), _C.stream_ptr()), "quantize_fp8")
    else:
        gamma, beta, eps = layernorm
        if ldx != K:
            x2 = x2.contiguous()
        g = gamma if gamma.dtype == x.dtype else gamma.to(x.dtype)
        b = beta if beta.dtype == x.dtype else beta.to(x.dtype)
        _C.check(lib.st_layer_norm_quantize_fp8(x2.data_ptr(),
                        g.data_ptr(), b.data_ptr(), q.data_ptr(), scale.data_ptr(), M, K, float(eps), _C.dtype_code(x.dtype),
                        _C.stream_ptr()), "layer_norm_quantize_fp8")
    return Fp8Rows(q, scale, x.shape)


def linear_fp8(x: "Fp8Rows", wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None, *, silu: bool = False,
               geglu: bool = False, residual: Optional[torch.Tensor] = None, out_dtype=torch.bfloat16) -> torch.Tensor:
    """epilogue((xq @ wq.T) * row_scale * w_scale): both operands e4m3 on the fp8 matrix pipe, fp32 accumulation, bf16 out."""
    _C.require_device(x.q, wq, w_scale, bias, residual)
    lib = _C.load()
    M, K = x.q.shape
    if wq.dtype != torch.uint8 or wq.dim() != 2 or wq.shape[1] != K or w_scale.numel() != wq.shape[0]:
        raise BackendError("linear_fp8: weight must be (N, K) e4m3 bytes (uint8) with one fp32 scale per row")
    if out_dtype != torch.bfloat16:
        raise BackendError("linear_fp8: the output type is bfloat16")
    N = wq.shape[0] // 2 if geglu else wq.shape[0]
    out = torch.empty(*x.shape[:-1], N, dtype=out_dtype, device=wq.device)
    epi = 0
    if bias is not None:
        epi |= _C.EPI_BIAS
        bias = bias if bias.dtype == out_dtype else bias.to(out_dtype)
    if silu:
        epi |= _C.EPI_SILU
    if geglu:
        epi |= _C.EPI_GEGLU
    ldr = 0
    if residual is not None:
        if residual.shape != out.shape or residual.dtype != out_dtype:
            raise BackendError("linear_fp8: residual must match the output shape and dtype")
        residual, _, ldr = _rows2d(residual)
        epi |= _C.EPI_RESIDUAL
    gws = _gemm_workspace(wq.device)
    nxt_p, nxt_b = _next_weights(wq, x.q.shape[0])
    _C.check(lib.st_linear_fp8(x.q.data_ptr(), x.scale.data_ptr(), wq.data_ptr(), w_scale.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                    M, N, K, K, N, ldr, epi, gws.data_ptr(), gws.numel(), nxt_p, nxt_b, _C.stream_ptr()), "linear_fp8")
    return out


def linear_fp8x(x: "Fp8Act", wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None, *, geglu: bool = False,
                residual: Optional[torch.Tensor] = None, ln=None, emit_stats: bool = False, emit_q8=None, want_out: bool = True):
    """The fp8 GEMM of the fp8 plan: epilogue((xq @ wq.T) * scale(x) * w_scale) with x an Fp8Act (ONE delayed scale for the whole
    tensor).  `ln = (RowStats, c, d, eps)` folds the LayerNorm in front of the projection (x is then the e4m3 copy of the
    un-normalised tensor).  Returns out [, RowStats] [, Fp8Act of the output]; with want_out=False only the Fp8Act."""
    _C.require_device(x.q, wq, w_scale, bias, residual)
    lib = _C.load()
    import ctypes
    M, K = x.q.shape
    if wq.dtype != torch.uint8 or wq.dim() != 2 or wq.shape[1] != K or w_scale.numel() != wq.shape[0]:
        raise BackendError("linear_fp8x: weight must be (N, K) e4m3 bytes (uint8) with one fp32 scale per row")
    N = wq.shape[0] // 2 if geglu else wq.shape[0]
    dev = wq.device
    sc = fp8_scales(dev)
    out = torch.empty(*x.shape[:-1], N, dtype=torch.bfloat16, device=dev) if want_out else None
    if not want_out and emit_q8 is None:
        raise BackendError("linear_fp8x: nothing to compute (no output, no e4m3 copy)")
    epi = 0
    if bias is not None:
        epi |= _C.EPI_BIAS
        bias = bias if bias.dtype == torch.bfloat16 else bias.to(torch.bfloat16)
    if geglu:
        epi |= _C.EPI_GEGLU
    ldr = 0
    if residual is not None:
        if not want_out or residual.shape != out.shape or residual.dtype != torch.bfloat16:
            raise BackendError("linear_fp8x: residual must match the bf16 output")
        residual, _, ldr = _rows2d(residual)
        epi |= _C.EPI_RESIDUAL
    st_in = c = d = None
    eps = 0.0
    if ln is not None:
        st_in, c, d, eps = ln
        if bias is not None:
            raise BackendError("linear_fp8x: with a folded LayerNorm the bias lives in d")
    stats = chunks = None
    if emit_stats:
        cap = min(STATS_MAX_CHUNKS, (N + 63) // 64)
        stats = torch.empty((M, cap, 2), dtype=torch.float32, device=dev)
        chunks = ctypes.c_int(0)
    q8 = act8 = None
    idx_out = 0
    if emit_q8 is not None:
        idx_out = sc.site(emit_q8)
        q8 = torch.empty((M, N), dtype=torch.uint8, device=dev)
        act8 = Fp8Act(q8, idx_out, tuple(x.shape[:-1]) + (N,))
    gws = _gemm_workspace(dev)
    nxt_p, nxt_b = _next_weights(wq, M)
    _C.check(lib.st_linear_fp8x(x.q.data_ptr(), sc.scale[x.index:].data_ptr(), 0, wq.data_ptr(), w_scale.data_ptr(), _ptr(bias), _ptr(residual),
                                _ptr(out), M, N, K, K, N, ldr, epi,
                                None if st_in is None else st_in.buf.data_ptr(), 0 if st_in is None else st_in.chunks, _ptr(c), _ptr(d), float(eps),
                                _ptr(stats), 0 if stats is None else stats.shape[1], None if chunks is None else ctypes.byref(chunks),
                                _ptr(q8), N, None if q8 is None else sc.inv_scale[idx_out:].data_ptr(), None if q8 is None else sc.amax[idx_out:].data_ptr(),
                                gws.data_ptr(), gws.numel(), nxt_p, nxt_b, _C.stream_ptr()), "linear_fp8x")
    res = []
    if want_out:
        res.append(out)
    if emit_stats:
        if chunks.value <= 0:
            raise BackendError("linear_fp8x: this shape cannot emit LayerNorm row statistics")
        res.append(RowStats(stats.view(-1)[:M * chunks.value * 2].view(M, chunks.value, 2), chunks.value))
    if act8 is not None:
        res.append(act8)
    return res[0] if len(res) == 1 else tuple(res)


@torch.no_grad()
def fold_layer_norm_fp8(gamma: torch.Tensor, beta: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """fold_layer_norm for the fp8 GEMM: (e4m3 bytes of W * diag(gamma), per-channel scales, c = row sums of the DEQUANTISED
    folded weights, d = W beta + bias), c / d fp32."""
    wf = weight.float() * gamma.float()[None, :]
    wq, ws = quantize_weight_fp8(wf)
    c = (wq.view(torch.float8_e4m3fn).float().sum(dim=1) * ws).contiguous()
    d = weight.float() @ beta.float()
    if bias is not None:
        d = d + bias.float()
    return wq, ws, c, d.contiguous()


# ----------------------------------------------------------------------------- attention
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, scale: float) -> torch.Tensor:
    """q (B,T,H*D), k/v (B,S,H*D) in projection layout -> (B,T,H*D)."""
    _C.require_device(q, k, v)
    lib = _C.load()
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
        raise BackendError("attention expects (B, T, H*D) tensors")
    B, T, Cc = q.shape
    S = k.shape[1]
    D = Cc // num_heads

    def tok(t):
        if t.stride(2) == 1 and t.stride(0) == t.stride(1) * t.shape[1]:
            return t, t.stride(1)
        t = t.contiguous()
        return t, t.shape[2]

    q_, ldq = tok(q)
    k_, ldk = tok(k)
    v_, ldv = tok(v)
    out = torch.empty((B, T, Cc), dtype=q.dtype, device=q.device)
    ki = _image_columns(k_, B * S, ldk) if q.dtype == torch.float32 else None      # strict mode: K / V columns of a producer's split image
    vi = _image_columns(v_, B * S, ldv) if ki is not None else None
    with _Armed(_arm_split(out, B * T, Cc)) as img:              # strict mode: the output projection reads the split image
        if ki is not None and vi is not None and D == 64:
            _C.check(lib.st_attention_split(q_.data_ptr(), ki[0].data_ptr() + 4 * ki[1], vi[0].data_ptr() + 4 * vi[1], out.data_ptr(), B, T, S, num_heads, D,
                                            ldq, ki[0].shape[1], vi[0].shape[1], Cc, float(scale), _C.stream_ptr()), "attention_split")
        else:
            _C.check(lib.st_attention(q_.data_ptr(), k_.data_ptr(), v_.data_ptr(), out.data_ptr(), B, T, S, num_heads, D,
                                      ldq, ldk, ldv, Cc, float(scale), _C.dtype_code(q.dtype), _C.stream_ptr()), "attention")
    _note_split(out, img, B * T, Cc)
    return out


# ----------------------------------------------------------------------------- conv
def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, padding: int, *,
           upsample2x: bool = False, rowbias: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, emit_colstats: bool = False):
    """NHWC implicit-GEMM conv.  x: (N,C,H,W) logical; returns a channels_last tensor.
    rowbias (N,Cout) is added per image (time-embedding projection); residual is
    (N,Cout,Hout,Wout) channels_last.  With emit_colstats the conv also writes the GroupNorm partials of its output
    channels and the call returns (out, ColStats or None)."""
    _C.require_device(x, weight, bias, rowbias, residual)
    lib = _C.load()
    if x.dim() != 4 or weight.dim() != 4:
        raise BackendError("conv2d expects 4-D input and weight")
    if weight.dtype != x.dtype:
        raise BackendError(f"conv2d: weight dtype {weight.dtype} != input dtype {x.dtype}")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    w = weight if weight.is_contiguous(memory_format=torch.channels_last) else weight.contiguous(
        memory_format=torch.channels_last)
    N, Cin, H, W = x.shape
    Cout, _, R, S = w.shape
    He, We = (2 * H, 2 * W) if upsample2x else (H, W)
    Ho = (He + 2 * padding - R) // stride + 1
    Wo = (We + 2 * padding - S) // stride + 1
    out = torch.empty((N, Cout, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    epi = 0
    if bias is not None:
        epi |= _C.EPI_BIAS
        if bias.dtype != x.dtype:
            bias = bias.to(x.dtype)
    if rowbias is not None:
        if rowbias.shape != (N, Cout) or rowbias.dtype != x.dtype:
            raise BackendError("conv2d: rowbias must be (N, Cout) of the input dtype")
        rowbias = rowbias.contiguous()
        epi |= _C.EPI_ROWBIAS
    if residual is not None:
        if residual.shape != out.shape or residual.dtype != x.dtype:
            raise BackendError("conv2d: residual must match the output")
        if not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        epi |= _C.EPI_RESIDUAL
    gws = _gemm_workspace(x.device)
    code = _C.dtype_code(x.dtype)
    if split_usable(x.dtype, Cin) and w is weight:
        # strict mode: the pixels' channel vectors and the filter taps as split images (32 channels per segment)
        x = _split_of(x, N * H * W, Cin, Cin)
        w, code = _split_weight(weight, _conv_weight_rows)[0], _C.ST_F32S
    nxt_p, nxt_b = _next_weights(w, N * H * W)
    cbuf = ctiles = crows = None
    if emit_colstats:
        import ctypes
        cbuf, ctiles, crows = _colstats_buffer(N * Ho * Wo, Cout, x.device)
    _C.check(lib.st_conv2d(x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), _ptr(rowbias), out.data_ptr(),
                           N, H, W, Cin, Cout, R, S, stride, padding, int(upsample2x), epi,
                           code, gws.data_ptr(), gws.numel(), _ptr(cbuf), ctiles or 0,
                           None if crows is None else ctypes.byref(crows), nxt_p, nxt_b, _C.stream_ptr()), "conv2d")
    if emit_colstats:
        return out, (ColStats(cbuf, crows.value, Cout) if crows.value > 0 else None)
    return out


def conv2d_cat(x0: torch.Tensor, x1: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], *,
               residual: Optional[torch.Tensor] = None, emit_colstats: bool = False):
    """1x1 convolution (stride 1, no padding) of torch.cat([x0, x1], 1) without writing the concatenated tensor (the resnet
    shortcut behind a skip connection).  Inputs the two-source kernel does not take go through torch.cat and `conv2d`.
    Bit-identical to that path (same K order, same tiles)."""
    kb = 32 if x0.dtype == torch.float32 else 64
    ok = (x0.dim() == 4 and x1.dim() == 4 and weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1) and x0.dtype == x1.dtype == weight.dtype
          and x0.shape[0] == x1.shape[0] and x0.shape[2:] == x1.shape[2:] and x0.shape[1] % kb == 0 and x1.shape[1] % kb == 0
          and weight.shape[1] == x0.shape[1] + x1.shape[1] and weight.shape[0] % 4 == 0)
    if not ok:
        return conv2d(torch.cat([x0, x1], dim=1), weight, bias, 1, 0, residual=residual, emit_colstats=emit_colstats)
    _C.require_device(x0, x1, weight, bias, residual)
    lib = _C.load()
    if not x0.is_contiguous(memory_format=torch.channels_last):
        x0 = x0.contiguous(memory_format=torch.channels_last)
    if not x1.is_contiguous(memory_format=torch.channels_last):
        x1 = x1.contiguous(memory_format=torch.channels_last)
    w = weight if weight.is_contiguous(memory_format=torch.channels_last) else weight.contiguous(memory_format=torch.channels_last)
    N, C0, H, W = x0.shape
    C1, Cout = x1.shape[1], w.shape[0]
    out = torch.empty((N, Cout, H, W), dtype=x0.dtype, device=x0.device, memory_format=torch.channels_last)
    epi = 0
    if bias is not None:
        epi |= _C.EPI_BIAS
        if bias.dtype != x0.dtype:
            bias = bias.to(x0.dtype)
    if residual is not None:
        if residual.shape != out.shape or residual.dtype != x0.dtype:
            raise BackendError("conv2d_cat: residual must match the output")
        if not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        epi |= _C.EPI_RESIDUAL
    gws = _gemm_workspace(x0.device)
    code = _C.dtype_code(x0.dtype)
    if split_usable(x0.dtype, C0) and C1 % SPLIT_K == 0 and w is weight:
        x0 = _split_of(x0, N * H * W, C0, C0)
        x1 = _split_of(x1, N * H * W, C1, C1)
        w, code = _split_weight(weight, _conv_weight_rows)[0], _C.ST_F32S
    nxt_p, nxt_b = _next_weights(w, N * H * W)
    cbuf = ctiles = crows = None
    if emit_colstats:
        import ctypes
        cbuf, ctiles, crows = _colstats_buffer(N * H * W, Cout, x0.device)
    _C.check(lib.st_conv1x1_cat(x0.data_ptr(), C0, x1.data_ptr(), C1, w.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                N, H, W, Cout, epi, code, gws.data_ptr(), gws.numel(), _ptr(cbuf), ctiles or 0,
                                None if crows is None else ctypes.byref(crows), nxt_p, nxt_b, _C.stream_ptr()), "conv2d_cat")
    if emit_colstats:
        return out, (ColStats(cbuf, crows.value, Cout) if crows.value > 0 else None)
    return out


# ----------------------------------------------------------------------------- loop pieces
TIMESTEP_TABLE_ROWS = 4097      # integer timesteps / sizes 0 .. 4096 (SDXL: 1000 train steps; time_ids = sizes and crops in pixels)
_timestep_tables = {}
_timestep_lock = threading.Lock()


def reference_timestep_features(t: torch.Tensor, dim: int) -> torch.Tensor:
    """The reference's own op sequence for the sinusoidal features (unet_pt.py:22-36), on the host in fp32: cos first, then sin."""
    import math
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32)
    exponent = exponent / (half - 0.0)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _timestep_table(device: torch.device, dim: int) -> torch.Tensor:
    """(TIMESTEP_TABLE_ROWS, dim) fp32 on `device`: row i = the features of timestep i as the reference's eager path computes
    them.  Why a table: t * f_j reaches 1e3 rad, one ulp of exp() moves a feature by 1.2e-4 - two correct fp32 implementations of
    this function disagree by that much, and the difference enters every resnet through the time embedding (round 5: it was the
    largest single term of the strict mode's deviation from the reference).  Built once per (device, width) by the host."""
    key = (device.type, device.index, dim)
    tbl = _timestep_tables.get(key)
    if tbl is None:
        with _timestep_lock:
            tbl = _timestep_tables.get(key)
            if tbl is None:
                with torch.no_grad():
                    host = reference_timestep_features(torch.arange(TIMESTEP_TABLE_ROWS, dtype=torch.float32), dim)
                tbl = _timestep_tables[key] = host.to(device)
    return tbl


def timestep_features(t: torch.Tensor, dim: int, dtype: torch.dtype, step: Optional[torch.Tensor] = None,
                      batch: Optional[int] = None, t_stride: int = 1) -> torch.Tensor:
    _C.require_device(t)
    lib = _C.load()
    t32 = t if t.dtype == torch.float32 else t.float()
    t32 = t32.contiguous()
    nb = t32.numel() if batch is None else batch
    out = torch.empty((nb, dim), dtype=dtype, device=t.device)
    if torch.cuda.is_current_stream_capturing() and (t.device.type, t.device.index, dim) not in _timestep_tables:
        tbl_p, tbl_n = None, 0          # (no host-to-device copy inside somebody's capture: the first eager call builds the table)
    else:
        tbl = _timestep_table(t.device, dim)
        tbl_p, tbl_n = tbl.data_ptr(), tbl.shape[0]
    _C.check(lib.st_timestep_features(t32.data_ptr(), t_stride, _ptr(step), out.data_ptr(), nb, dim,
                                      _C.dtype_code(dtype), tbl_p, tbl_n, _C.stream_ptr()), "timestep_features")
    return out


def timestep_sincos(x: torch.Tensor):
    """The reference's timestep operator (kernels/timestep.py:13-45): x is (..., half), already broadcast;
    returns (sin(x*f_j), cos(x*f_j)) with f_j = exp(-ln(1e4) * j / half) along the last dimension."""
    _C.require_device(x)
    lib = _C.load()
    x32 = x.float().contiguous()
    s, c = torch.empty_like(x32), torch.empty_like(x32)
    _C.check(lib.st_timestep_sincos(x32.data_ptr(), s.data_ptr(), c.data_ptr(), x32.numel(), x32.shape[-1], _C.stream_ptr()),
             "timestep_sincos")
    return s, c


def euler_step(latent: torch.Tensor, eps: torch.Tensor, next_in: torch.Tensor, dsigma: torch.Tensor,
               in_scale: torch.Tensor, step: torch.Tensor) -> None:
    """In place: latent (fp32) += eps*dsigma[*step]; next_in = latent*in_scale[*step+1]."""
    _C.require_device(latent, eps, next_in, dsigma, in_scale, step)
    lib = _C.load()
    if latent.dtype != torch.float32 or eps.dtype != next_in.dtype:
        raise BackendError("euler_step: latent must be fp32 and eps/next_in share a dtype")
    if not (latent.stride() == eps.stride() == next_in.stride()) or latent.numel() != eps.numel():
        raise BackendError("euler_step: latent, eps and next_in must share one dense layout")
    _C.check(lib.st_euler_step(latent.data_ptr(), eps.data_ptr(), next_in.data_ptr(), dsigma.data_ptr(),
                               in_scale.data_ptr(), step.data_ptr(), latent.numel(), dsigma.numel(),
                               _C.dtype_code(eps.dtype), _C.stream_ptr()), "euler_step")


def step_advance(step: torch.Tensor, n_steps: int) -> None:
    _C.require_device(step)
    _C.check(_C.load().st_step_advance(step.data_ptr(), n_steps, _C.stream_ptr()), "step_advance")
