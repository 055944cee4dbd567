"""Drop-in replacement for `diff_gaussian_rasterization` as used by the reference.

Mirrors the call protocol of tgs/models/renderer_one_shot.py:281-296, :338-346, :355-379:

    raster_settings = GaussianRasterizationSettings(image_height=..., image_width=..., tanfovx=..., tanfovy=...,
        bg=..., scale_modifier=..., viewmatrix=..., projmatrix=..., sh_degree=..., campos=...,
        prefiltered=False, debug=False)
    rendered_image, radii = GaussianRasterizer(raster_settings=raster_settings)(
        means3D=..., means2D=..., shs=..., colors_precomp=..., opacities=..., scales=..., rotations=...,
        cov3D_precomp=None)

plus `rasterize_views`, the view-batched entry with the attribute blend of :298-334 fused into the
kernels. All compute goes through the C-ABI of include/gh_raster.h (hand-written HIP for gfx950);
there is no PyTorch/CPU fallback — a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Dict, NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

from . import _abi, _lib
from .camera import pack_camera


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GhOverflowError(RuntimeError):
    """The tile-instance capacity (max_instances) was too small for a sync-free call."""


class GhStaleGeometryError(GhOverflowError):
    """A static-geometry call (gh_forward_refresh) found an opacity above the bound its tile lists were built for: the lists
    may miss a tile. The call returned a NaN image (device-side guard, same mechanism as an instance overflow); every
    GeometryCache has been cleared, so re-running the step rebuilds the lists."""


class GhDepthBoundMiss(GhOverflowError):
    """A call with a speculative occlusion bound (DepthBoundCache) met a pixel that ran off the end of its truncated tile list: the
    Gaussians moved further than the margin allows. That pixel is NaN in the returned image; the bound has been dropped, re-run
    the step (it renders without a bound and produces a fresh one)."""


class DepthBoundCache:
    """Speculative per-tile occlusion bound for loops whose Gaussians move a LITTLE between steps — the one-shot fit's network-
    side trainables move them every step (infer_one_shot.py:340-343), so the static tile lists of GeometryCache cannot serve it.
    Every forward through the cache reports, per tile, the depth of the last list entry any pixel looked at (tiles whose pixels
    all reached the early stop; +inf elsewhere), times (1 + margin); the NEXT forward does not list instances behind it — on the
    hand scenes more than half of all instances lie behind a saturated surface and were emitted, sorted and gathered for nothing.
    The forward verifies the speculation per pixel (GhInputs.tile_depth_bound): where it holds the result is the unbounded
    call's bit for bit; where it fails the pixel is NaN, GhCounters.overflow bit 2 is set and the host re-runs without the bound
    (sync=True: transparently; sync-free: GhDepthBoundMiss from the backward / check_overflow). Not used inside captured graphs
    (the two buffers alternate between calls). margin: relative depth margin for the motion between two steps; slack: list
    entries kept behind the last one any pixel of the tile looked at. refresh_every: a forward that REPORTS a bound runs the
    kernel variant that walks on behind every pixel's stop until its transmittance has halved again (what tells a robustly
    saturated tile from one a rounding away from needing its whole list) — 26 % more forward time, measured; the bound is
    therefore refreshed every refresh_every-th call only and re-used in between (every call still verifies it; the margin has to
    cover the motion of that many steps). min_pixels: calls that render fewer pixels in all ignore the cache (the launch floors of a
    small call leave nothing for the bound to win: one 512x334 view 240 -> 257 us with it)."""

    def __init__(self, margin: float = 2e-3, slack: int = 8, refresh_every: int = 4, min_pixels: int = 4 * 512 * 334):
        self.margin, self.slack, self.refresh_every = float(margin), int(slack), max(1, int(refresh_every))
        self.min_pixels = int(min_pixels)        # calls smaller than this render without a bound (measured: one or two 512x334 views lose)
        self.bufs, self.key, self.cur, self.valid = None, None, 0, False
        self._cams, self._cams_version = None, None
        self.bounded_calls, self.misses, self.age = 0, 0, 0

    def clear(self) -> None:
        self.valid = False

    def _buffers(self, dev, NV: int, H: int, W: int, key=None):
        """(bound to apply or None, buffer to report into or None) for the next call; advances the cache's state.
        key = (P, per_view, packed camera tensor): a bound reported for OTHER cameras, another Gaussian count or the other
        Gaussian layout says nothing about this call (it would stay exact — the forward verifies — but miss on nearly every
        tile: a re-render under sync=True, a skipped step under sync=False), so the cache starts over when any of them changes.
        The cameras are identified by the tensor OBJECT and its version: pass the same packed tensor (camera.pack_cameras_from_w2c)
        every step for the bound to carry over."""
        P, per_view, cams = key if key is not None else (None, None, None)
        cam_ref = self._cams() if self._cams is not None else None
        same_cams = cams is None or (cam_ref is cams and self._cams_version == cams._version)
        full = (dev, NV, H, W, P, per_view)
        if self.key != full or not same_cams:
            T = NV * ((W + 15) // 16) * ((H + 15) // 16)
            if self.key is None or self.key[:4] != full[:4]:
                self.bufs = (torch.empty(T, 2, dtype=torch.float32, device=dev), torch.empty(T, 2, dtype=torch.float32, device=dev))   # (depth, block mask)
            self.key, self.cur, self.valid, self.age = full, 0, False, 0
            self._cams = None if cams is None else weakref.ref(cams)
            self._cams_version = None if cams is None else cams._version
        if self.valid and self.age + 1 < self.refresh_every:      # re-use the bound, plain kernels
            self.age += 1
            self.bounded_calls += 1
            return self.bufs[self.cur], None
        bound, seen = (self.bufs[self.cur] if self.valid else None), self.bufs[1 - self.cur]
        # the buffer this call writes is the next call's bound — also after a miss: a tile whose pixels did not all stop reports
        # +inf, so what the call leaves behind is a valid bound source either way
        self.cur, self.valid, self.age = 1 - self.cur, True, 0
        self.bounded_calls += 1 if bound is not None else 0
        return bound, seen


class GeometryCache:
    """Static geometry for calls that render the SAME Gaussians (means3D, scales, rotations, xyz_b) from the SAME cameras again
    and again while only opacities and colours move — the one-shot fit (infer_one_shot.py:489-524). The first call through a
    cache is a full forward with GH_FLAG_STATIC_LISTS; its context is kept here, and later calls whose geometry inputs are
    the very same tensor objects, unmodified (`is` + `_version`), go through gh_forward_refresh: no projection, no sorts.
    Anything else — another tensor, a bumped version, another image size — is a miss: a full call that replaces the entry.
    `clear()` forces the next call to rebuild (call it after changing a geometry tensor in place through `.data`, which
    `_version` cannot see). Opt-in: a caller passes its cache to rasterize_views / rendered_*_loss."""
    _all = weakref.WeakSet()

    def __init__(self):
        self.ctx, self.key, self.refs = None, None, None
        self.hits, self.builds = 0, 0
        GeometryCache._all.add(self)

    def clear(self) -> None:
        self.ctx, self.key, self.refs = None, None, None

    @staticmethod
    def clear_all() -> None:
        for c in list(GeometryCache._all):
            c.clear()

    def lookup(self, objs, vals):
        """The cached context if `objs` are the cached tensors, unmodified, and `vals` the cached scalars."""
        if self.ctx is None or self.key != vals or len(self.refs) != len(objs):
            return None
        for r, o in zip(self.refs, objs):
            if (r is None) != (o is None) or (r is not None and r() is not o):
                return None
        return self.ctx

    def store(self, objs, vals, ctx) -> None:
        self.refs = tuple(None if o is None else weakref.ref(o) for o in objs)
        self.key, self.ctx = vals, ctx


# ---------------------------------------------------------------------------------------------------
# What the rasteriser remembers between calls. Two kinds, kept apart (round 5):
#   * _DeviceState — per DEVICE, behind one re-entrant lock: learned instance capacities, the GH_FLAG_DEPTH24 verdicts, outstanding
#     counter read-backs, the workspace pool, the geometry record of the two-call protocol, graph-mode counters, the latest
#     workspace / gradient block. Two devices in one process share nothing; two threads on one device (autograd runs `backward` on
#     a thread of its own) take the lock around every read-modify-write of it.
#   * _Policy — process-wide SWITCHES a program sets once (graph mode, split-stream policy, geometry reuse, stage timing): plain
#     configuration, no per-call state.
# The module attributes earlier rounds' tests and tools read (`_capacity`, `_depth24`, `_geom_last`, `_graph_counters`, `_pending`,
# `_split_policy` ...) resolve to the CURRENT device's state through the module __getattr__ at the end of this file.
class _DeviceState:
    def __init__(self, index: int):
        self.index = index
        self.lock = threading.RLock()
        # capacity policy for the data-dependent instance count D: call shape -> learned max_instances
        self.capacity: Dict[Tuple[int, int, int, int, bool], int] = {}
        # GH_FLAG_DEPTH24 (three depth-sort passes instead of four: two launches less per forward) per call shape. The flag is a
        # speculation the device can only answer with a NaN image, so it is used ONLY for a shape a read-back has shown it to hold
        # for (GhCounters.overflow bit 4, reported by every full forward, also one made WITHOUT the flag): absent = unknown = four
        # passes; True = observed to hold; False = observed to fail once (bit 3, or bit 4 missing): four passes from then on.
        self.depth24: Dict[Tuple[int, int, int, int, bool], bool] = {}
        self.pending = []        # _Pending records of sync-free calls not yet checked
        self.free_slots = []     # recycled (pinned 4-int buffer, event) pairs: a sync-free call allocates neither
        self.last_D = 0
        self.last_ws = None      # workspace of the most recent forward (its first 16 bytes are the GhCounters)
        self.graph_counters = {} # workspace address -> (device view of GhCounters, cap, key) of forwards issued in graph mode
        self.ws_pool: Dict[Tuple[int, int], list] = {}
        self.geom_last = None    # (identity objects, values, weakref to the context) of the latest full drop-in forward
        self.last_grad_block = None
        self.stage_events = []   # (stage name, start event, end event)


class _Policy:
    graph_mode = False
    split = False                # False | True | "auto"
    reuse_geometry = True
    stage_timing = False
    # raster_forward(l1_target=... / fit_loss=...) takes the render kernel's epilogue (GhOutputs.l1_* / fit_loss) where the library
    # offers it; GH_FUSED_LOSS=0 in the environment: measurement A/B (set_fused_loss)
    fused_loss = os.environ.get("GH_FUSED_LOSS", "1") != "0"


_policy = _Policy()
_states: Dict[int, _DeviceState] = {}
_states_lock = threading.Lock()


def _dev_index(dev=None) -> int:
    if dev is None:
        return torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(dev, int):
        return dev
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _state(dev=None) -> _DeviceState:
    """The rasteriser's state for `dev` (a torch.device, an index, or None = the current device)."""
    i = _dev_index(dev)
    st = _states.get(i)
    if st is None:
        with _states_lock:
            st = _states.setdefault(i, _DeviceState(i))
    return st


def _learn_depth24(st: _DeviceState, key, flags_word: int, d: int = 1) -> None:
    """Update the GH_FLAG_DEPTH24 verdict of a call shape from a counter word that has been read back — the word of a FULL forward
    whose depth sort ran (callers pass nothing else: a refresh, a shared call or a call with nothing to project sorted nothing, and
    a word that says nothing must not decide anything). d: the call's instance count; with d = 0 no key took part in
    the (OR, AND) of the key bits, so "the top byte did not vary" is vacuous and "bit 4 absent" impossible: unknown stays unknown."""
    if flags_word & 8:
        st.depth24[key] = False
    elif d == 0:
        return
    elif flags_word & _abi.GH_COUNTER_DEPTH24_OK:
        st.depth24.setdefault(key, True)
    elif not (flags_word & 1):             # a complete call whose depths' top byte varies (truncated lists prove nothing)
        st.depth24[key] = False


_DEPTH24_MSG = ("the visible depths span more than the 24 key bits the three-pass depth sort covers; the call returned a NaN image; "
                "the four-pass sort is used for this call shape from now on, re-run the step")
_PENDING_MAX = 64


class _Pending:
    """The asynchronous counter read-back of one sync-free forward. `resolve()` waits for it (normally long done), recycles
    the pinned buffer and remembers the verdict, so that both check_overflow() and the call's own backward can ask."""
    __slots__ = ("st", "ev", "host", "cap", "key", "done", "d", "over", "stale", "miss", "wide", "dbound", "told", "learn24")

    def __init__(self, st, ev, host, cap, key, dbound=None, learn24=True):
        self.st, self.ev, self.host, self.cap, self.key, self.done, self.d, self.over, self.stale = st, ev, host, cap, key, False, 0, False, False
        self.miss, self.wide, self.dbound = False, False, dbound
        self.learn24 = learn24     # a full forward (the depth sort ran): its counter word says whether GH_FLAG_DEPTH24 holds
        self.told = False          # the caller has been given this record's error (a look at the counters by the geometry-reuse check is not that)

    def resolve(self) -> bool:
        """True when the call overflowed its capacity (the learned capacity of its shape is raised then)."""
        st = self.st
        with st.lock:
            if not self.done:
                self.ev.synchronize()
                c = self.host.tolist()
                self.d = c[0] & 0xFFFFFFFF
                st.last_D = self.d
                # overflowed = the device-side error bits, nothing else; a split call's reserved[0] (the max_instances that would
                # give each half a large enough share) only sizes the NEXT capacity
                self.over = (c[1] & _abi.GH_COUNTER_ERROR_MASK) != 0
                self.stale = (c[1] & 2) != 0
                self.miss = (c[1] & 4) != 0
                self.wide = (c[1] & 8) != 0                    # GH_FLAG_DEPTH24 did not hold: four passes for this shape from now on
                if self.learn24:
                    _learn_depth24(st, self.key, c[1], self.d)
                if self.miss and self.dbound is not None:      # the speculation failed: the re-run of the step renders without a bound
                    self.dbound.clear()
                    self.dbound.misses += 1
                self.dbound = None
                need = (c[2] & 0xFFFFFFFF) if self.key[-1] else self.d
                if self.over and (c[1] & 3):           # stale lists, or lists truncated by an instance overflow: never re-use them
                    GeometryCache.clear_all()
                if self.over and (c[1] & 1):
                    st.capacity[self.key] = max(st.capacity.get(self.key, 0), int(max(need, self.d) * 1.5) + 1024)
                st.free_slots.append((self.host, self.ev))
                self.host = self.ev = None
                self.done = True
        return self.over

    def message(self) -> str:
        if self.stale:
            return _STALE_MSG
        if self.miss and not (self.over and self.d > self.cap):
            return _MISS_MSG
        if not self.d > self.cap and self.wide and not self.stale and not self.miss:
            return _DEPTH24_MSG
        return (f"tile instances D={self.d} exceeded max_instances={self.cap}; the call returned a NaN image; "
                "capacity raised, re-run the step")

    def error(self) -> GhOverflowError:
        self.told = True
        if self.stale:
            return GhStaleGeometryError(self.message())
        if self.miss and not self.d > self.cap:
            return GhDepthBoundMiss(self.message())
        return GhOverflowError(self.message())


_STALE_MSG = ("an opacity rose above the bound the static tile lists were built for; the call returned a NaN image; "
              "the geometry caches are cleared, re-run the step (it rebuilds the lists)")


_MISS_MSG = ("a pixel ran off the end of a tile list truncated by the speculative occlusion bound (the Gaussians moved further than "
             "the margin); that pixel is NaN; the bound has been dropped, re-run the step")


def last_guard(dev=None):
    """Device view of the GhCounters of the most recent forward on `dev` (None = the current device): the `guard` argument of
    gh_l1_loss / gh_fit_loss / gh_adam_reg_step (device-side overflow guard). Holds that workspace alive until the next forward."""
    ws = _state(dev).last_ws
    return None if ws is None else ws[:16]


def last_num_rendered(dev=None) -> int:
    """Tile instances D of the most recent forward whose counters have been read back."""
    return _state(dev).last_D


def enable_stage_timing(on: bool) -> None:
    """Bracket every pipeline stage with HIP events on the launch stream (bench.py's roofline leg)."""
    _policy.stage_timing = bool(on)
    if on:
        for st in list(_states.values()):
            st.stage_events.clear()


def stage_timing_summary() -> Dict[str, float]:
    """Average milliseconds per launch of each stage since enable_stage_timing(True) (current device)."""
    torch.cuda.synchronize()
    acc: Dict[str, list] = {}
    for name, e0, e1 in _state().stage_events:
        acc.setdefault(name, []).append(e0.elapsed_time(e1))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def _run_stages(st: _DeviceState, fn, args, stages):
    """Call a *_stages entry point once per stage with events in between (same stream, same order)."""
    rc = 0
    for name, bit in stages:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args, C.c_uint32(bit))
        e1.record()
        st.stage_events.append((name, e0, e1))
        if rc != 0:
            break
    return rc


def set_graph_mode(on: bool) -> None:
    """Graph mode: sync-free forwards issue no counter read-back at all (nothing but kernel launches and async
    memsets reaches the stream), so a whole step can be captured with torch.cuda.CUDAGraph / hipGraph.
    check_overflow() then reads the counters of the captured workspaces directly."""
    _policy.graph_mode = bool(on)
    if not on:
        for st in list(_states.values()):
            with st.lock:
                st.graph_counters.clear()


def graph_counters(dev=None):
    """[(device view of GhCounters, max_instances, call-shape key)] of the forwards issued in graph mode on `dev` so far — what a
    capture keeps to check its replays (fit.CapturedFitStep)."""
    st = _state(dev)
    with st.lock:
        return list(st.graph_counters.values())


_SPLIT_AUTO_MIN_PIXELS = 1 << 19   # "auto": where the split measured a gain (1024x1024 x 8 views: +4.9 %; 512x334: 8 views -1 %, 16 views +0.8 %)


def set_split_streams(mode) -> None:
    """Module policy for calls that do not say (split_streams=None): False = one stream, True = split whenever n_views >= 2,
    "auto" = split where it measured a gain: four or more views of more than half a megapixel each (GH_FLAG_SPLIT_STREAMS;
    results are bit-identical either way)."""
    if mode not in (False, True, "auto"):
        raise ValueError("set_split_streams: False, True or 'auto'")
    _policy.split = mode


def set_fused_loss(on: bool) -> None:
    """Module policy: let raster_forward(l1_target=...) — i.e. loss.rendered_l1_loss — take the image loss from the render kernel's
    own epilogue (GhOutputs.l1_*, the default) or always run gh_l1_loss on the stored image (measurement A/B; same gradients bit for
    bit, the loss up to the order of its float32 sums)."""
    _policy.fused_loss = bool(on)


def capacity_key(P: int, NV: int, H: int, W: int, split: bool = False):
    """Key of the learned instance capacity of a call shape (tests / tools)."""
    return (P, NV, H, W, bool(split))


def _initial_capacity(P: int, NV: int) -> int:
    return max(1 << 16, 8 * P * NV)


def report_counter_word(key, flags_word: int, d: int, cap: int, reserved0: int = 0, dev=None, where: str = "", learn24: bool = True) -> None:
    """Act on a GhCounters.overflow word read back from a workspace the caller holds itself (a captured graph's, see
    fit.CapturedFitStep.check): learn what it says (capacity, GH_FLAG_DEPTH24 verdict, stale caches) and raise the matching error.
    learn24: the word is a FULL forward's (its depth sort ran); a refresh over static lists passes False."""
    st = _state(dev)
    with st.lock:
        st.last_D = d
        if learn24 or (flags_word & 8):
            _learn_depth24(st, key, flags_word, d)
        if flags_word & 2:                                  # a static-geometry replay met an opacity above its lists' bound
            GeometryCache.clear_all()
            raise GhStaleGeometryError(_STALE_MSG + where)
        if flags_word & 8:                                  # GH_FLAG_DEPTH24 did not hold for a captured call
            GeometryCache.clear_all()
            raise GhOverflowError(_DEPTH24_MSG + where)
        if flags_word & _abi.GH_COUNTER_ERROR_MASK:         # the device-side flag decides; reserved[0] of a split call sizes the next capacity
            need = max(d, reserved0 if key[-1] else d)
            st.capacity[key] = max(st.capacity.get(key, 0), int(need * 1.5) + 1024)
            GeometryCache.clear_all()                       # lists truncated by the overflow must not be refreshed again
            raise GhOverflowError(f"tile instances D={d} exceeded max_instances={cap}" + (where or " inside a captured graph"))


def check_overflow(block: bool = True, keep_recent: int = 0, dev=None) -> None:
    """Verify every outstanding sync-free forward on `dev` (None = the current device) fitted its capacity (raises
    GhOverflowError). block=False only looks at read-backs that have arrived — except that all but the `keep_recent` latest
    calls are waited for regardless (they finished long ago), which bounds how late an overflow can surface."""
    st = _state(dev)
    with st.lock:
        for counters, cap, key, full in list(st.graph_counters.values()):   # graph mode: workspaces are static, read them directly
            c4 = counters.tolist()
            report_counter_word(key, c4[1], c4[0] & 0xFFFFFFFF, cap, c4[2] & 0xFFFFFFFF, dev=st.index,
                                where=" [inside a captured graph: capture again]" if (c4[1] & 10) else "", learn24=full)
        keep, bad = [], []
        n_old = len(st.pending) - keep_recent if keep_recent > 0 else 0
        for i, pc in enumerate(st.pending):
            if pc.done:
                if pc.over and not pc.told:                    # resolved by somebody who did not report it (the geometry-reuse check)
                    bad.append(pc)
                continue
            if not block and i >= n_old and not pc.ev.query():
                keep.append(pc)
                continue
            if pc.resolve():
                bad.append(pc)
        if bad:
            # ONE error per call, the most severe first: stale lists / a capacity overflow ask for a rebuild or a larger
            # workspace, an occlusion-bound miss or a depth-key verdict only for the step again; the records not reported now stay
            # in the list (done, untold) and surface at the next check instead of being dropped.
            sev = lambda p: 3 if p.stale else (2 if (p.over and p.d > p.cap) else (1 if p.wide and not p.miss else 0))
            worst = max(bad, key=sev)
            st.pending = [p for p in bad if p is not worst] + keep
            raise worst.error()
        st.pending = keep


# Workspace pool: a forward takes its workspace from here and the context gives it back when it dies (after its backward,
# or when a no-grad caller drops it), so steady-state calls allocate nothing. Per device, keyed by (stream, bytes): a workspace
# only ever returns to the stream that used it last, and that stream orders its next use behind the previous one (callers
# that render on several streams at once — e.g. two half-batches overlapped in one captured graph — get one set per stream).
_WS_POOL_DEPTH = 4


def _ws_acquire(st: _DeviceState, dev: torch.device, nbytes: int, stream: int) -> torch.Tensor:
    with st.lock:
        free = st.ws_pool.get((stream, nbytes))
        if free:
            return free.pop()
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


def _ws_release(ws: Optional[torch.Tensor], stream: int) -> None:
    if ws is None:
        return
    st = _state(ws.device)
    with st.lock:
        free = st.ws_pool.setdefault((stream, ws.numel()), [])
        if len(free) < _WS_POOL_DEPTH:
            free.append(ws)


def clear_workspace_pool() -> None:
    for st in list(_states.values()):
        with st.lock:
            st.ws_pool.clear()


def _raw_stream(dev: torch.device) -> int:
    """Handle of the caller's current stream on `dev` (the torch.cuda.current_stream() object costs ~10 us of host time)."""
    return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())


class _OnDevice:
    """`with torch.cuda.device(dev)` only when `dev` is not already current (the context manager costs ~10 us per call)."""
    __slots__ = ("ctx",)

    def __init__(self, dev: torch.device):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.ctx = None if torch.cuda.current_device() == idx else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _ptr(t: Optional[torch.Tensor]):
    # (a plain int: a ctypes pointer FIELD takes it as it stands — thirty c_void_p objects per call were thirty GC-tracked
    # allocations per call for nothing)
    return None if t is None else t.data_ptr()


def _prep(t: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.device != dev:
        raise ValueError(f"tensor on {t.device}, expected {dev}")
    if t.dtype is torch.float32 and t.is_contiguous():          # the common case: only the pointer is needed
        return t
    return t.detach().to(torch.float32).contiguous()


class _Ctx:
    __slots__ = ("dims", "inp", "tensors", "ws", "layout", "H", "W", "P", "NV", "M", "wpg", "b_rgb", "rows", "stream", "alpha",
                 "parent", "radii", "pending", "verdict", "refresh", "l1", "fit", "defer_loss", "__weakref__")

    def __del__(self):
        try:
            _ws_release(getattr(self, "ws", None), getattr(self, "stream", 0))
        except Exception:                      # interpreter shutdown
            pass


class _Call(NamedTuple):
    """One call's arguments after validation: prepared tensors and the shape / flag words derived from them."""
    t: dict
    rows: int
    NV: int
    P: int
    M: int
    flags: int
    wpg: bool
    b_rgb: bool
    split: bool
    H: int
    W: int
    sh_degree: int
    scale_modifier: float
    cams_obj: object          # the caller's camera tensor as passed (identity: DepthBoundCache)


def _prepare_call(dev, cams, means3D, opacities, scales, rotations, H, W, shs, colors_precomp, sh_degree, scale_modifier, xyz_b,
                  opacity_b, color_w, color_b, per_view_gaussians, split_streams, plain: bool, cov3D_precomp=None) -> _Call:
    """Validate the arguments of raster_forward and derive rows / P / flags. plain: a full forward without static lists (the only
    kind that may be split over two streams)."""
    if (shs is None) == (colors_precomp is None):
        raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    t = dict(cov3D=_prep(cov3D_precomp, dev), cams=_prep(cams, dev).reshape(-1, _abi.GH_CAM_FLOATS), means3D=_prep(means3D, dev),
             opacities=_prep(opacities, dev).reshape(-1), scales=_prep(scales, dev), rotations=_prep(rotations, dev),
             shs=_prep(shs, dev), colors_precomp=_prep(colors_precomp, dev), xyz_b=_prep(xyz_b, dev),
             opacity_b=None if opacity_b is None else _prep(opacity_b, dev).reshape(-1),
             color_w=_prep(color_w, dev), color_b=_prep(color_b, dev))
    rows, NV = t["means3D"].shape[0], t["cams"].shape[0]
    M = 0 if shs is None else t["shs"].shape[1]
    flags = 0
    if per_view_gaussians:
        if NV == 0 or rows % NV:
            raise ValueError("per_view_gaussians: the Gaussian tensors must hold n_views * P rows")
        flags |= _abi.GH_FLAG_PER_VIEW_GAUSSIANS
    P = rows // NV if per_view_gaussians else rows
    if split_streams is None:
        split_streams = _policy.split is True or (_policy.split == "auto" and NV >= 4 and H * W > _SPLIT_AUTO_MIN_PIXELS)
    split = bool(split_streams) and NV >= 2 and P > 0 and plain and not _policy.stage_timing
    if split:
        flags |= _abi.GH_FLAG_SPLIT_STREAMS
    wpg = False
    if t["color_w"] is not None:
        if t["color_w"].numel() == 48:
            pass
        elif t["color_w"].numel() == rows * 48:
            flags |= _abi.GH_FLAG_BLEND_W_PER_GAUSSIAN
            wpg = True
        else:
            raise ValueError("color_w must have 48 or P*48 elements")
    b_rgb = False
    if t["color_b"] is not None:
        if t["color_b"].numel() == rows * 3 and colors_precomp is not None:  # the 3 columns RGB mode reads (renderer_one_shot.py:328)
            flags |= _abi.GH_FLAG_BLEND_COLOR_B_RGB
            b_rgb = True
        elif t["color_b"].numel() != rows * 48:
            raise ValueError("color_b must have P*48 elements (or P*3 with colors_precomp)")
    return _Call(t, rows, NV, P, M, flags, wpg, b_rgb, split, int(H), int(W), int(sh_degree), float(scale_modifier), cams)


def _inputs_struct(c: _Call, with_shs: bool = True, bound=None):
    t = c.t
    return _abi.GhInputs(_ptr(t["cams"]), _ptr(t["means3D"]), _ptr(t["opacities"]), _ptr(t["scales"]), _ptr(t["rotations"]),
                         _ptr(t["shs"]) if with_shs else None, _ptr(t["colors_precomp"]), _ptr(t["xyz_b"]), _ptr(t["opacity_b"]),
                         _ptr(t["color_w"]), _ptr(t["color_b"]), _ptr(bound), _ptr(t["cov3D"]))


def _make_ctx(c: _Call, dims, inp, ws, stream, alpha, parent, radii, pending, verdict, refresh, l1=None, fit=None, defer=False) -> _Ctx:
    ctx = _Ctx()
    ctx.dims, ctx.inp, ctx.tensors, ctx.ws, ctx.H, ctx.W, ctx.P, ctx.NV, ctx.M, ctx.wpg = dims, inp, c.t, ws, c.H, c.W, c.P, c.NV, c.M, c.wpg
    ctx.b_rgb, ctx.rows, ctx.alpha, ctx.parent, ctx.radii, ctx.stream = c.b_rgb, c.rows, alpha, parent, radii, stream
    ctx.pending, ctx.verdict, ctx.refresh = pending, verdict, refresh
    ctx.l1 = l1                                    # (loss, dL/dimage) of a fused image loss (GhOutputs.l1_*), else None
    ctx.fit = fit                                  # (loss, dL/dimage, dL/dalpha) of a fused fit loss (GhOutputs.fit_loss), else None
    # GH_FLAG_DEFER_LOSS_SUM: the fused loss's final sum is the backward's to write (GhGrads.deferred_loss)
    ctx.defer_loss = bool(defer) and (l1 is not None or fit is not None)
    return ctx


def _queue_readback(st: _DeviceState, counters, cap, key, dbound=None, learn24=True) -> _Pending:
    """Asynchronous copy of a call's GhCounters into a pinned buffer + an event: the record of a sync-free call."""
    host, ev = st.free_slots.pop() if st.free_slots else (torch.empty(4, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
    host.copy_(counters, non_blocking=True)
    ev.record()
    pc = _Pending(st, ev, host, cap, key, dbound, learn24)
    st.pending.append(pc)
    return pc


def _fused_loss(dev, image, alpha, l1_target, fit_loss, allowed: bool):
    """GhOutputs' trailing fields for a fused image loss (l1_target / l1_dL_dimage / l1_loss / fit_loss) + what the context keeps:
    ((loss, dL/dimage) or None, (loss, dL/dimage, dL/dalpha) or None, the four GhOutputs fields). The combinations the library does
    not fuse (allowed False, or the L1 form with a mask channel, or the fit form without one) yield (None, None, no fields): the caller
    runs gh_l1_loss / gh_fit_loss on the stored images (loss.py does)."""
    none = (None, None, (None, None, None, None))
    if not allowed or not _policy.fused_loss:
        return none
    if l1_target is not None and alpha is None:
        if l1_target.shape != image.shape or l1_target.dtype is not torch.float32 or not l1_target.is_contiguous():
            raise ValueError("l1_target must be a contiguous float32 (n_views, 3, H, W) tensor")
        l1 = (torch.empty((), dtype=torch.float32, device=dev), torch.empty_like(image))
        return l1, None, (_ptr(l1_target), _ptr(l1[1]), _ptr(l1[0]), None)
    if fit_loss is not None and alpha is not None:
        gt_rgb, gt_mask, bbox, lam_l1, lam_m, scale = fit_loss
        NV, _, H, W = image.shape
        for t, shape in ((gt_rgb, (NV, H, W, 3)), (gt_mask, (NV, H, W)), (bbox, (NV, H, W))):
            if t is not None and (tuple(t.shape) != shape or t.dtype is not torch.float32 or not t.is_contiguous()):
                raise ValueError("fit_loss: gt_rgb (n_views,H,W,3), gt_mask and bbox (n_views,H,W) must be contiguous float32 tensors")
        fit = (torch.empty((), dtype=torch.float32, device=dev), torch.empty_like(image), torch.empty_like(alpha))
        rec = _abi.GhFitLoss(_ptr(gt_rgb), _ptr(gt_mask), _ptr(bbox), float(lam_l1), float(lam_m), float(scale), _ptr(fit[1]), _ptr(fit[2]),
                             _ptr(fit[0]))
        return None, fit, (None, None, None, C.pointer(rec))          # (the pointer object keeps `rec` alive through the call)
    return none


def _forward_shared(st, L, dev, c: _Call, g0: _Ctx, return_alpha: bool):
    """gh_forward_shared: a second call over the lists of `g0` (the reference's mask pass after the RGB pass of a view)."""
    if c.t["shs"] is not None or (g0.dims.flags & _abi.GH_FLAG_SPLIT_STREAMS) or (g0.P, g0.NV, g0.H, g0.W, g0.rows) != (c.P, c.NV, c.H, c.W, c.rows) or \
            (g0.dims.flags & _abi.GH_FLAG_PER_VIEW_GAUSSIANS) != (c.flags & _abi.GH_FLAG_PER_VIEW_GAUSSIANS):
        raise ValueError("geometry_of: the second call must have the first one's shapes and precomputed colours")
    cap = int(g0.dims.max_instances)
    dims = _abi.GhDims(c.P, c.NV, c.H, c.W, c.sh_degree, c.M, c.scale_modifier, c.flags, cap)
    nbytes = L.gh_workspace_bytes(C.byref(dims))
    stream = _raw_stream(dev)
    ws = _ws_acquire(st, dev, nbytes, stream)
    image = torch.empty(c.NV, 3, c.H, c.W, dtype=torch.float32, device=dev)
    alpha = torch.empty(c.NV, c.H, c.W, dtype=torch.float32, device=dev) if return_alpha else None
    inp = _inputs_struct(c, with_shs=False)
    out = _abi.GhOutputs(_ptr(image), None, _ptr(alpha))
    with _OnDevice(dev):
        rc = L.gh_forward_shared(C.byref(dims), C.byref(inp), C.byref(out), C.c_void_p(g0.ws.data_ptr()),
                                 C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"gh_forward_shared failed: {_abi.status_name(rc)}")
    st.last_ws = ws
    # the overflow flag is the geometry owner's; same geometry, same radii; an overflow is the first call's (NaN image here too)
    ctx = _make_ctx(c, dims, inp, ws, stream, alpha, g0, g0.radii, g0.pending, g0.verdict, False)
    return image, g0.radii, ctx


def _forward_refresh(st, L, dev, c: _Call, g0: _Ctx, return_alpha: bool, sync, expect_backward: bool, l1_target=None, fit_loss=None,
                     defer_loss=False):
    """gh_forward_refresh: this call's opacities / colours over the static lists of `g0`."""
    if not (g0.dims.flags & _abi.GH_FLAG_STATIC_LISTS) or (g0.P, g0.NV, g0.H, g0.W, g0.rows) != (c.P, c.NV, c.H, c.W, c.rows) or \
            (g0.dims.flags & _abi.GH_FLAG_PER_VIEW_GAUSSIANS) != (c.flags & _abi.GH_FLAG_PER_VIEW_GAUSSIANS) or g0.parent is not None:
        raise ValueError("refresh_of: needs the context of a static_lists forward with this call's shapes")
    cap = int(g0.dims.max_instances)
    defer = bool(defer_loss) and _policy.fused_loss and ((l1_target is not None and not return_alpha) or (fit_loss is not None and return_alpha))
    dims = _abi.GhDims(c.P, c.NV, c.H, c.W, c.sh_degree, c.M, c.scale_modifier, c.flags | (_abi.GH_FLAG_DEFER_LOSS_SUM if defer else 0), cap)
    nbytes = L.gh_workspace_bytes(C.byref(dims))
    # the library applies THIS call's layout to the owner's workspace, and the arrays a refresh reads of it (cull_bound,
    # inst_c) lie behind the ones sized by the colour mode (sh_rgb, dmean_sh, sh_scratch: M != 0): the two calls must agree
    # on it — and the owner's workspace must span this call's layout
    if (g0.M != 0) != (c.M != 0) or g0.ws.numel() < nbytes:
        raise ValueError("refresh_of: the static lists were built in the other colour mode (shs vs colors_precomp); "
                         "build them again with this call's colour inputs")
    stream = _raw_stream(dev)
    ws = _ws_acquire(st, dev, nbytes, stream)
    image = torch.empty(c.NV, 3, c.H, c.W, dtype=torch.float32, device=dev)
    alpha = torch.empty(c.NV, c.H, c.W, dtype=torch.float32, device=dev) if return_alpha else None
    inp = _inputs_struct(c)
    l1, fit, loss_fields = _fused_loss(dev, image, alpha, l1_target, fit_loss, True)
    out = _abi.GhOutputs(_ptr(image), None, _ptr(alpha), None, 1.0, 0, *loss_fields)
    with _OnDevice(dev):
        rc = L.gh_forward_refresh(C.byref(dims), C.byref(inp), C.byref(out), C.c_void_p(g0.ws.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"gh_forward_refresh failed: {_abi.status_name(rc)}")
    counters = ws[:16].view(torch.int32)
    st.last_ws = ws
    pending, auto = None, sync is None
    gkey = (c.P, c.NV, c.H, c.W, False)
    if sync is None:
        sync = not expect_backward
    if sync:
        c4 = counters.tolist()
        st.last_D = c4[0] & 0xFFFFFFFF
        if c4[1] & 2:
            GeometryCache.clear_all()
            raise GhStaleGeometryError(_STALE_MSG)
        if c4[1] & 1:
            GeometryCache.clear_all()
            raise GhOverflowError("the static tile lists were built by a call that overflowed its capacity; caches cleared")
    elif _policy.graph_mode:
        st.graph_counters[ws.data_ptr()] = (counters, cap, gkey, False)       # (a refresh: no depth sort ran, nothing to learn about it)
    else:
        # a refresh whose opacity guard fired (or whose lists an overflow truncated) poisons every later step too: look at
        # the read-backs that have arrived, so that a stale loop surfaces within two calls instead of _PENDING_MAX
        check_overflow(block=False, keep_recent=2, dev=st.index)
        if len(st.pending) >= _PENDING_MAX:
            check_overflow(block=True, dev=st.index)
        pc = _queue_readback(st, counters, cap, gkey, learn24=False)      # (no depth sort in a refresh: nothing to learn about it)
        if auto:
            pending = pc
    ctx = _make_ctx(c, dims, inp, ws, stream, alpha, g0, g0.radii, pending, None, True, l1, fit, defer)
    return image, g0.radii, ctx


def _forward_full(st, L, dev, c: _Call, return_alpha: bool, sync, expect_backward: bool, max_instances, static_lists: bool,
                  depth_bound, l1_target=None, fit_loss=None, defer_loss=False):
    """gh_forward: projection, both sorts, lists, render — with the capacity / GH_FLAG_DEPTH24 / occlusion-bound policies around it
    (a synced call that the device flags is re-run with what it learned; a sync-free call is recorded for check_overflow)."""
    P, NV, H, W = c.P, c.NV, c.H, c.W
    key = (P, NV, H, W, c.split)
    if depth_bound is not None and (static_lists or _policy.graph_mode or P == 0 or NV * H * W < depth_bound.min_pixels):
        depth_bound = None                       # lists that outlive the call / a captured call / a small call: no per-call speculation
    base_flags = c.flags | (_abi.GH_FLAG_STATIC_LISTS if static_lists else 0)
    # the final sum of a fused image loss inside the backward (only where a loss IS fused: see _fused_loss below)
    if defer_loss and _policy.fused_loss and depth_bound is None and not c.split and \
            ((l1_target is not None and not return_alpha) or (fit_loss is not None and return_alpha)):
        base_flags |= _abi.GH_FLAG_DEFER_LOSS_SUM
    while True:
        cap = int(max_instances) if max_instances is not None else st.capacity.get(key, _initial_capacity(P, NV))
        # three depth-sort passes only for a shape a read-back has shown them to suffice for (_DeviceState.depth24)
        flags = base_flags | (_abi.GH_FLAG_DEPTH24 if st.depth24.get(key) is True else 0)
        dims = _abi.GhDims(P, NV, H, W, c.sh_degree, c.M, c.scale_modifier, flags, cap)
        nbytes = L.gh_workspace_bytes(C.byref(dims))
        if nbytes == 0:
            raise RuntimeError("gh_workspace_bytes rejected the dimensions")
        stream = _raw_stream(dev)
        ws = _ws_acquire(st, dev, nbytes, stream)
        # The launch-order hint in the workspace describes the cameras of the call that left it: other cameras (or a buffer
        # fresh from the allocator) -> GH_FLAG_FRESH_ORDER, the order by this call's own list lengths. Results are the same either way.
        # (the tensor OBJECT through a weak reference + its version counter: the address of a freed camera tensor is what the caching
        #  allocator hands to the next one, and so is a freed object's id())
        # A step being CAPTURED into a graph is replayed over this very buffer with this very camera tensor: from its second replay on
        # the hint is its own, and the flag would be baked into every replay — never set under capture.
        prev = getattr(ws, "_gh_cam_sig", None)
        if (prev is None or prev[0]() is not c.cams_obj or prev[1:] != (c.cams_obj._version, NV, H, W)) \
                and not torch.cuda.is_current_stream_capturing():
            dims.flags |= _abi.GH_FLAG_FRESH_ORDER
        ws._gh_cam_sig = (weakref.ref(c.cams_obj), c.cams_obj._version, NV, H, W)
        image = torch.empty(NV, 3, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(NV, P, dtype=torch.int32, device=dev)
        alpha = torch.empty(NV, H, W, dtype=torch.float32, device=dev) if return_alpha else None
        bound, seen = depth_bound._buffers(dev, NV, H, W, key=(P, bool(c.flags & _abi.GH_FLAG_PER_VIEW_GAUSSIANS), c.cams_obj)) \
            if depth_bound is not None else (None, None)
        inp = _inputs_struct(c, bound=bound)
        # fused image loss: the loss and its gradient(s) from the render kernel's epilogue (GhOutputs.l1_* / fit_loss); the
        # combinations the library does not fuse fall back to gh_l1_loss / gh_fit_loss in loss.py (ctx.l1 / ctx.fit are None)
        l1, fit, loss_fields = _fused_loss(dev, image, alpha, l1_target, fit_loss, depth_bound is None and not c.split)
        out = _abi.GhOutputs(_ptr(image), _ptr(radii), _ptr(alpha), _ptr(seen), 1.0 + (depth_bound.margin if depth_bound is not None else 0.0),
                             depth_bound.slack if depth_bound is not None else 0, *loss_fields)
        with _OnDevice(dev):
            fargs = (C.byref(dims), C.byref(inp), C.byref(out), C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(stream))
            if _policy.stage_timing:
                rc = _run_stages(st, L.gh_forward_stages, fargs, (("preprocess_fwd", _abi.GH_FWD_PREPROCESS),
                                                                   ("binning", _abi.GH_FWD_BINNING),
                                                                   ("render_fwd", _abi.GH_FWD_RENDER)))
            else:
                rc = L.gh_forward(*fargs)
        if rc != 0:
            raise RuntimeError(f"gh_forward failed: {_abi.status_name(rc)}")
        counters = ws[:16].view(torch.int32)
        st.last_ws = ws
        pending, auto, verdict = None, sync is None, None
        do_sync = sync
        if do_sync is None:
            # auto: read D back once per shape to size the capacity, then sync-free — but only for calls whose backward will
            # come (it checks this call's counters before it produces a gradient, see raster_backward). A call outside
            # autograd (inference) has no such second chance and reads D back like the reference wrapper does.
            do_sync = (max_instances is None and key not in st.capacity) or not expect_backward
            if not do_sync:
                check_overflow(block=False, keep_recent=2, dev=st.index)  # an overflow of an earlier sync-free call surfaces at most two calls late
        if do_sync:
            c4 = counters.tolist()                        # the one host read-back, as in the reference wrapper
            d = c4[0] & 0xFFFFFFFF
            st.last_D = d
            _learn_depth24(st, key, c4[1], d)
            if c4[1] & 8:                                      # the three-pass depth sort does not cover this call's depths
                if depth_bound is not None:
                    depth_bound.clear()
                continue
            over = (c4[1] & 1) != 0
            need = max(d, c4[2] & 0xFFFFFFFF) if c.split else d   # split: the capacity that gives each half a large enough share
            if over:
                if max_instances is not None:
                    raise GhOverflowError(f"tile instances D={d} exceed max_instances={cap}")
                st.capacity[key] = int(need * 1.5) + 1024
                if depth_bound is not None:
                    depth_bound.clear()                        # (the truncated lists' report is not a bound)
                continue                                       # (the too-small workspace is simply dropped)
            if c4[1] & 4:                                      # the occlusion bound missed: the same call again, unbounded
                depth_bound.clear()
                depth_bound.misses += 1
                continue
            if max_instances is None and key not in st.capacity:
                st.capacity[key] = max(int(need * 1.5) + 1024, 1 << 16)
        elif _policy.graph_mode:
            # every captured workspace is registered (keyed by its address, so a workspace re-used by later captures is
            # listed once); check_overflow() reads each of them
            st.graph_counters[ws.data_ptr()] = (counters, cap, key, True)
        else:
            if len(st.pending) >= _PENDING_MAX:
                check_overflow(block=False, dev=st.index)
                if len(st.pending) >= _PENDING_MAX:
                    check_overflow(block=True, dev=st.index)
            pc = _queue_readback(st, counters, cap, key, depth_bound)
            verdict = pc
            if auto:                                       # an explicit sync=False never blocks: check_overflow() is the caller's job
                pending = pc
        break
    # verdict: the counter read-back of a sync-free call, whoever is to ask for it
    ctx = _make_ctx(c._replace(flags=flags), dims, inp, ws, stream, alpha, None, radii, pending, verdict, False, l1, fit,
                    bool(flags & _abi.GH_FLAG_DEFER_LOSS_SUM))
    return image, radii, ctx


def raster_forward(cams, means3D, opacities, scales, rotations, *, H: int, W: int, shs=None, colors_precomp=None,
                   sh_degree: int = 0, scale_modifier: float = 1.0, xyz_b=None, opacity_b=None, color_w=None,
                   color_b=None, max_instances: Optional[int] = None, sync: Optional[bool] = True, return_alpha: bool = False,
                   per_view_gaussians: bool = False, geometry_of: Optional["_Ctx"] = None,
                   split_streams: Optional[bool] = None, expect_backward: bool = False, static_lists: bool = False,
                   refresh_of: Optional["_Ctx"] = None, depth_bound: Optional[DepthBoundCache] = None, cov3D_precomp=None,
                   l1_target: Optional[torch.Tensor] = None, fit_loss=None, defer_loss: bool = False):
    """Low-level forward through the C-ABI. Returns (image (NV,3,H,W), radii (NV,P) int32, ctx);
    cov3D_precomp (P,6): the published module's precomputed 3-D covariance (xx xy xz yy yz zz, used as given: scale_modifier is
    not applied) in place of scales + rotations (pass None for both); its gradient comes back as "cov3D_precomp".
    with return_alpha the fused mask channel (NV,H,W) is available as ctx.alpha.
    per_view_gaussians (pose batch, the batch loop of GS3DRenderer.forward): every per-Gaussian tensor holds NV*P rows and
    view v renders rows [v*P, (v+1)*P) — NV different Gaussian sets in one launch sequence.
    geometry_of: context of an earlier forward with the SAME means3D / opacities / scales / rotations / cameras (apart from
    bg): this call re-uses its projection and tile lists (gh_forward_shared) and only walks the lists with its own colours —
    the reference's mask pass after the RGB pass of a view. Colours must be colors_precomp.
    split_streams (GH_FLAG_SPLIT_STREAMS, n_views >= 2): the views run as two halves on two HIP streams inside the library
    (forked from / joined into the current stream, graph-capturable); bit-identical images, radii and gradients.
    None = the module policy (set_split_streams).
    static_lists (GH_FLAG_STATIC_LISTS): build tile lists that stay valid for other opacities / colours (culling treats
    every opacity o as max(2, 2 o)); same image and gradients, larger D. refresh_of: context of such a forward with the SAME
    means3D / scales / rotations / xyz_b / cameras: this call (gh_forward_refresh) skips projection and sorts, refreshes the
    per-instance records with ITS opacities and colours (shs or colors_precomp) and walks the lists; forward bit-identical
    to a full call. An opacity above the lists' bound poisons the call (NaN image, GhStaleGeometryError). GeometryCache is
    the policy object on top of the two.
    depth_bound: a DepthBoundCache (full forwards only): instances behind the previous call's per-tile occlusion depth are
    not listed; verified by the forward, re-run without the bound on a miss.
    l1_target (n_views,3,H,W): fused image loss — ctx.l1 = (mean|image - l1_target|, its gradient w.r.t. image) from the render
    kernel's own epilogue (GhOutputs.l1_*). ctx.l1 is None where the library does not fuse it (alpha, a depth bound, two
    streams, shared calls): the caller then runs gh_l1_loss on the image (loss.py does).
    fit_loss (gt_rgb (n_views,H,W,3), gt_mask (n_views,H,W), bbox (n_views,H,W) or None, lambda_l1, lambda_mask, scale), with
    return_alpha: the fit's image loss the same way — ctx.fit = (loss, dL/dimage, dL/dalpha) (GhOutputs.fit_loss), None where
    it is not fused (loss.py then runs gh_fit_loss).
    defer_loss (GH_FLAG_DEFER_LOSS_SUM, with a fused loss only): the loss tensor of ctx.l1 / ctx.fit is written by this context's
    raster_backward (GhGrads.deferred_loss) instead of a one-workgroup sum kernel behind the forward; ctx.defer_loss says so.
    sync: True = read D back (and re-run with a larger capacity if needed); False = never block (check_overflow() is the
    caller's job); None = auto: read D back for the first call of a shape and for calls whose backward will not come
    (expect_backward False), otherwise sync-free with the check at the start of raster_backward.
    The three-pass depth sort (GH_FLAG_DEPTH24) is used only for call shapes a read-back has shown it to hold for: the first
    call of a shape, and every call of a loop that never reads its counters back, run the four-pass sort."""
    L = _lib.lib()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("guassianhand_amd rasteriser needs tensors on a ROCm device (no CPU fallback)")
    plain = geometry_of is None and refresh_of is None and not static_lists
    c = _prepare_call(dev, cams, means3D, opacities, scales, rotations, H, W, shs, colors_precomp, sh_degree, scale_modifier, xyz_b,
                      opacity_b, color_w, color_b, per_view_gaussians, split_streams, plain, cov3D_precomp)
    st = _state(dev)
    with st.lock:
        if geometry_of is not None:
            return _forward_shared(st, L, dev, c, geometry_of, return_alpha)
        if refresh_of is not None:
            return _forward_refresh(st, L, dev, c._replace(flags=c.flags | _abi.GH_FLAG_STATIC_LISTS), refresh_of, return_alpha, sync,
                                    expect_backward, l1_target, fit_loss, defer_loss)
        return _forward_full(st, L, dev, c, return_alpha, sync, expect_backward, max_instances, static_lists, depth_bound, l1_target,
                             fit_loss, defer_loss)


def cached_raster_forward(cache: Optional[GeometryCache], cams, means3D, opacities, scales, rotations, **kw):
    """raster_forward through a GeometryCache (None: a plain call). A hit re-uses the cached static tile lists
    (gh_forward_refresh); a miss is a full forward with static_lists=True whose context becomes the cache entry. With
    sync=True a refresh that finds an opacity above the lists' bound is re-run as a full call, transparently."""
    if cache is None:
        return raster_forward(cams, means3D, opacities, scales, rotations, **kw)
    if isinstance(cache, DepthBoundCache):         # moving geometry: a speculative occlusion bound instead of static lists
        return raster_forward(cams, means3D, opacities, scales, rotations, depth_bound=cache, **kw)
    if kw.get("geometry_of") is not None or kw.get("split_streams"):
        raise ValueError("a GeometryCache call takes neither geometry_of nor split_streams")
    xyz_b = kw.get("xyz_b")
    objs = (cams, means3D, scales, rotations, xyz_b)
    shs = kw.get("shs")
    vals = tuple(None if o is None else (o._version, tuple(o.shape), o.device) for o in objs) + \
        (int(kw["H"]), int(kw["W"]), float(kw.get("scale_modifier", 1.0)), bool(kw.get("per_view_gaussians", False)),
         # the colour mode: the workspace layout behind the static part depends on M (one cache used with
         # use_rgb=True and use_rgb=False is two entries, i.e. a miss)
         0 if shs is None else int(shs.shape[1]), int(kw.get("sh_degree", 0)))
    g0 = cache.lookup(objs, vals)
    if g0 is not None:
        try:
            out = raster_forward(cams, means3D, opacities, scales, rotations, refresh_of=g0,
                                 **{k: v for k, v in kw.items() if k != "max_instances"})
            cache.hits += 1
            return out
        except GhStaleGeometryError:               # (sync=True only) the caches are cleared: rebuild below
            pass
    image, radii, ctx = raster_forward(cams, means3D, opacities, scales, rotations, static_lists=True, split_streams=False, **kw)
    cache.store(objs, vals, ctx)
    cache.builds += 1
    return image, radii, ctx


def raster_backward(ctx: _Ctx, dL_dimage: Optional[torch.Tensor], want_means2D: bool = True,
                    dL_dalpha: Optional[torch.Tensor] = None, grad_scale: Optional[torch.Tensor] = None,
                    want=None) -> Dict[str, torch.Tensor]:
    """grad_scale: optional one-element device tensor multiplied into dL_dimage / dL_dalpha as the kernel reads them
    (GhGrads.upstream_scale): the dL/dloss of a scalar loss whose image gradient was produced unscaled.
    want: the gradients to produce, a subset of {means3D, opacities, scales, rotations, shs, colors_precomp, xyz_b, opacity_b,
    color_w, color_b} (None = all): the others get a NULL pointer in GhGrads — with no geometry gradient asked for the
    per-Gaussian chain rule is skipped altogether (the one-shot fit trains colour / opacity biases only)."""
    L = _lib.lib()
    t = ctx.tensors
    dev = t["means3D"].device
    st = _state(dev)
    if ctx.pending is not None and ctx.pending.resolve():
        # auto-sync forward (the drop-in's default): its counters are checked HERE, before any gradient exists — an overflowed
        # call returned a NaN image, and a NaN loss must not reach the caller's optimiser step. The forward finished long ago
        # on any host-bound loop, so this wait is normally free.
        with st.lock:
            st.geom_last = None                    # the re-run of the step must not be taken for this call's mask pass
        raise ctx.pending.error()
    P, NV, M = ctx.P, ctx.NV, ctx.M
    if dL_dimage is None:
        dL_dimage = torch.zeros(NV, 3, ctx.H, ctx.W, dtype=torch.float32, device=dev)
    g = dL_dimage.detach().to(torch.float32).reshape(NV, 3, ctx.H, ctx.W).contiguous()
    ga = None if dL_dalpha is None else dL_dalpha.detach().to(torch.float32).reshape(NV, ctx.H, ctx.W).contiguous()
    gs = None if grad_scale is None else grad_scale.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()
    R_ = ctx.rows                                  # rows of the per-Gaussian tensors (P, or NV*P for a pose batch)
    # ONE contiguous fp32 block for every gradient the kernels write (the C-ABI takes plain pointers, so they may all point
    # into one allocation): a single torch.empty per backward, and the sharded fit all-reduces `block` as it stands —
    # [4 caller floats | color_w | opacity_b | color_b | xyz_b | means3D | opacities | scales | rotations | colour | (means2D)]:
    # the blend-parameter gradients (what the one-shot fit trains) lead, so the fit reduces a short prefix.
    # Every part starts on a 16-byte boundary (the 48-wide rows are written as float4s).
    shapes = [("color_w", ((R_, 48) if ctx.wpg else (48,)) if t["color_w"] is not None else None),
              ("opacity_b", (R_,) if t["opacity_b"] is not None else None),
              ("color_b", (R_, 3 if ctx.b_rgb else 48) if t["color_b"] is not None else None),
              ("xyz_b", (3,) if t["xyz_b"] is not None else None),
              ("means3D", (R_, 3)), ("opacities", (R_,)), ("scales", (R_, 3) if t["scales"] is not None else None),
              ("rotations", (R_, 4) if t["rotations"] is not None else None), ("cov3D_precomp", (R_, 6) if t["cov3D"] is not None else None),
              ("shs", (R_, M, 3) if M else None), ("colors_precomp", (R_, 3) if t["colors_precomp"] is not None else None),
              ("means2D", (NV, P, 3) if want_means2D else None)]
    off, spans = 4, []                             # floats 0..3: reserved for the caller (e.g. the loss of the sharded fit)
    for k, shp in shapes:
        if shp is None or (want is not None and k != "means2D" and k not in want):
            continue
        n = 1
        for d_ in shp:
            n *= d_
        spans.append((k, shp, off, n))
        off += (n + 3) & ~3
    block = torch.empty(off, dtype=torch.float32, device=dev)
    o = {k: None for k, _ in shapes}
    for k, shp, a, n in spans:
        o[k] = block[a:a + n].view(shp)
    reducible = spans[-1][2] if (want_means2D and spans and spans[-1][0] == "means2D") else off   # means2D is per view: not part of the all-reduced prefix
    blend_end = 4
    for k, shp, a, n in spans:
        if k in ("color_w", "opacity_b", "color_b", "xyz_b"):
            blend_end = a + ((n + 3) & ~3)
    ctx_block = (block, reducible, [(k, shp, a, n) for k, shp, a, n in spans if k != "means2D"], blend_end)
    gr = _abi.GhGrads(dL_dimage=_ptr(g), dL_dalpha=_ptr(ga), dL_dmeans3D=_ptr(o["means3D"]), dL_dmeans2D=_ptr(o["means2D"]),
                      dL_dopacities=_ptr(o["opacities"]), dL_dscales=_ptr(o["scales"]), dL_drotations=_ptr(o["rotations"]),
                      dL_dshs=_ptr(o["shs"]), dL_dcolors=_ptr(o["colors_precomp"]), dL_dblend_xyz_b=_ptr(o["xyz_b"]),
                      dL_dblend_opacity_b=_ptr(o["opacity_b"]), dL_dblend_color_w=_ptr(o["color_w"]),
                      dL_dblend_color_b=_ptr(o["color_b"]), upstream_scale=_ptr(gs), dL_dcov3D=_ptr(o["cov3D_precomp"]),
                      deferred_loss=_ptr((ctx.l1 or ctx.fit)[0]) if getattr(ctx, "defer_loss", False) else None)
    stream = _raw_stream(dev)
    ctx.stream = stream                            # the workspace goes back to the pool of the stream that used it last
    with _OnDevice(dev):
        bargs = (C.byref(ctx.dims), C.byref(ctx.inp), C.byref(gr), C.c_void_p(ctx.ws.data_ptr()), ctx.ws.numel(),
                 C.c_void_p(stream))
        if ctx.parent is not None:
            fn = L.gh_backward_refresh if ctx.refresh else L.gh_backward_shared
            rc = fn(C.byref(ctx.dims), C.byref(ctx.inp), C.byref(gr), C.c_void_p(ctx.parent.ws.data_ptr()),
                    C.c_void_p(ctx.ws.data_ptr()), ctx.ws.numel(), C.c_void_p(stream))
        elif _policy.stage_timing:
            rc = _run_stages(st, L.gh_backward_stages, bargs, (("render_bwd", _abi.GH_BWD_RENDER),
                                                            ("preprocess_bwd", _abi.GH_BWD_PREPROCESS)))
        else:
            rc = L.gh_backward(*bargs)
    if rc != 0:
        raise RuntimeError(f"gh_backward failed: {_abi.status_name(rc)}")
    st.last_grad_block = ctx_block
    return {k: v for k, v in o.items() if v is not None}


def last_grad_block(dev=None):
    """(block, n_reducible, spans, blend_end) of the most recent raster_backward: `block[:n_reducible]` is the contiguous fp32
    buffer [4 caller floats | every view-summed gradient] the kernels wrote in place, spans = [(name, shape, offset, numel)],
    block[:blend_end] = [4 caller floats | gradients of the blend parameters] (the prefix the one-shot fit reduces).
    The sharded fit all-reduces it directly (dist.allreduce_block): no torch.cat of the parts."""
    return _state(dev).last_grad_block


def workspace_counters(ctx: _Ctx):
    """GhCounters of a forward as a host list [num_rendered, overflow, reserved0, reserved1] (synchronises; tests / tools)."""
    return [c & 0xFFFFFFFF for c in ctx.ws[:16].view(torch.int32).tolist()]


def workspace_views(ctx: _Ctx) -> Dict[str, torch.Tensor]:
    """Typed views of the internal stage arrays (for tests / profiling), via gh_workspace_layout."""
    L = _lib.lib()
    lay = _abi.GhLayout()
    L.gh_workspace_layout(C.byref(ctx.dims), C.byref(lay))
    if ctx.dims.flags & _abi.GH_FLAG_SPLIT_STREAMS:
        raise ValueError("workspace_views: a split call keeps two sets of per-instance arrays; render with split_streams=False")
    ws, N, cap = ctx.ws, ctx.NV * ctx.P, int(ctx.dims.max_instances)
    gx, gy = (ctx.W + 15) // 16, (ctx.H + 15) // 16
    pix = ctx.NV * ctx.H * ctx.W

    def v(off, nbytes, dtype, *shape):
        return ws[off:off + nbytes].view(dtype).reshape(*shape)

    geom = v(lay.geom, N * 64, torch.float32, N, 16)
    sorted_tile = v(lay.keys_a, cap * 4, torch.int32, cap)
    if L.gh_partition_is_per_view(C.byref(ctx.dims)) == 1:      # the keys hold tile ids inside the view: the global id through the payload's view
        sorted_tile = sorted_tile + (v(lay.vals_a, cap * 4, torch.int32, cap).clamp(0, max(N - 1, 0)) // max(ctx.P, 1)) * (gx * gy)
    return dict(counters=v(lay.counters, 16, torch.int32, 4), g0=geom[:, 0:4], g1=geom[:, 4:8], gb=geom[:, 8],
                # (view-space depth and 3-sigma tile rect of every projected Gaussian: the geometry line's last float4, v0.8)
                depth=geom[:, 13], rect=geom[:, 14].contiguous().view(torch.int32),
                tiles_touched=v(lay.tiles_touched, N * 4, torch.int32, N), slot_begin=v(lay.slot_begin, N * 4, torch.int32, N),
                depth_order=v(lay.depth_vals_b if (ctx.dims.flags & _abi.GH_FLAG_DEPTH24) else lay.depth_vals_a, N * 4, torch.int32, N),
                sorted_tile=sorted_tile, sorted_slot=v(lay.sorted_slot, cap * 4, torch.int32, cap),
                sorted_gid=v(lay.vals_a, cap * 4, torch.int32, cap),
                inst_r2=v(lay.inst_r2, cap * 8, torch.int32, cap, 2),
                ranges=v(lay.ranges, ctx.NV * gx * gy * 8, torch.int32, ctx.NV * gx * gy, 2),
                final_T=v(lay.final_T, pix * 4, torch.float32, ctx.NV, ctx.H, ctx.W),
                n_contrib=v(lay.n_contrib, pix * 4, torch.int32, ctx.NV, ctx.H, ctx.W))


# ---------------------------------------------------------------------------------------------------
def set_geometry_reuse(on: bool) -> None:
    """The reference renders every view twice over identical geometry (RGB pass, then mask pass with colour 1 / bg 0,
    renderer_one_shot.py:338-346, :372-379). With reuse on (default) a drop-in call that receives the very same tensor
    objects (unmodified) for means3D / opacities / scales / rotations and the same camera as the previous call, with
    precomputed colours, shares the previous call's projection and tile lists (bit-identical results)."""
    _policy.reuse_geometry = bool(on)
    for st in list(_states.values()):
        with st.lock:
            st.geom_last = None


def _geometry_key(means3D, opacities, scales, rotations, rs):
    objs = (means3D, opacities, scales, rotations, rs.viewmatrix, rs.projmatrix, rs.campos)
    vals = tuple(o._version for o in objs) + (int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                                               float(rs.scale_modifier), _raw_stream(means3D.device))
    return objs, vals


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, raster_settings, sync, expect_backward,
                cov3D_precomp=None):
        rs = raster_settings
        cams = pack_camera(rs.viewmatrix, rs.projmatrix, rs.campos, rs.tanfovx, rs.tanfovy, rs.bg)
        parent = None
        st = _state(means3D.device)
        if _policy.reuse_geometry:
            # Only the documented pair is shared: the call DIRECTLY after a full call, with the very same (unmodified, `is` +
            # `_version`) geometry tensors and camera on the same stream, precomputed colours, while the first call's context is
            # still alive (its backward has not run). The record holds weak references and is dropped after one reuse, so
            # nothing of an earlier step can be picked up. (An in-place update through `.data` does not move `_version`:
            # between an RGB call and its mask call nothing updates parameters.)
            # (shape of the Gaussians: scales + rotations, or the precomputed covariance in both places)
            objs, vals = _geometry_key(means3D, opacities, scales if cov3D_precomp is None else cov3D_precomp,
                                       rotations if cov3D_precomp is None else cov3D_precomp, rs)
            with st.lock:
                g, st.geom_last = st.geom_last, None
            if g is not None and sh is None and g[1] == vals and all(a() is b for a, b in zip(g[0], objs)):
                parent = g[2]()                    # alive until its backward has run
                # never the lists of a call that overflowed: its verdict is known when its counters were read already (the caller
                # has been told, and this IS the re-run the message asks for — found by tools/fuzz_dropin.py --shrink), and a call
                # that reads D back itself (sync=True, or an inference call) asks now
                p = None if parent is None else parent.verdict
                if p is not None and (p.done or sync is True or (sync is None and not expect_backward)) and p.resolve():
                    parent = None
        image, radii, rctx = raster_forward(
            cams, means3D, opacities, scales, rotations, H=int(rs.image_height), W=int(rs.image_width),
            shs=sh, colors_precomp=colors_precomp, sh_degree=int(rs.sh_degree),
            scale_modifier=float(rs.scale_modifier), sync=sync, geometry_of=parent,
            expect_backward=expect_backward, cov3D_precomp=cov3D_precomp)
        if _policy.reuse_geometry and parent is None:
            with st.lock:
                st.geom_last = (tuple(weakref.ref(o) for o in objs), vals, weakref.ref(rctx))
        ctx.rctx = rctx
        # the backward kernels read these tensors again, through the pointers the context holds: registering them makes autograd
        # raise its usual "modified by an inplace operation" error when one of them was written between forward and backward,
        # as it does for the reference extension (which saves them) — instead of gradients of a scene that never existed
        ctx.save_for_backward(*[t for t in (means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp) if t is not None])
        ctx.shapes = (means3D.shape, means2D.shape, None if sh is None else sh.shape,
                      None if colors_precomp is None else colors_precomp.shape, opacities.shape,
                      None if scales is None else scales.shape, None if rotations is None else rotations.shape,
                      None if cov3D_precomp is None else cov3D_precomp.shape)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return image[0], radii[0]

    @staticmethod
    def backward(ctx, grad_image, _grad_radii):
        ctx.saved_tensors                                  # (version check of the inputs, see forward)
        g = raster_backward(ctx.rctx, grad_image)
        ctx.rctx = None
        s = ctx.shapes
        return (g["means3D"].reshape(s[0]), g["means2D"][0].reshape(s[1]),
                g["shs"].reshape(s[2]) if s[2] is not None else None,
                g["colors_precomp"].reshape(s[3]) if s[3] is not None else None,
                g["opacities"].reshape(s[4]), g["scales"].reshape(s[5]) if s[5] is not None else None,
                g["rotations"].reshape(s[6]) if s[6] is not None else None, None, None, None,
                g["cov3D_precomp"].reshape(s[7]) if s[7] is not None else None)


def _any_global_hook() -> bool:
    m = torch.nn.modules.module
    f = getattr(m, "_has_any_global_hook", None)
    if f is not None:
        return bool(f())
    return bool(getattr(m, "_global_forward_hooks", None) or getattr(m, "_global_forward_pre_hooks", None) or
                getattr(m, "_global_backward_hooks", None) or getattr(m, "_global_backward_pre_hooks", None))


class GaussianRasterizer(nn.Module):
    """Same constructor / call keywords / 2-tuple return as the module the reference imports at
    tgs/models/renderer_one_shot.py:3 and calls at :338-346 and :372-379."""

    def __init__(self, raster_settings: GaussianRasterizationSettings, sync: Optional[bool] = None):
        """sync=None (default): the first call of an image / Gaussian-count shape reads the instance count D back once to size
        the workspace capacity (1.5 D + 1024); later calls under autograd are sync-free in the forward, and their BACKWARD first
        checks the forward's counters (an event that completed long before on a host-bound loop): an overflowed call — which
        returned a NaN image, device-side guard — raises GhOverflowError there, before a gradient exists, with the capacity
        already raised; the caller's optimiser never sees the NaN. Calls outside autograd (inference: no backward will come)
        read D back like the reference wrapper and re-run transparently with a larger capacity.
        sync=True reads D back on every call like the reference wrapper does; sync=False never blocks (check_overflow())."""
        super().__init__()
        self.raster_settings = raster_settings
        self.sync = sync

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        # (cov3D_precomp: the reference never passes it, renderer_one_shot.py:313, :346; supported since round 5 — the covariance is
        # used as given, scale_modifier is not applied, and its gradient flows back like the published module's)
        if means3D.shape[0] == 0:
            # No Gaussians. The published wrapper skips the kernels then (`if(P != 0)` around the rasteriser call in the extension's
            # RasterizeGaussiansCUDA — third-party source, not in the reference tree) and returns the image it allocated: ZEROS, not the
            # background. The drop-in returns what the published module returns; the C-ABI / raster_forward composite the background
            # over nothing (T = 1), which is what App. A's formulas give. Connected to the graph so that a backward yields empty gradients.
            rs = self.raster_settings
            img = torch.zeros(3, int(rs.image_height), int(rs.image_width), dtype=torch.float32, device=means3D.device)
            return img + 0.0 * means3D.sum(), torch.zeros(0, dtype=torch.int32, device=means3D.device)
        # will a backward come? (decided here: inside an autograd Function's forward the grad mode is always off, and
        # needs_input_grad ignores torch.no_grad())
        expect_backward = torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp))
        # (The reference wraps the call in torch.autocast(dtype=float32), renderer_one_shot.py:337. Nothing below is an autocast-
        # eligible torch op — tensors go to the C-ABI as raw pointers after an explicit float32 check in _prep — so no
        # autocast(enabled=False) context is entered here: it cost 6 us per call for nothing.)
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                         self.raster_settings, self.sync, expect_backward, cov3D_precomp)

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """The published module's frustum test (its `mark_visible`: a point is visible when its view-space depth exceeds 0.2 — the
        near cull of App. A.1; the reference never calls it). (P,) bool, no gradient. A handful of torch ops on the caller's device:
        not part of the render path."""
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix.to(positions.dtype)      # row-vector convention: p_view = [x y z 1] @ viewmatrix
            return (positions[:, 0] * vm[0, 2] + positions[:, 1] * vm[1, 2] + positions[:, 2] * vm[2, 2] + vm[3, 2]) > 0.2

    # nn.Module.__call__ runs the hook machinery around forward(); this module never has hooks worth 3 us per call on a path
    # that is called twice per view (hooks registered by a caller are honoured: fall back to the full protocol then)
    def __call__(self, *args, **kwargs):
        if self._forward_hooks or self._forward_pre_hooks or self._backward_hooks or self._backward_pre_hooks or _any_global_hook():
            return super().__call__(*args, **kwargs)
        return self.forward(*args, **kwargs)


# ---------------------------------------------------------------------------------------------------
_GRAD_NAMES = ("means3D", "opacities", "scales", "rotations", "colour", "xyz_b", "opacity_b", "color_w", "color_b")


def _wanted(needs, use_rgb: bool):
    """Names of the gradients autograd asks for, from the needs_input_grad flags of the nine tensor arguments."""
    w = {n for n, need in zip(_GRAD_NAMES, needs) if need}
    if "colour" in w:
        w.discard("colour")
        w.add("colors_precomp" if use_rgb else "shs")
    return w


class _RasterizeViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cache, cams, H, W, sh_degree, scale_modifier, use_rgb, sync, max_instances, want_alpha, per_view, xyz, opacity,
                scaling, rotation, shs, xyz_b, opacity_b, color_w, color_b):
        kw = dict(colors_precomp=shs.reshape(shs.shape[0], 3)) if use_rgb else dict(shs=shs)
        image, radii, rctx = cached_raster_forward(cache, cams, xyz, opacity, scaling, rotation, H=H, W=W, sh_degree=sh_degree,
                                                   scale_modifier=scale_modifier, xyz_b=xyz_b, opacity_b=opacity_b,
                                                   color_w=color_w, color_b=color_b, sync=sync, max_instances=max_instances,
                                                   return_alpha=want_alpha, per_view_gaussians=per_view, **kw)
        ctx.rctx = rctx
        ctx.use_rgb = use_rgb
        ctx.save_for_backward(*[t for t in (cams, xyz, opacity, scaling, rotation, shs, xyz_b, opacity_b, color_w, color_b) if t is not None])
        ctx.shapes = [None if t is None else t.shape for t in (xyz, opacity, scaling, rotation, shs, xyz_b, opacity_b, color_w, color_b)]
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)             # an unused output (image or alpha) arrives as None, not as a zero image
        alpha = rctx.alpha if want_alpha else image.new_zeros(0)
        if not want_alpha:
            ctx.mark_non_differentiable(alpha)
        return image, alpha, radii

    @staticmethod
    def backward(ctx, grad_image, grad_alpha, _gr):
        ctx.saved_tensors                                  # inputs written in place since the forward: autograd's own error (see _RasterizeGaussians)
        if ctx.rctx.alpha is None:
            grad_alpha = None
        g = raster_backward(ctx.rctx, grad_image, want_means2D=False, dL_dalpha=grad_alpha,
                            want=_wanted(ctx.needs_input_grad[11:20], ctx.use_rgb))
        ctx.rctx = None
        s = ctx.shapes
        col = g.get("colors_precomp" if ctx.use_rgb else "shs")
        opt = lambda k, i: g[k].reshape(s[i]) if (s[i] is not None and k in g) else None
        return (None,) * 11 + (opt("means3D", 0), opt("opacities", 1), opt("scales", 2), opt("rotations", 3),
                              None if col is None else col.reshape(s[4]), opt("xyz_b", 5), opt("opacity_b", 6),
                              opt("color_w", 7), opt("color_b", 8))


def rasterize_views(cams: torch.Tensor, xyz, opacity, scaling, rotation, shs, *, H: int, W: int, use_rgb: bool,
                    sh_degree: int = 3, scale_modifier: float = 1.0, xyz_b=None, opacity_b=None, color_w=None,
                    color_b=None, sync: bool = True, max_instances: Optional[int] = None, return_alpha: bool = False,
                    per_view_gaussians: bool = False, geometry_cache: Optional[GeometryCache] = None,
                    depth_bound: Optional[DepthBoundCache] = None):
    """View-batched render with the attribute blend of renderer_one_shot.py:298-334 fused into the kernels.
    per_view_gaussians=True renders a POSE BATCH: the Gaussian tensors hold Nv*P rows and camera v sees rows
    [v*P, (v+1)*P) only — the batch loop of GS3DRenderer.forward (renderer_one_shot.py:615-633) in one launch sequence.

    cams: (Nv, GH_CAM_FLOATS) from camera.pack_cameras_from_w2c; returns (images (Nv,3,H,W), radii (Nv,P)) or, with
    return_alpha, (images, alpha (Nv,H,W), radii): alpha is the reference's mask render (colour 1, bg 0,
    renderer_one_shot.py:353-380) produced by the same pass as a 4th channel (bit-identical to a separate pass).
    Differentiable w.r.t. xyz, opacity, scaling, rotation, shs and the blend parameters (through image and alpha); only
    the gradients autograd needs are computed. geometry_cache: see GeometryCache (static Gaussians and cameras);
    depth_bound: see DepthBoundCache (Gaussians that move a little between steps) — one or the other.
    """
    if depth_bound is not None:
        if geometry_cache is not None:
            raise ValueError("rasterize_views: geometry_cache (static geometry) or depth_bound (moving geometry), not both")
        geometry_cache = depth_bound
    image, alpha, radii = _RasterizeViews.apply(geometry_cache, cams, int(H), int(W), int(sh_degree if not use_rgb else 0), float(scale_modifier),
                                 bool(use_rgb), bool(sync), max_instances, bool(return_alpha), bool(per_view_gaussians), xyz, opacity,
                                 scaling,
                                 rotation, shs, xyz_b, opacity_b, color_w, color_b)
    return (image, alpha, radii) if return_alpha else (image, radii)


# ---------------------------------------------------------------------------------------------------
# Names of the module-level state of rounds 1-4, kept readable for tests and tools: they resolve to the CURRENT device's state /
# the policy object (PEP 562). Code in this package uses _state(dev) / _policy directly.
_LEGACY_STATE = {"_capacity": "capacity", "_depth24": "depth24", "_pending": "pending", "_free_slots": "free_slots", "_last_D": "last_D",
                 "_last_ws": "last_ws", "_graph_counters": "graph_counters", "_ws_pool": "ws_pool", "_geom_last": "geom_last",
                 "_last_grad_block": "last_grad_block", "_stage_events": "stage_events"}
_LEGACY_POLICY = {"_graph_mode": "graph_mode", "_split_policy": "split", "_reuse_geometry": "reuse_geometry", "_stage_timing": "stage_timing"}


def __getattr__(name):
    if name in _LEGACY_STATE:
        return getattr(_state(), _LEGACY_STATE[name])
    if name in _LEGACY_POLICY:
        return getattr(_policy, _LEGACY_POLICY[name])
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
