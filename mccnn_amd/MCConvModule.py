"""Python op surface of the Monte-Carlo convolution layer.

Same function names, argument orders and gradient wiring as the reference's
tf_ops/MCConvModuleSrc (generated into MCConvModule.py, genCompileScript.py:40-48), with
torch CUDA tensors instead of TF tensors. Every op calls the C-ABI of include/mccnn.h
(libmccnn_hip.so, hand-written HIP for gfx950) on the current torch stream; there is no
CPU or PyTorch fallback -- a missing library raises.

Shape / attribute validation mirrors the OP_REQUIRES blocks of the reference wrappers and
raises InvalidArgumentError (the analogue of tf.errors.InvalidArgumentError).
"""
import ctypes as C
import os
import threading
import time
import weakref

import numpy as _np
import torch

from . import _env, _lib
from ._lib import check, ptr, stream_handle


class InvalidArgumentError(ValueError):
    pass


def _req(cond, msg):
    if not cond:
        raise InvalidArgumentError(msg)


def _f32(t, name):
    _req(isinstance(t, torch.Tensor) and t.is_cuda, "%s must be a CUDA tensor" % name)
    _req(t.dtype == torch.float32, "%s must be float32" % name)
    return t.contiguous()


def _feat(t, name):
    """Feature rows: float32, or bfloat16 storage (extension, BASELINE cfg3: depth-wise layers gather half the bytes).
    bf16 rows need an even number of features -- the row permutations move them as 32-bit words."""
    _req(isinstance(t, torch.Tensor) and t.is_cuda, "%s must be a CUDA tensor" % name)
    _req(t.dtype in (torch.float32, torch.bfloat16), "%s must be float32 (or bfloat16 storage)" % name)
    if t.dtype == torch.bfloat16:
        _req(t.dim() == 2 and t.shape[1] % 2 == 0, "%s: bfloat16 rows need an even number of features" % name)
    return t.contiguous()


def _rows32(t):
    """A bf16 row tensor seen as rows of 32-bit words (no copy); f32 tensors pass through."""
    return t.view(torch.float32) if t.dtype == torch.bfloat16 else t


def _like_rows(words, ref):
    """Undo _rows32 for a result tensor."""
    return words.view(torch.bfloat16) if ref.dtype == torch.bfloat16 else words


def _i32(t, name):
    _req(isinstance(t, torch.Tensor) and t.is_cuda, "%s must be a CUDA tensor" % name)
    _req(t.dtype == torch.int32, "%s must be int32" % name)
    return t.contiguous()


_WS_POOL = {}  # (thread, device, raw stream) -> grow-only scratch buffer


def _ws(nbytes, device):
    """Scratch memory for ONE library call (or one count / fill pair issued back to back): a grow-only buffer per host
    thread and stream instead of an allocation per call (~120 per step of a segmentation network). Consecutive calls
    on a stream may share it because they are stream-ordered; nothing that has to outlive the call sequence lives
    here (lists, plans and the forward state are tensors of their own)."""
    nbytes = max(int(nbytes), 256)
    key = (threading.get_ident(), device.index, stream_handle())
    buf = _WS_POOL.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS_POOL[key] = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=device)
    return buf


def _check_points(pts, name, op):
    _req(pts.dim() == 2, "%s expects %s with the following dimensions (numPoints, pointComponents)" % (op, name))
    _req(pts.shape[1] == 3, "%s expects %s with three components" % (op, name))


def _check_batch_ids(bids, n, op):
    _req(bids.dim() == 2 and bids.shape[0] == n and bids.shape[1] == 1,
         "%s expects as batch ids input the following dimensions (numPoints, 1)" % op)


def _check_aabb(mn, mx, batchSize, op):
    for t in (mn, mx):
        _req(t.dim() == 2 and t.shape[0] == batchSize and t.shape[1] == 3,
             "%s expects bounding box points with shape (batchSize, 3)" % op)


def get_block_size():
    """genCompileScript.py:46-47"""
    return int(_lib.load().mccnn_block_size())


# ---------------------------------------------------------------------------------------------
# Visiting-order hints for find_neighbors. The op surface of the reference has no such argument, so the shim
# remembers, per point tensor, a permutation that walks the points cell by cell: for the points a grid was built
# from it is the inverse of index_new_pos (Poisson samples get theirs from the native path's counting sort, csrc/exec.hip).
# A hint only changes the order in which GPU threads visit the centres (speed), never the result.
_ORDER_HINTS = {}


def _tensor_key(t):
    # storage address + version: autograd hands the op a different tensor OBJECT over the same storage, so object
    # identity cannot be used here. A recycled address can only yield a stale permutation of the same length,
    # which is still a valid visiting order (results do not depend on it).
    return (t.device.index, t.data_ptr(), t._version, t.shape[0])


def _evict_oldest(cache, limit):
    """Drop the OLDER half of a full hint cache (dicts keep insertion order): the hints of the step in flight are the
    youngest entries and stay."""
    if len(cache) > limit:
        for k in list(cache)[:len(cache) // 2]:
            del cache[k]


def _remember_order(points, kind, payload):
    _evict_oldest(_ORDER_HINTS, 64)
    key = _tensor_key(points)
    _ORDER_HINTS.pop(key, None)  # re-registered: moves to the young end
    _ORDER_HINTS[key] = [kind, payload, None]


def _order_hint(points):
    ent = _ORDER_HINTS.get(_tensor_key(points))
    if ent is None:
        return None
    if ent[0] == "order_weak":
        return ent[1]()  # None once the grid that owns the permutation is gone
    if ent[2] is None:
        kind, payload = ent[0], ent[1]
        if kind == "order":
            ent[2] = payload
        elif kind == "new_idx":
            inv = torch.empty_like(payload)
            check(_lib.load().mccnn_invert_permutation(ptr(payload), payload.shape[0], ptr(inv), stream_handle()),
                  "invert_permutation")
            ent[2] = inv
        else:
            return None
    return ent[2]


_EDGE_GUESS = {}  # (device, M, N, radius, B, scaleInv) -> capacity to try first in find_neighbors
# ... and, for batches whose sizes change from step to step (ragged training batches), the edges per centre of the last
# search with the same radius: (device, radius, scaleInv) -> E / M
_EDGE_RATIO = {}


def _edge_guess(gkey):
    g = _EDGE_GUESS.get(gkey, 0)
    if g <= 0:
        ratio = _EDGE_RATIO.get((gkey[0], gkey[3], gkey[5]), 0.0)
        if ratio > 0.0:
            g = int(ratio * gkey[1] * 1.25) + 1024  # sizes differ: more head room than for a repeated shape
    return g


def _remember_edges(gkey, e):
    if len(_EDGE_GUESS) > 256:
        _EDGE_GUESS.clear()
    _EDGE_GUESS[gkey] = e + e // 16 + 64  # a little head room: totals of a shape vary slightly from batch to batch
    if gkey[1] > 0:
        _EDGE_RATIO[(gkey[0], gkey[3], gkey[5])] = e / float(gkey[1])
_TLS = threading.local()
_MAILBOX_COPY = _env.debug("mailbox_copy", False)


def _pinned_int():
    """One pinned int32 per host thread: the copy into it and the wait for it happen back to back on the calling
    thread, so concurrent callers (threads / streams) never share a slot."""
    buf = getattr(_TLS, "pinned", None)
    if buf is None:
        buf = _TLS.pinned = torch.empty(1, dtype=torch.int32).pin_memory()
    return buf


def _host_mailbox():
    """(pinned int32 tensor, NumPy view of it) per host thread. Pinned host memory is device-accessible, so a kernel
    can store a count STRAIGHT into it: no device-to-host copy is enqueued and the host sees the value as soon as the
    producing kernel retires -- it polls the word instead of sleeping on an event (a blocking wait wakes up ~10 us
    late, which at a sub-millisecond step is idle GPU time)."""
    box = getattr(_TLS, "mailbox", None)
    if box is None:
        t = torch.empty(1, dtype=torch.int32).pin_memory()
        box = _TLS.mailbox = (t, t.numpy())
    return box


# find_neighbors: True = the count kernel stores the edge total into the pinned mailbox and the host polls it;
# False = total in device memory + an asynchronous copy + an event wait
COUNT_MAILBOX = _env.debug("count_mailbox", True)
_MAILBOX_SPIN_S = 200e-6  # tight polling for this long (the producing kernel retires within tens of microseconds) ...
_MAILBOX_YIELD_S = 0.25   # ... then polling that hands the GIL to other threads between reads, then a synchronisation


#: seconds this process has spent WAITING for sizes computed on the device (edge totals through the mailbox, the sample
#: counts of a hierarchy); diagnostics only (bench.py: host work per step = issue time - waits)
HOST_WAIT_S = [0.0]   # ... measured by this module (the op-by-op paths)
HOST_LAG_WAIT_S = [0.0]   # the part of HOST_WAIT_S spent in ConvolutionBuilder.reset() holding the host k steps behind the device


def host_wait_seconds():
    """Seconds the calling side has spent so far WAITING for device-side sizes or for a helper thread that waits for them:
    this module's own waits + the library's (mccnn_debug_wait_ns) + the extension's (wait_ns). A step's host time minus the
    increase of this number is the host's own work."""
    total = HOST_WAIT_S[0] + _lib.load().mccnn_debug_wait_ns() * 1e-9
    ext = _torch_ext()
    if ext is not None:
        total += ext.wait_ns() * 1e-9
    return total


def _await_mailbox(view):
    v = int(view[0])
    if v >= 0:
        return v
    t0 = time.perf_counter()
    try:
        return _await_mailbox_slow(view)
    finally:
        HOST_WAIT_S[0] += time.perf_counter() - t0


def _await_mailbox_slow(view):
    """Value a kernel of the current stream wrote to the mailbox (armed with -1 by the caller before the launch). The
    tight spin is bounded by TIME: a data-loader or autograd thread of the same process is not starved for longer than
    a fraction of a millisecond; after that every read is followed by time.sleep(0), which releases the GIL."""
    v = int(view[0])
    if v >= 0:
        return v
    t0 = time.perf_counter()
    while True:
        for _ in range(64):
            v = int(view[0])
            if v >= 0:
                return v
        dt = time.perf_counter() - t0
        if dt > _MAILBOX_SPIN_S:
            break
    while time.perf_counter() - t0 < _MAILBOX_YIELD_S:
        time.sleep(0)
        v = int(view[0])
        if v >= 0:
            return v
    torch.cuda.synchronize()  # every stream of the device: the producer may not be the calling stream
    return int(view[0])


_NUM_CELLS_CACHE = {}


def prefetch_transposed(packed, n, stream):
    """Builds the transposed neighbour list of `packed` on `stream` (a side stream) ahead of the backward pass that will
    ask for it; the consumer waits for the recorded event (see _transposed_neighbors). `packed` must be ready on
    `stream`."""
    with torch.cuda.stream(stream):
        if getattr(packed, "_mccnn_transposed", None) is None:
            _transposed_neighbors(packed, n)
            ev = torch.cuda.Event()
            ev.record(stream)
            try:
                packed._mccnn_transposed_event = ev
            except AttributeError:
                stream.synchronize()


def _transposed_neighbors(packed, n):
    """Transposed neighbour list (CSR by neighbour index) of `packed`, shared by every depth-wise layer that
    convolves over the same list -- the counterpart of ConvolutionBuilder's cacheNeighs_ for the backward pass. It is
    stored ON the neighbour-list tensor object, so it lives exactly as long as the list itself (the builder's cache
    entry) and there is no global table to go stale."""
    hit = getattr(packed, "_mccnn_transposed", None)
    if hit is not None and hit[2] == (packed._version, n):
        ev = getattr(packed, "_mccnn_transposed_event", None)
        if ev is not None:  # built ahead of time on another stream (prefetch_transposed): order this stream behind it
            torch.cuda.current_stream().wait_event(ev)
            for t in hit[:2]:   # allocated on that stream, read on this one
                t.record_stream(torch.cuda.current_stream())
            # the event stays with the list: a later consumer on another stream has to wait for it, too (waiting for an
            # event that has fired costs nothing on the queue)
        return hit
    lib = _lib.load()
    e = packed.shape[0]
    start_t = torch.empty(n + 1, dtype=torch.int32, device=packed.device)
    perm_t = torch.empty(max(e, 1), dtype=torch.int32, device=packed.device)
    ws = _ws(lib.mccnn_transpose_neighbors_workspace_bytes(n, e), packed.device)
    check(lib.mccnn_transpose_neighbors(ptr(packed), e, n, ptr(start_t), ptr(perm_t), ptr(ws), ws.numel(),
                                        stream_handle()), "transpose_neighbors")
    hit = (start_t, perm_t, (packed._version, n))
    try:
        packed._mccnn_transposed = hit
    except AttributeError:  # a tensor subclass without a __dict__: recompute next time
        pass
    return hit


#: depth-wise layers (numFeatures % 8 == 0) run the row-per-lane kernels over SELL layouts of the neighbour list
#: (conv_rows.hip); False = the edge-streaming kernels of conv.hip for every layer
ROW_KERNELS = _env.flag("ROW_KERNELS")
ROWS_MIN_DEGREE = _env.debug("rows_min_degree", 16.0)
#: levels of up to this many points: the row kernels read the feature rows of the UNSORTED points in place (featIndex)
UNSORTED_MAX_POINTS = _env.debug("unsorted_max_points", 32768)


class RowPlan:
    """SELL-64 layout of a neighbour list (include/mccnn.h, mccnn_rowplan_*): ONE device buffer; vrow ... other are the
    device ADDRESSES of its pieces (plain ints, handed to the C-ABI as they are). Every size is fixed by (rows, e), so
    building a plan involves no host read-back."""
    __slots__ = ("buf", "vrow", "vcode", "slice_off", "vpos_row", "rec", "other", "scratch_rows", "row_start", "key", "event",
                 "num_slices", "slot_capacity")


_PLAN_LAYOUTS = {}  # (rows, e, transposed) -> (offsets[6], total bytes, S, slot capacity, scratch rows, workspace bytes)


def _plan_layout(lib, rows, e, transposed):
    key = (rows, e, transposed)
    hit = _PLAN_LAYOUTS.get(key)
    if hit is None:
        offs = (C.c_longlong * 6)()
        total, cap, srows, S = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        check(lib.mccnn_rowplan_buffer(rows, e, offs, C.byref(total), C.byref(S), C.byref(cap), C.byref(srows)), "rowplan_buffer")
        if len(_PLAN_LAYOUTS) > 512:
            _PLAN_LAYOUTS.clear()
        hit = _PLAN_LAYOUTS[key] = (tuple(offs), total.value, S.value, cap.value, srows.value,
                                    lib.mccnn_rowplan_build_workspace_bytes(rows, e, int(transposed)),
                                    bool(lib.mccnn_rowplan_inline_records(rows, e)))
    return hit


def _row_plan(packed_obj, transposed, pts, bids, pdfs, smp, st, pk, mn, mx, n, m, e, batchSize, radius, scaleInv, avg,
              centre_points=None):
    """The forward (rows = centres) or transposed (rows = neighbour points) row plan of a neighbour list, built on
    first use -- one allocation, one library call -- and kept ON the neighbour-list tensor object: like the transposed
    list, it lives as long as the builder's cache entry and is shared by every layer over the list (same PDFs, radius
    and avg flag)."""
    transposed = bool(transposed)
    key = (transposed, pdfs.data_ptr(), pdfs._version, bool(avg), float(radius), bool(scaleInv), pk._version, n, m)
    plans = getattr(packed_obj, "_mccnn_rowplans", None)
    if plans is None:
        plans = {}
        try:
            packed_obj._mccnn_rowplans = plans
        except AttributeError:
            pass
    hit = plans.get(transposed)
    if hit is not None and hit.key == key:
        if hit.event is not None:  # built ahead of time on another stream (prefetch_rowplan): order this stream behind it
            torch.cuda.current_stream().wait_event(hit.event)
            hit.buf.record_stream(torch.cuda.current_stream())   # ... and its memory behind this stream's sweeps
        return hit
    lib = _lib.load()
    dev = pk.device
    start_t = perm_t = order = None
    tready = 1
    if transposed:
        rows = n
        tl = getattr(packed_obj, "_mccnn_transposed", None)
        if tl is not None and tl[2] == (packed_obj._version, n):
            start_t, perm_t, _ = _transposed_neighbors(packed_obj, n)  # waits for its event if it was built elsewhere
        else:  # written by the same call, ahead of the layout
            start_t = torch.empty(n + 1, dtype=torch.int32, device=dev)
            perm_t = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
            tready = 0
        row_start = start_t
    else:
        rows, row_start = m, st
        order = _order_hint(centre_points) if centre_points is not None else None
        if order is not None and order.shape[0] != m:
            order = None
    offs, total, S, cap, srows, wsb, inline_rec = _plan_layout(lib, rows, e, transposed)
    plan = RowPlan()
    plan.key, plan.event, plan.row_start = key, None, row_start
    plan.scratch_rows, plan.num_slices, plan.slot_capacity = srows, S, cap
    buf = plan.buf = torch.empty(total, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    plan.vrow, plan.vcode, plan.slice_off, plan.vpos_row, plan.other, plan.rec = [base + o for o in offs]
    # the per-edge records in edge order: written once per list, permuted into both plans
    rec_e = plans.get("rec_edges")
    rready = 1
    if inline_rec:  # a small list: the fill evaluates the records itself, no edge-order array
        rec_e = (None, None, None)
    elif rec_e is None or rec_e[0] != key[1:]:
        rec_e = plans["rec_edges"] = [key[1:], torch.empty((max(e, 1), 4), dtype=torch.float32, device=dev), None]
        rready = 0
    elif rec_e[2] is not None:  # written on another stream (prefetch_rowplan)
        torch.cuda.current_stream().wait_event(rec_e[2])
    ws = _ws(wsb, dev)
    check(lib.mccnn_rowplan_build(int(transposed), ptr(pts), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk), ptr(mn), ptr(mx),
                                  n, m, e, batchSize, float(radius), int(bool(scaleInv)), int(bool(avg)), ptr(order),
                                  ptr(rec_e[1]), rready, ptr(start_t), ptr(perm_t), tready, base, ptr(ws), ws.numel(),
                                  stream_handle()), "rowplan_build")
    if transposed and not tready:
        try:
            packed_obj._mccnn_transposed = (start_t, perm_t, (packed_obj._version, n))
        except AttributeError:
            pass
    plans[transposed] = plan
    return plan


def prefetch_rowplan(packed, transposed, stream, *args):
    """Builds a row plan of `packed` on `stream` (a side stream) ahead of the pass that will ask for it; the consumer waits
    for the recorded event (see _row_plan). The inputs must be ready on `stream`. args: _row_plan's after `transposed`."""
    main = torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        plans = getattr(packed, "_mccnn_rowplans", None) or {}
        if plans.get(bool(transposed)) is None:
            had_rec = plans.get("rec_edges") is not None
            had_tl = getattr(packed, "_mccnn_transposed", None) is not None
            plan = _row_plan(packed, transposed, *args)
            ev = torch.cuda.Event()
            ev.record(stream)
            plan.event = ev
            if transposed and not had_tl:  # the transposed list was written by the same call, on this stream
                try:
                    packed._mccnn_transposed_event = ev
                except AttributeError:
                    stream.synchronize()
            rec_e = (getattr(packed, "_mccnn_rowplans", None) or {}).get("rec_edges")
            if rec_e is not None and not had_rec:
                # the edge-order records were written (and allocated) on this stream; the other plan of the list will read
                # them on the caller's stream: ordered by the event, lifetime handed to the allocator
                rec_e[2] = ev
                rec_e[1].record_stream(main)


def _rows_shape(combin, fin, feats, rows, e, backward=False):
    """Row-per-lane kernels for this (layer, list)? Depth-wise rows of 8-feature blocks. Every list of <= 500 k edges: yes
    (long rows are cut into pieces, lists of short rows put several slices into a workgroup). Forward on larger lists: see
    below. Backward: the edge-streaming kernels keep
    LARGE lists (> 500 k edges) unless the layer is wide and its rows long: their waves own equal edge ranges, while the
    176-sum sweep of the row kernel, two waves per SIMD, has too few (slice, block) items on such a list to hide its
    quantisation and its per-item reduction when the layer has <= 16 blocks, and pays set-up and window padding too often
    below ~16 edges per point. Measured (rows against streaming, ms): BASELINE cfg3 Up_1_2 (128 features, 41 k points x
    18.9 edges) 0.49 / 0.36, Conv_2 (64, 41 k x 25.4) 0.27 / 0.24, DeConv_1 (128, 131 k x 8.7) 0.54 / 0.42; the 100k room
    (45 edges per point) 64 features 1.07 / 0.87, 128: 1.67 / 1.53, 256: 2.43 / 2.73; cfg3 Up_1_3 (256, 3 495 points x 386)
    0.77 / 1.18; every list below 500 k edges: rows, by up to 6x on the coarse levels."""
    if not (ROW_KERNELS and _DEBUG_IMPL == 0 and (not combin) and fin % 8 == 0 and e > 0 and rows > 0
            and (feats.data_ptr() & 15) == 0):
        return False
    if e <= 500000:
        return True
    if not backward:
        # the row kernel walks windows of 1 024 centres sorted by row length, the streaming kernel the centres in their
        # cell-coherent order: on a large list whose gathered rows do not fit the L2s (>= 64 k points) and whose layer is
        # narrow (<= 16 blocks: little arithmetic per gathered byte) the better locality wins -- room, 64 features 0.38 /
        # 0.32 ms; cfg3 Pool_1 0.16 / 0.12, DeConv_1 0.24 / 0.21; the other way round on 41 k points (Conv_2 0.13 / 0.15)
        return not (fin <= 128 and feats.shape[0] >= 65536)
    return fin >= 256 and e / float(rows) >= ROWS_MIN_DEGREE


def clear_caches():
    """Drop the per-shape launch hints (visiting orders, edge-count guesses, num_cells read-backs). They only affect
    speed, never results, and are bounded in size; call this to release the device tensors they hold."""
    _ORDER_HINTS.clear()
    _WS_POOL.clear()
    _NUM_CELLS_CACHE.clear()
    _EDGE_GUESS.clear()
    _EDGE_RATIO.clear()


#: debug aid: raise MCCNN_E_BATCHID from compute_aabb (the entry of every op chain) when a batch id lies outside
#: [0, batchSize) -- one 4-byte read-back per call. The kernels clamp ids either way (memory-safe).
CHECK_BATCH_IDS = False


def check_batch_ids(inBatchIds, batchSize):
    """Number of batch ids outside [0, batchSize) (synchronises)."""
    bids = _i32(inBatchIds, "batch_ids")
    bad = torch.empty(1, dtype=torch.int32, device=bids.device)
    check(_lib.load().mccnn_check_batch_ids(ptr(bids), bids.shape[0], int(batchSize), ptr(bad), stream_handle()),
          "check_batch_ids")
    return int(bad.item())


_DEBUG_IMPL = 0


def debug_conv_impl(mask):
    """Test hook: bit 0 = VALU fallback kernels, bit 1 = general MFMA kernels for Fin = 1, bit 2 (this layer only) = the
    edge-streaming MFMA kernels for depth-wise layers instead of the row-per-lane ones. Returns the previous mask."""
    global _DEBUG_IMPL
    prev = _DEBUG_IMPL
    _DEBUG_IMPL = int(mask)
    _lib.load().mccnn_debug_conv_impl(int(mask) & 3)
    return prev


def _num_cells(aabbMin, aabbMax, batchSize, cellSize, scaleInv):
    """determineNumCells (sort_gpu.cu:397-420). scaleInv=False needs the box extent on the host: ONE 24-byte read-back
    per box tensor pair (mccnn_aabb_extent), cached per tensor OBJECT (weak reference + version counter -- never by
    address, the allocator recycles addresses); every grid over the same boxes -- all levels of a hierarchy, all
    convolution radii -- then computes its cell count on the host with the reference's float arithmetic."""
    lib = _lib.load()
    if scaleInv:
        out = C.c_int(0)
        check(lib.mccnn_num_cells(None, None, batchSize, float(cellSize), 1, C.byref(out), None), "num_cells")
        return out.value
    key = (id(aabbMin), id(aabbMax))
    hit = _NUM_CELLS_CACHE.get(key)
    ext = None
    if hit is not None:
        rmin, rmax, vmin, vmax, val = hit
        if rmin() is aabbMin and rmax() is aabbMax and vmin == aabbMin._version and vmax == aabbMax._version:
            ext = val
    if ext is None:
        out = C.c_float(0.0)
        check(lib.mccnn_aabb_extent(ptr(aabbMin), ptr(aabbMax), C.byref(out), stream_handle()), "aabb_extent")
        ext = _np.float32(out.value)
        _evict_oldest(_NUM_CELLS_CACHE, 256)
        _NUM_CELLS_CACHE[key] = (weakref.ref(aabbMin), weakref.ref(aabbMax), aabbMin._version, aabbMax._version, ext)
    _req(cellSize > 0, "cell size must be positive")
    nc = int(ext / _np.float32(cellSize))   # float32 divide, truncation: sort_gpu.cu:415-416
    return nc if nc != 0 else 1


def _seed_num_cells(aabbMin, aabbMax, extent):
    """The box extent of a prefetched hierarchy (read back on its helper thread): later grids over these boxes compute their
    cell counts from it without a read-back of their own."""
    _evict_oldest(_NUM_CELLS_CACHE, 256)
    _NUM_CELLS_CACHE[(id(aabbMin), id(aabbMax))] = (weakref.ref(aabbMin), weakref.ref(aabbMax), aabbMin._version,
                                                    aabbMax._version, _np.float32(extent))


# ---------------------------------------------------------------------------------------------
_AABB_EXT = _env.debug("aabb_ext", True)   # A/B: 0 = the ctypes op


def compute_aabb(inPts, inBatchIds, batchSize, scaleInv=True):
    """ComputeAabb (MCConvModuleSrc:20, aabb_gpu.cc:22-86). Non differentiable."""
    op = "ComputeAabbOp"
    _req(batchSize > 0, op + " expects a positive batch size")
    pts, bids = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    _check_points(pts, "points", op)
    _check_batch_ids(bids, pts.shape[0], op)
    lib = _lib.load()
    if CHECK_BATCH_IDS and check_batch_ids(bids, batchSize):
        check(-2, op)  # MCCNN_E_BATCHID
    ext = _torch_ext()
    if ext is not None and _AABB_EXT and pts.is_cuda and bids.dim() in (1, 2):   # one C++ call (the op every hierarchy of a step starts with)
        mn, mx = ext.compute_aabb(pts, bids.view(-1), int(batchSize), bool(scaleInv))
        return mn, mx
    mn = torch.empty((batchSize, 3), dtype=torch.float32, device=pts.device)
    mx = torch.empty_like(mn)
    ws = _ws(lib.mccnn_compute_aabb_workspace_bytes(batchSize), pts.device)
    check(lib.mccnn_compute_aabb(ptr(pts), ptr(bids), pts.shape[0], batchSize, int(bool(scaleInv)), ptr(mn), ptr(mx),
                                 ptr(ws), ws.numel(), stream_handle()), "compute_aabb")
    return mn, mx


def sort_points_step1(inPts, inBatchIds, aabbMin, aabbMax, batchSize, cellSize, scaleInv):
    """SortPointsStep1 (MCConvModuleSrc:24, sort_gpu.cc:184-268) -> (keys, indexs). Non differentiable."""
    op = "SortPointsStep1Op"
    _req(batchSize > 0, op + " expects a positive batch size")
    _req(cellSize > 0, op + " expects a positive cell size")
    pts, bids = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    _check_points(pts, "points", op)
    _check_batch_ids(bids, pts.shape[0], op)
    _check_aabb(mn, mx, batchSize, op)
    lib = _lib.load()
    n = pts.shape[0]
    nc = _num_cells(mn, mx, batchSize, cellSize, scaleInv)
    keys = torch.empty(n, dtype=torch.int32, device=pts.device)
    idx = torch.empty(n, dtype=torch.int32, device=pts.device)
    wsb = lib.mccnn_sort_step1_workspace_bytes(n, batchSize, nc)
    _req(wsb > 0, op + ": batch_size * num_cells^3 does not fit 32-bit keys")
    ws = _ws(wsb, pts.device)
    check(lib.mccnn_sort_step1(ptr(pts), ptr(bids), ptr(mn), ptr(mx), n, batchSize, nc, ptr(keys), ptr(idx), ptr(ws),
                               ws.numel(), stream_handle()), "sort_points_step1")
    return keys, idx


def build_grid(inPts, inBatchIds, aabbMin, aabbMax, batchSize, cellSize, scaleInv):
    """sort_points_step1 + sort_points_step2 of the points alone, in one library call (extension for
    ConvolutionBuilder: the feature rows are sorted by the convolution that consumes them, spatial_conv(sortIndex=)).
    -> (sortPts, sortBatchs, cellIndexs, index_new_pos, inverse permutation), the builder's grid tuple plus the sorted ->
    unsorted row index. Not differentiable: points that require a gradient take the two ops."""
    op = "SortPointsStep1Op"
    _req(batchSize > 0, op + " expects a positive batch size")
    pts, bids = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    _check_points(pts, "points", op)
    n = pts.shape[0]
    _check_batch_ids(bids, n, op)
    _check_aabb(mn, mx, batchSize, op)
    lib = _lib.load()
    dev = pts.device
    nc = _num_cells(mn, mx, batchSize, cellSize, scaleInv)
    wsb = lib.mccnn_build_grid_workspace_bytes(n, batchSize, nc)
    _req(wsb > 0, op + ": batch_size * num_cells^3 does not fit 32-bit keys")
    ws = _ws(wsb, dev)
    # separate allocations: the builder's lifetime bookkeeping of prefetched grids counts the owners of a STORAGE
    idx = torch.empty(n, dtype=torch.int32, device=dev)
    inv = torch.empty(n, dtype=torch.int32, device=dev)
    oB = torch.empty((n, 1), dtype=torch.int32, device=dev)
    oP = torch.empty_like(pts)
    cells = torch.empty((batchSize, nc, nc, nc, 2), dtype=torch.int32, device=dev)
    check(lib.mccnn_build_grid(ptr(pts), ptr(bids), ptr(mn), ptr(mx), n, batchSize, nc, ptr(idx), ptr(oP), ptr(oB), ptr(cells),
                               ptr(inv), ptr(ws), ws.numel(), stream_handle()), "build_grid")
    # a weak reference: the grid tuple owns the permutation
    _remember_order(inPts, "order_weak", weakref.ref(inv))
    return oP, oB, cells, idx, inv


def _gather_rows(src, idx, n_rows):
    lib = _lib.load()
    w = _rows32(src)
    out = torch.empty((n_rows, w.shape[1]), dtype=torch.float32, device=src.device)
    check(lib.mccnn_permute_gather(ptr(w), ptr(idx), n_rows, w.shape[1], ptr(out), stream_handle()),
          "permute_gather")
    return _like_rows(out, src)


def _scatter_rows(src, idx, n_out, zero_fill):
    lib = _lib.load()
    w = _rows32(src)
    out = torch.empty((n_out, w.shape[1]), dtype=torch.float32, device=src.device)
    check(lib.mccnn_permute_scatter(ptr(w), ptr(idx), w.shape[0], w.shape[1], ptr(out), n_out,
                                    int(zero_fill), stream_handle()), "permute_scatter")
    return _like_rows(out, src)


class _SortPointsStep2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inPts, inBatchIds, inFeatures, keys, indexs, aabbMin, aabbMax, batchSize, cellSize, scaleInv):
        op = "SortPointsStep2Op"
        _req(batchSize > 0, op + " expects a positive batch size")
        pts, bids = _f32(inPts, "points"), _i32(inBatchIds, "batch_ids")
        feats = _feat(inFeatures, "features")
        keys, indexs = _i32(keys, "keys"), _i32(indexs, "index_new_pos")
        mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
        _check_points(pts, "points", op)
        n = pts.shape[0]
        _check_batch_ids(bids, n, op)
        _req(feats.dim() == 2 and feats.shape[0] == n, op + " expects features with dimensions (numPoints, numFeatures)")
        _req(feats.shape[1] > 0, op + " expects features with at least one component")
        _req(keys.dim() == 1 and keys.shape[0] == n, op + " expects the same number of keys and points")
        _req(indexs.dim() == 1 and indexs.shape[0] == n, op + " expects the same number of indexs and points")
        _check_aabb(mn, mx, batchSize, op)
        lib = _lib.load()
        nc = _num_cells(mn, mx, batchSize, cellSize, scaleInv)
        oP = torch.empty_like(pts)
        oB = torch.empty_like(bids)
        oF = torch.empty_like(feats)
        fw, oFw = _rows32(feats), _rows32(oF)
        cells = torch.empty((batchSize, nc, nc, nc, 2), dtype=torch.int32, device=pts.device)
        ws = _ws(lib.mccnn_sort_step2_workspace_bytes(n), pts.device)
        inv = torch.empty_like(indexs)  # visiting order for find_neighbors over these points, a by-product of the move
        check(lib.mccnn_sort_step2(ptr(pts), ptr(bids), ptr(fw), ptr(keys), ptr(indexs), n, fw.shape[1],
                                   batchSize, nc, ptr(oP), ptr(oB), ptr(oFw), ptr(cells), ptr(inv), ptr(ws), ws.numel(),
                                   stream_handle()), "sort_points_step2")
        ctx.save_for_backward(indexs)
        if inPts.requires_grad:
            ctx.mark_non_differentiable(oB, cells)
        else:
            # the sorted points are cached by the builder and handed to every later convolution over this grid: without
            # a gradient to route they must not tie those graphs to this node (a node reached only through such an edge
            # would still be executed -- with undefined gradients -- by every backward pass that touches a later graph)
            ctx.mark_non_differentiable(oP, oB, cells)
        # outputs nobody differentiates through (the sorted points, usually) arrive as None in backward instead of
        # materialised zero tensors that would be permuted for nothing
        ctx.set_materialize_grads(False)
        ctx.needs = (inPts.requires_grad, inFeatures.requires_grad)
        _remember_order(inPts, "order", inv)
        return oP, oB, oF, cells

    @staticmethod
    def backward(ctx, gPts, gBids, gFeats, gCells):
        # _sort_points_step2_grad (MCConvModuleSrc:30-33): in[i] = out[index_new_pos[i]]
        (indexs,) = ctx.saved_tensors
        n = indexs.shape[0]
        dPts = _gather_rows(_f32(gPts, "grad"), indexs, n) if (gPts is not None and ctx.needs[0]) else None
        dFeats = _gather_rows(_feat(gFeats, "grad"), indexs, n) if (gFeats is not None and ctx.needs[1]) else None
        return dPts, None, dFeats, None, None, None, None, None, None, None


def sort_points_step2(inPts, inBatchIds, inFeatures, keys, indexs, aabbMin, aabbMax, batchSize, cellSize, scaleInv):
    """SortPointsStep2 (MCConvModuleSrc:28-33) -> (sortPts, sortBatchs, sortFeatures, cellIndexs)."""
    return _SortPointsStep2.apply(inPts, inBatchIds, inFeatures, keys, indexs, aabbMin, aabbMax, batchSize, cellSize,
                                  scaleInv)


class _SortFeatures(torch.autograd.Function):
    """sort_features == SortFeaturesBackGrad: out[idx[i]] = in[i] (MCConvModuleSrc:35-39)."""

    @staticmethod
    def forward(ctx, inFeatures, indexs):
        f, idx = _feat(inFeatures, "features"), _i32(indexs, "index_new_pos")
        _req(idx.dim() == 1, "SortFeaturesBackGradOp expects indexs with the following dimensions (numPoints)")
        _req(f.dim() == 2 and f.shape[1] > 0 and f.shape[0] == idx.shape[0],
             "SortFeaturesBackGradOp expects features with dimensions (numPoints, numFeatures)")
        ctx.save_for_backward(idx)
        return _scatter_rows(f, idx, f.shape[0], False)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _gather_rows(_feat(g, "grad"), idx, idx.shape[0]), None


def sort_features(inFeatures, indexs):
    return _SortFeatures.apply(inFeatures, indexs)


class _SortFeaturesBack(torch.autograd.Function):
    """SortFeaturesBack: out[i] = in[idx[i]] (MCConvModuleSrc:41-45)."""

    @staticmethod
    def forward(ctx, inFeatures, indexs):
        f, idx = _feat(inFeatures, "features"), _i32(indexs, "index_new_pos")
        _req(idx.dim() == 1, "SortFeaturesBackOp expects indexs with the following dimensions (numPoints)")
        _req(f.dim() == 2 and f.shape[1] > 0 and f.shape[0] == idx.shape[0],
             "SortFeaturesBackOp expects features with dimensions (numPoints, numFeatures)")
        ctx.save_for_backward(idx)
        return _gather_rows(f, idx, idx.shape[0])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _scatter_rows(_feat(g, "grad"), idx, idx.shape[0], False), None


def sort_features_back(inFeatures, indexs):
    return _SortFeaturesBack.apply(inFeatures, indexs)


def transform_indexs(inIndexs, inNewPositions):
    """TransformIndexs (MCConvModuleSrc:47, sort_gpu.cc:496-533). Non differentiable."""
    a, b = _i32(inIndexs, "curr_indexs"), _i32(inNewPositions, "index_new_pos")
    _req(a.dim() == 1 and b.dim() == 1, "TransformIndexsOp expects indexs with the following dimensions (numPoints)")
    lib = _lib.load()
    out = torch.empty_like(a)
    ws = _ws(lib.mccnn_transform_indexs_workspace_bytes(b.shape[0]), a.device)
    check(lib.mccnn_transform_indexs(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), ptr(ws), ws.numel(),
                                     stream_handle()), "transform_indexs")
    return out


def find_neighbors(inPts, inBatchIds, inPts2, cellIndexs, aabbMin, aabbMax, radius, batchSize, scaleInv):
    """FindNeighbors (MCConvModuleSrc:51, find_neighbors.cc:80-185) -> (startIndexs [M,1], packedNeighs [E,2]).
    Reads E back to the host to size the second output, like the reference (find_neighbors.cu:307-309)."""
    op = "FindNeighborsOp"
    _req(radius > 0.0, op + " expects a positive radius")
    _req(batchSize > 0, op + " expects a positive batch size")
    c, cb = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    p2, cells = _f32(inPts2.detach(), "points2"), _i32(cellIndexs, "cell_indexs")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    _check_points(c, "points", op)
    _check_batch_ids(cb, c.shape[0], op)
    _check_points(p2, "points2", op)
    _req(cells.dim() == 5 and cells.shape[0] == batchSize, op + " expects a five dimension tensor for the cell indices")
    _check_aabb(mn, mx, batchSize, op)
    lib = _lib.load()
    m, nc = c.shape[0], cells.shape[1]
    start = torch.empty((m, 1), dtype=torch.int32, device=c.device)
    box, boxv = _host_mailbox()
    n2 = p2.shape[0]
    ws = _ws(lib.mccnn_find_neighbors_workspace_bytes(m, n2), c.device)
    order = _order_hint(inPts)
    if order is not None and (order.shape[0] != m or _env.debug("nw_no_order", False)):
        order = None
    args = (ptr(c), ptr(cb), m, ptr(p2), n2, ptr(cells), ptr(mn), ptr(mx), batchSize, nc, float(radius),
            int(bool(scaleInv)), ptr(order))
    # The size of the second output is only known on the device: the prefix sum stores the total straight into a
    # pinned host word (no copy is enqueued) and the host polls it.
    total = None
    if COUNT_MAILBOX:
        boxv[0] = -1
        total_ptr = box.data_ptr()
    else:  # the total lands in device memory and is copied to the host (stream-ordered BEFORE the fill)
        total = torch.empty(1, dtype=torch.int32, device=c.device)
        total_ptr = ptr(total)
    check(lib.mccnn_find_neighbors_count(*args, ptr(start), total_ptr, ptr(ws), ws.numel(), stream_handle()),
          "find_neighbors(count)")
    ev = None
    if total is not None:
        box.copy_(total, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()

    def read_total():
        if ev is None:
            return _await_mailbox(boxv)
        ev.synchronize()
        return int(boxv[0])
    # Searches repeat with the same shapes step after step, so the fill is launched into a buffer sized from the last
    # total of this shape BEFORE the total is read: the host round trip hides behind the kernel. Too small a guess ->
    # exact rerun.
    gkey = (c.device.index, m, n2, float(radius), int(batchSize), bool(scaleInv))
    guess = _edge_guess(gkey)
    packed = None
    if guess > 0:
        buf = torch.empty((guess, 2), dtype=torch.int32, device=c.device)
        check(lib.mccnn_find_neighbors_fill(*args, ptr(start), guess, ptr(buf), ptr(ws), ws.numel(), stream_handle()),
              "find_neighbors(fill)")
        e = read_total()
        if e <= guess:
            packed = buf[:e]
    else:
        e = read_total()
    if packed is None:
        packed = torch.empty((e, 2), dtype=torch.int32, device=c.device)
        check(lib.mccnn_find_neighbors_fill(*args, ptr(start), e, ptr(packed), ptr(ws), ws.numel(), stream_handle()),
              "find_neighbors(fill)")
    _remember_edges(gkey, e)
    return start, packed


#: 0 = the reference's double-precision-exp arithmetic, 1 = single-precision (default; ~1e-6 rel. apart)
PDF_MODE = 1
# keep the forward's per-centre sums for the backward pass (layers with one input feature); False = the backward
# recomputes them, as a binding without an extra forward output has to
KEEP_CONV_STATE = True
# Poisson sampling: try the single-launch dataflow form first (falls back to 27 launches on a timed-out wait).
# False = phased form only; 2 = dataflow form that gives up at the first unfinished dependency (tests of the fallback)
POISSON_DATAFLOW = True
POISSON_FALLBACKS = 0  # number of calls that had to repeat with the phased form


def compute_pdf(inPts, inBatchIds, aabbMin, aabbMax, startIndexs, neighbors, window, radius, batchSize, scaleInv,
                mode=None):
    """ComputePDF (MCConvModuleSrc:55, compute_pdf.cc:57-142) -> pdfs [E,1]. Non differentiable."""
    op = "ComputePDFOp"
    _req(radius > 0.0, op + " expects a positive radius")
    _req(window > 0.0, op + " expects a positive window")
    _req(batchSize > 0, op + " expects a positive batch size")
    p, b = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    st, pk = _i32(startIndexs, "start_indexs"), _i32(neighbors, "neighbors")
    _check_points(p, "points", op)
    _check_batch_ids(b, p.shape[0], op)
    _req(st.dim() == 2 and st.shape[1] == 1, op + " expects start indexs with dimensions (numSamples, 1)")
    _req(pk.dim() == 2 and pk.shape[1] == 2, op + " expects a neighbor list with dimensions (numNeighbors, 2)")
    _check_aabb(mn, mx, batchSize, op)
    lib = _lib.load()
    e = pk.shape[0]
    pdfs = torch.empty((e, 1), dtype=torch.float32, device=p.device)
    md = int(PDF_MODE if mode is None else mode)
    ws = _ws(lib.mccnn_compute_pdf_workspace_bytes(e, md), p.device)
    check(lib.mccnn_compute_pdf(ptr(p), ptr(b), ptr(st), st.shape[0], ptr(pk), e, ptr(mn), ptr(mx), batchSize,
                                float(window), float(radius), int(bool(scaleInv)), md, ptr(pdfs), ptr(ws), ws.numel(),
                                stream_handle()), "compute_pdf")
    return pdfs


class DeferredNeighborsPDF:
    """find_neighbors + compute_pdf enqueued WITHOUT a host wait (extension for ConvolutionBuilder.prefetch_geometry):
    the lists are sized from the last total of the same shape, the KDE kernels read the true total from device memory,
    and the total travels to a pinned host word that finalize() reads later -- by then it has long arrived. If the
    guess turns out too small, finalize() repeats both ops with the exact size."""

    def __init__(self, args_fn, pdf_fn, start, packed, pdfs, total_dev, slot, guess, gkey):
        self._args_fn, self._pdf_fn = args_fn, pdf_fn
        self.start, self._packed, self._pdfs, self._total_dev = start, packed, pdfs, total_dev
        self._slot, self._guess, self._gkey = slot, guess, gkey

    def finalize(self):
        """-> (startIndexs [M,1], packedNeighs [E,2], pdfs [E,1]); call on the stream that will consume them, after it
        has been ordered behind the producing stream."""
        if self._slot is None:
            raise RuntimeError("DeferredNeighborsPDF.finalize() called twice")
        e = _await_mailbox(self._slot[1])
        _slot_pool().append(self._slot)
        self._slot = None
        if e <= self._guess:
            packed, pdfs = self._packed[:e], self._pdfs[:e]
        else:
            packed = self._args_fn(e)
            pdfs = self._pdf_fn(packed)
        _remember_edges(self._gkey, e)
        return self.start, packed, pdfs


def _slot_pool():
    pool = getattr(_TLS, "slots", None)
    if pool is None:
        pool = _TLS.slots = []
    return pool


def find_neighbors_pdf_deferred(inPts, inBatchIds, sortedPts, sortedBatchIds, cellIndexs, aabbMin, aabbMax, radius,
                                batchSize, scaleInv, window):
    """See DeferredNeighborsPDF. Returns None when there is no size guess for this shape yet (first call) or the
    reference-arithmetic KDE is selected: the caller then takes find_neighbors() + compute_pdf()."""
    if int(PDF_MODE) != 1:
        return None
    op = "FindNeighborsOp"
    _req(radius > 0.0 and window > 0.0 and batchSize > 0, op + " expects positive radius, window and batch size")
    c, cb = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    p2, b2 = _f32(sortedPts.detach(), "points2"), _i32(sortedBatchIds, "batch_ids2")
    cells = _i32(cellIndexs, "cell_indexs")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    _check_points(c, "points", op)
    _check_batch_ids(cb, c.shape[0], op)
    _check_points(p2, "points2", op)
    _check_batch_ids(b2, p2.shape[0], op)
    _req(cells.dim() == 5 and cells.shape[0] == batchSize, op + " expects a five dimension tensor for the cell indices")
    _check_aabb(mn, mx, batchSize, op)
    m, n2, nc = c.shape[0], p2.shape[0], cells.shape[1]
    gkey = (c.device.index, m, n2, float(radius), int(batchSize), bool(scaleInv))
    guess = _edge_guess(gkey)
    if guess <= 0 or m == 0:
        return None
    lib = _lib.load()
    start = torch.empty((m, 1), dtype=torch.int32, device=c.device)
    total_dev = torch.empty(1, dtype=torch.int32, device=c.device)
    ws = _ws(lib.mccnn_find_neighbors_workspace_bytes(m, n2), c.device)
    order = _order_hint(inPts)
    if order is not None and (order.shape[0] != m or _env.debug("nw_no_order", False)):
        order = None
    args = (ptr(c), ptr(cb), m, ptr(p2), n2, ptr(cells), ptr(mn), ptr(mx), batchSize, nc, float(radius),
            int(bool(scaleInv)), ptr(order))
    pool = _slot_pool()
    if pool:
        slot = pool.pop()
    else:
        t = torch.empty(1, dtype=torch.int32).pin_memory()
        slot = (t, t.numpy())
    slot[1][0] = -1
    # the prefix sum stores the total twice: in device memory for the KDE kernels and straight into the pinned host word
    # (no copy is enqueued); nobody waits for it here
    if _MAILBOX_COPY:  # A/B switch: the total travels through an enqueued device-to-host copy instead
        check(lib.mccnn_find_neighbors_count(*args, ptr(start), ptr(total_dev), ptr(ws), ws.numel(), stream_handle()),
              "find_neighbors(count)")
        slot[0].copy_(total_dev, non_blocking=True)
    else:
        check(lib.mccnn_find_neighbors_count2(*args, ptr(start), ptr(total_dev), slot[0].data_ptr(), ptr(ws), ws.numel(),
                                              stream_handle()), "find_neighbors(count)")
    packed = torch.empty((guess, 2), dtype=torch.int32, device=c.device)
    check(lib.mccnn_find_neighbors_fill(*args, ptr(start), guess, ptr(packed), ptr(ws), ws.numel(), stream_handle()),
          "find_neighbors(fill)")
    pdfs = torch.empty((guess, 1), dtype=torch.float32, device=c.device)
    pws = _ws(lib.mccnn_compute_pdf_workspace_bytes(guess, 1), c.device)
    check(lib.mccnn_compute_pdf_dn(ptr(p2), ptr(b2), ptr(start), m, ptr(packed), guess, ptr(total_dev), ptr(mn), ptr(mx),
                                   batchSize, float(window), float(radius), int(bool(scaleInv)), ptr(pdfs), ptr(pws),
                                   pws.numel(), stream_handle()), "compute_pdf(dn)")

    def refill(e):  # the guess was too small: exact repeat (the count's hit masks are gone: a fresh count first)
        st, pk = find_neighbors(inPts, inBatchIds, sortedPts, cellIndexs, aabbMin, aabbMax, radius, batchSize, scaleInv)
        start.copy_(st)
        return pk

    def repdf(pk):
        return compute_pdf(sortedPts, sortedBatchIds, aabbMin, aabbMax, start, pk, window, radius, batchSize, scaleInv)

    return DeferredNeighborsPDF(refill, repdf, start, packed, pdfs, total_dev, slot, guess, gkey)


def poisson_sampling(inPts, inBatchIds, cellIndexs, aabbMin, aabbMax, radius, batchSize, scaleInv):
    """PoissonSampling (MCConvModuleSrc:59, poisson_sampling.cc:109-211) -> (pts [S,3], batchIds [S,1], indexs [S]).
    indexs point into the SORTED input list. Non differentiable."""
    op = "PoissonSamplingOp"
    _req(radius > 0.0, op + " expects a positive radius")
    _req(batchSize > 0, op + " expects a positive batch size")
    p, b = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    cells = _i32(cellIndexs, "cell_indexs")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    _check_points(p, "points", op)
    _check_batch_ids(b, p.shape[0], op)
    _req(cells.dim() == 5 and cells.shape[0] == batchSize, op + " expects a five dimension tensor for the cell indices")
    _check_aabb(mn, mx, batchSize, op)
    lib = _lib.load()
    n, nc = p.shape[0], cells.shape[1]
    wsb = lib.mccnn_poisson_sampling_workspace_bytes(n, batchSize, nc)
    _req(wsb > 0, op + ": grid too large")
    ws = _ws(wsb, p.device)
    total = torch.empty(1, dtype=torch.int32, device=p.device)
    # mode 1: all 27 colour phases in one launch (cells wait on their earlier-phase neighbours); a timed-out wait reports
    # -1 and the phases are run one launch at a time instead
    global POISSON_FALLBACKS
    s = -1
    for mode in ((2 if POISSON_DATAFLOW == 2 else 1, 0) if POISSON_DATAFLOW else (0,)):
        check(lib.mccnn_poisson_sampling_count(ptr(p), ptr(b), n, ptr(cells), ptr(mn), ptr(mx), batchSize, nc,
                                               float(radius), int(bool(scaleInv)), mode, ptr(total), ptr(ws), ws.numel(),
                                               stream_handle()), "poisson_sampling(count)")
        s = int(total.item())
        if s >= 0:
            break
        POISSON_FALLBACKS += 1
    oP = torch.empty((s, 3), dtype=torch.float32, device=p.device)
    oB = torch.empty((s, 1), dtype=torch.int32, device=p.device)
    oI = torch.empty(s, dtype=torch.int32, device=p.device)
    check(lib.mccnn_poisson_sampling_fill(ptr(p), n, ptr(cells), batchSize, nc, s, ptr(oP), ptr(oB), ptr(oI), ptr(ws),
                                          ws.numel(), stream_handle()), "poisson_sampling(fill)")
    return oP, oB, oI


def _torch_ext():
    """The torch extension over the C-ABI (mccnn_amd/lib/_mccnn_torch.so) when it is built and enabled."""
    from . import native
    return native._EXT


def point_hierarchy_prefetch(inPts, inBatchIds, radiusList, batchSize, scaleInv, after=None, features=None):
    """Boxes and level geometry of a point hierarchy (compute_aabb + point_hierarchy_levels) started on a stream of its own,
    issued by a helper thread of the PyTorch-ROCm extension (csrc/torch_ext.cpp: hierarchy_prefetch): the call returns at once
    with a future, `future.result()` -> (aabbMin, aabbMax, extent, levels) orders the calling stream behind the build. The
    build starts behind what the calling stream holds NOW (after=None), at once (after=True: the inputs are complete) or
    behind a torch.cuda.Event (after=event: the upload's stream recorded it). None when the extension (or the single-launch
    Poisson form) is not available -- the caller then builds the hierarchy inline."""
    ext = _torch_ext()
    if ext is None or not POISSON_DATAFLOW or len(radiusList) == 0 or not getattr(inPts, "is_cuda", False):
        return None
    op = "PointHierarchy"
    pts, bids = inPts.detach(), inBatchIds
    if pts.dtype != torch.float32 or bids.dtype != torch.int32 or not pts.is_contiguous() or not bids.is_contiguous() \
            or pts.shape[0] == 0:
        return None
    _check_points(pts, "points", op)
    _check_batch_ids(bids, pts.shape[0], op)
    _req(batchSize > 0, op + " expects a positive batch size")
    for radius in radiusList:
        _req(radius > 0.0, op + " expects positive radii")
    pmode = 2 if POISSON_DATAFLOW == 2 else _env.debug("hier_pmode", 1)
    mode, handle = 0, 0
    if after is True:
        mode = 1
    elif after is not None and after is not False:
        _req(isinstance(after, torch.cuda.Event), op + ".prefetch: `after` is None, True or a torch.cuda.Event")
        mode, handle = 2, int(after.cuda_event)
    # features (optional): level 0's input feature rows -- rows without a gradient are gathered for every level on the
    # hierarchy's own stream (the extension decides: short contiguous f32 / bf16 rows); the levels then carry a 5th tensor
    feats = features if (features is not None and getattr(features, "is_cuda", False) and not features.requires_grad) else None
    return ext.hierarchy_prefetch(pts, bids, [float(r) for r in radiusList], int(batchSize), bool(scaleInv), pmode, mode, handle,
                                  feats)


def point_hierarchy_levels(inPts, inBatchIds, aabbMin, aabbMax, radiusList, batchSize, scaleInv):
    """Geometry of ALL levels of a point hierarchy (MCConvBuilder.py:101-128: sort_points_step1/2 -> poisson_sampling ->
    transform_indexs per level) with ONE host read-back at the end instead of one per level: every level takes its point
    count from device memory (mccnn_*_dn), launches and buffers are sized by the first level's point count.
    Returns [(sampledPts [S,3], sampledBatchIds [S,1], sampledIndexs [S] into the level's sorted list,
    transformedIndexs [S] into the level's input order)] per level -- bit-identical to the op-by-op chain -- or None
    if a wait of the single-launch Poisson kernel timed out somewhere (the caller then runs the op-by-op chain)."""
    op = "PointHierarchy"
    pts, bids = _f32(inPts.detach(), "points"), _i32(inBatchIds, "batch_ids")
    mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
    _check_points(pts, "points", op)
    _check_batch_ids(bids, pts.shape[0], op)
    _check_aabb(mn, mx, batchSize, op)
    lib = _lib.load()
    dev, cap, L = pts.device, pts.shape[0], len(radiusList)
    if L == 0:
        return []
    if cap == 0 or not POISSON_DATAFLOW:
        return None  # an empty cloud, or the single-launch Poisson kernel switched off: the op-by-op chain handles it
    pmode = 2 if POISSON_DATAFLOW == 2 else 1
    for radius in radiusList:
        _req(radius > 0.0, op + " expects positive radii")
    ext = _torch_ext()
    if ext is not None:
        # the same sequence from C++ (csrc/torch_ext.cpp): no Python between the levels, sizes through a pinned buffer
        ncs = [_num_cells(mn, mx, batchSize, r, scaleInv) for r in radiusList]
        lv = ext.hierarchy_levels(pts, bids, mn, mx, [float(r) for r in radiusList], ncs, batchSize, bool(scaleInv), pmode)
        if not lv:
            return None
        return [tuple(x) for x in lv]
    sizes = torch.empty(L + 1, dtype=torch.int32, device=dev)
    sizes[0] = cap
    sizes_ptr = sizes.data_ptr()
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    cur_pts, cur_bids, levels = pts, bids, []
    si = int(bool(scaleInv))
    for l, radius in enumerate(radiusList):
        _req(radius > 0.0, op + " expects positive radii")
        nc = _num_cells(mn, mx, batchSize, radius, scaleInv)
        wsb = lib.mccnn_hierarchy_level_workspace_bytes(cap, batchSize, nc)
        _req(wsb > 0, op + ": batch_size * num_cells^3 does not fit 32-bit keys")
        ws = _ws(wsb, dev)
        # one call per level: both sort steps, the Poisson sampling (count + fill) and transform_indexs, every count in
        # device memory; two allocations per level (the level's int32 and float32 rows)
        ca = (cap + 63) // 64 * 64  # 256-byte aligned pieces (the cell table is written as int2)
        ints = i32(5 * ca + 2 * batchSize * nc * nc * nc)  # index_new_pos | sorted batch ids | oB | oI | ti | cell table
        flts = f32(6 * ca)                                 # sorted points | sampled points
        oB, oI, ti = ints[2 * ca:2 * ca + cap], ints[3 * ca:3 * ca + cap], ints[4 * ca:4 * ca + cap]
        oP = flts[3 * ca:3 * ca + 3 * cap].view(cap, 3)
        base_i, base_f = ints.data_ptr(), flts.data_ptr()
        check(lib.mccnn_hierarchy_level(ptr(cur_pts), ptr(cur_bids), ptr(mn), ptr(mx), cap, sizes_ptr + 4 * l, batchSize, nc,
                                        float(radius), si, pmode, base_i, base_f, base_i + 4 * ca, base_i + 20 * ca,
                                        base_f + 12 * ca, base_i + 8 * ca, base_i + 12 * ca, base_i + 16 * ca,
                                        sizes_ptr + 4 * (l + 1), ptr(ws), ws.numel(), stream_handle()), "hierarchy_level")
        levels.append((oP, oB.view(cap, 1), oI, ti))
        cur_pts, cur_bids = oP, oB
    t_wait = time.perf_counter()
    host = sizes.cpu().tolist()  # the ONE read-back: every level's sample count
    HOST_WAIT_S[0] += time.perf_counter() - t_wait
    if any(s < 0 for s in host[1:]):
        return None
    out = []
    for (oP, oB, oI, ti), s in zip(levels, host[1:]):
        sp, sb, si_, t_ = oP[:s], oB[:s], oI[:s], ti[:s]
        out.append((sp, sb, si_, t_))
    return out


class _GetSampledFeatures(torch.autograd.Function):
    """GetSampledFeatures (+Grad: scatter with zero fill), MCConvModuleSrc:63-68."""

    @staticmethod
    def forward(ctx, inSampledIndexs, pInFeatures):
        idx, f = _i32(inSampledIndexs, "sampled_indexs"), _feat(pInFeatures, "features")
        _req(idx.dim() == 1, "GetSampledFeaturesOp expects indexs with the following dimensions (numSamples)")
        _req(f.dim() == 2 and f.shape[1] > 0, "GetSampledFeaturesOp expects features with dimensions (numPoints, numFeatures)")
        ctx.save_for_backward(idx)
        ctx.n = f.shape[0]
        return _gather_rows(f, idx, idx.shape[0])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return None, _scatter_rows(_feat(g, "grad"), idx, ctx.n, True)


def get_sampled_features(inSampledIndexs, pInFeatures):
    ext = _torch_ext()
    if (ext is not None and torch.is_tensor(pInFeatures) and pInFeatures.is_cuda and pInFeatures.dim() == 2
            and pInFeatures.is_contiguous() and torch.is_tensor(inSampledIndexs) and inSampledIndexs.is_cuda
            and inSampledIndexs.dtype == torch.int32 and inSampledIndexs.dim() == 1 and inSampledIndexs.is_contiguous()
            and (pInFeatures.dtype == torch.float32 or (pInFeatures.dtype == torch.bfloat16 and pInFeatures.shape[1] % 2 == 0))
            and pInFeatures.shape[1] > 0):
        return ext.sampled_features(inSampledIndexs, pInFeatures)   # the same two library calls, the autograd node in C++
    return _GetSampledFeatures.apply(inSampledIndexs, pInFeatures)


# ---------------------------------------------------------------------------------------------
def _conv_checks(op, pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, b1, w2, b2, w3, b3, numOutFeatures, combin,
                 batchSize, radius):
    bs = get_block_size()
    _req(numOutFeatures > 0, op + " expects a positive number of output features")
    _req(radius > 0.0, op + " expects a positive radius")
    _req(batchSize > 0, op + " expects a positive batch size")
    _check_points(pts, "points", op)
    n = pts.shape[0]
    _req(feats.dim() == 2 and feats.shape[0] == n, op + " expects as feature inputs the following dimensions (numPoints, numFeatures)")
    _check_batch_ids(bids, n, op)
    _req(pdfs.dim() == 2 and pdfs.shape[1] == 1, op + " expects as pdfs the following dimensions (numNeighbors, 1)")
    e = pdfs.shape[0]
    _check_points(smp, "sample points", op)
    m = smp.shape[0]
    _req(st.dim() == 2 and st.shape[1] == 1 and st.shape[0] == m, op + " expects start indexs with dimensions (numSamples, 1)")
    _req(pk.dim() == 2 and pk.shape[0] == e and pk.shape[1] == 2, op + " expects a neighbor list with dimensions (numNeighbors, 2)")
    _check_aabb(mn, mx, batchSize, op)
    _req(w1.dim() == 2 and w1.shape[0] == 3 and b1.dim() == 1 and w1.shape[1] == b1.shape[0] and w1.shape[1] % bs == 0,
         op + " expects a correct first hidden layer")
    _req(w2.dim() == 2 and w2.shape[0] == bs and b2.dim() == 1 and w2.shape[1] == b2.shape[0] and w2.shape[1] == w1.shape[1],
         op + " expects a correct second hidden layer")
    _req(w3.dim() == 2 and w3.shape[0] == bs and b3.dim() == 1 and w3.shape[1] == b3.shape[0],
         op + " expects a correct output layer")
    fin = feats.shape[1]
    _req(w3.shape[1] % fin == 0, op + " expects a number of output neurons multiple of the number of features.")
    if not combin:
        _req(w3.shape[1] == fin, op + " expects the same number of features in the input and the output")
    neurons = fin * numOutFeatures if combin else fin
    nb = (neurons + bs - 1) // bs
    _req(w1.shape[1] == nb * bs and w3.shape[1] == nb * bs, op + " expects %d neurons per layer" % (nb * bs))
    return n, m, e, fin


class _SpatialConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs, aabbMin,
                aabbMax, weights1, biases1, weights2, biases2, weightsOut, biasesOut, numOutFeatures, combin,
                batchSize, radius, scaleInv, avg, sortIndex=None, trusted=False, featIndex=None):
        op = "SpatialConvOp"
        feats = _feat(inFeatures, "features")
        bf16 = feats.dtype == torch.bfloat16
        # the kernel-MLP tensors may arrive in the shapes the builder STORES them in ([numBlocks, bs, bs] / [numBlocks, bs],
        # MCConvBuilder.py:407-419) -- the same memory as the [bs, numBlocks*bs] / [numBlocks*bs] operands of the op, so the
        # reshape happens here instead of as four extra nodes of every convolution's graph
        ctx.wshapes = (weights2.shape, biases2.shape, weightsOut.shape, biasesOut.shape)
        if weights2.dim() == 3:
            weights2 = weights2.reshape(weights2.shape[1], -1)
        if weightsOut.dim() == 3:
            weightsOut = weightsOut.reshape(weightsOut.shape[1], -1)
        if biases2.dim() == 2:
            biases2 = biases2.reshape(-1)
        if biasesOut.dim() == 2:
            biasesOut = biasesOut.reshape(-1)
        if trusted:
            # ConvolutionBuilder's own call: the geometry tensors are outputs of this module's ops and the kernel-MLP
            # tensors the builder's variables -- types, layouts and the shape rules hold by construction
            # (the sample points of level 0 are the caller's tensor, and the module may have been cast: those two are
            # still looked at)
            pts, bids, pdfs, st, pk, mn, mx = inPts, inBatchIds, inPDFs, neighStartIndexs, packedNeighs, aabbMin, aabbMax
            smp, w1 = _f32(inSamplePts, "sample_pts"), _f32(weights1, "weight_hidden_1")
            b1, w2, b2, w3, b3 = biases1, weights2, biases2, weightsOut, biasesOut
            _req(w3.dtype == torch.float32 and w3.device == w1.device, op + " expects float32 kernel-MLP tensors on one device")
        else:
            pts, bids = _f32(inPts, "points"), _i32(inBatchIds, "batch_ids")
            pdfs, smp = _f32(inPDFs, "pdfs"), _f32(inSamplePts, "sample_pts")
            st, pk = _i32(neighStartIndexs, "start_neighs_indexs"), _i32(packedNeighs, "neighs_indexs")
            mn, mx = _f32(aabbMin, "aabb_min"), _f32(aabbMax, "aabb_max")
            w1, b1 = _f32(weights1, "weight_hidden_1"), _f32(biases1, "bias_hidden_1")
            w2, b2 = _f32(weights2, "weight_hidden_2"), _f32(biases2, "bias_hidden_2")
            w3, b3 = _f32(weightsOut, "weight_out_layer"), _f32(biasesOut, "bias_out_layer")
        sidx = None
        unsorted = False
        if sortIndex is not None:
            # the feature rows arrive in the order of the UNSORTED points: sort_features folded into this node (one
            # autograd node and one op call less per convolution; same two kernels)
            sidx = _i32(sortIndex, "index_new_pos")
            _req(sidx.dim() == 1 and feats.dim() == 2 and feats.shape[0] == sidx.shape[0],
                 "SortFeaturesBackGradOp expects features with dimensions (numPoints, numFeatures)")
            # ... or no sorted copy at all: on a SMALL level whose layer takes the row kernels in both directions, those
            # read the rows where they lie (feat_index = the grid's inverse permutation) and write the feature gradient
            # back in the same order -- two launches and two allocations less per convolution. Large levels keep the
            # sorted copy: its gathers follow the grid's spatial order.
            e_ = inPDFs.shape[0]
            unsorted = (featIndex is not None and feats.shape[0] <= UNSORTED_MAX_POINTS and e_ > 0 and inSamplePts.shape[0] > 0
                        and _rows_shape(combin, feats.shape[1], feats, inSamplePts.shape[0], e_)
                        and _rows_shape(combin, feats.shape[1], feats, feats.shape[0], e_, backward=True))
            if not unsorted:
                feats = _scatter_rows(feats, sidx, feats.shape[0], False)
        # saved with the other tensors: released when the backward pass has run
        sx = () if sidx is None else ((sidx, featIndex) if unsorted else (sidx,))
        ctx.unsorted = unsorted
        if trusted:
            n, m, e, fin = pts.shape[0], smp.shape[0], pdfs.shape[0], feats.shape[1]
            _req(feats.dim() == 2 and feats.shape[0] == n,
                 op + " expects as feature inputs the following dimensions (numPoints, numFeatures)")
            # the two shape rules that depend on the CALLER's layer shape (spatial_conv.cc:290-296)
            _req(w3.shape[1] % fin == 0, op + " expects a number of output neurons multiple of the number of features.")
            _req(combin or w3.shape[1] == fin, op + " expects the same number of features in the input and the output")
        else:
            n, m, e, fin = _conv_checks(op, pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, b1, w2, b2, w3, b3,
                                        numOutFeatures, combin, batchSize, radius)
        lib = _lib.load()
        outF = numOutFeatures if combin else fin
        if _rows_shape(combin, fin, feats, m, e):
            # depth-wise layer on the row-per-lane kernels: forward plan of the neighbour list (built once per list)
            plan = _row_plan(packedNeighs if pk is packedNeighs else pk, False, pts, bids, pdfs, smp, st, pk, mn, mx, n, m,
                             e, batchSize, radius, scaleInv, avg, centre_points=inSamplePts)
            out = torch.empty((m, outF), dtype=feats.dtype, device=pts.device)
            scratch = torch.empty((plan.scratch_rows, outF), dtype=torch.float32, device=pts.device)
            check(lib.mccnn_spatial_conv_fwd_rows(ptr(pts), ptr(feats), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk),
                                                  ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                                  n, m, e, fin, batchSize, float(radius), int(bool(scaleInv)),
                                                  int(bool(avg)), int(bf16), plan.vrow, plan.vcode,
                                                  plan.slice_off, plan.vpos_row, plan.rec, plan.other,
                                                  ptr(out), ptr(scratch), ptr(featIndex) if unsorted else None, stream_handle()),
                  "spatial_conv(rows)")
            ctx.save_for_backward(pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, b1, w2, b2, w3, b3, *sx)
            ctx.state = None
            ctx.packed_ref = weakref.ref(packedNeighs if pk is packedNeighs else pk)
            ctx.attrs = (numOutFeatures, bool(combin), batchSize, float(radius), bool(scaleInv), bool(avg))
            return out
        ws = _ws(lib.mccnn_spatial_conv_fwd_workspace_bytes(m, e, fin, numOutFeatures, int(bool(combin))), pts.device)
        if bf16:
            # bf16 feature storage (extension): depth-wise layers only, rows in / rows out as bf16, f32 arithmetic
            _req(not combin and fin % 8 == 0, op + ": bfloat16 feature storage needs a depth-wise layer with numFeatures % 8 == 0")
            out = torch.empty((m, outF), dtype=torch.bfloat16, device=pts.device)
            check(lib.mccnn_spatial_conv_fwd_bf16(ptr(pts), ptr(feats), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk),
                                                  ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                                  n, m, e, fin, batchSize, float(radius), int(bool(scaleInv)),
                                                  int(bool(avg)), ptr(out), ptr(ws), ws.numel(), stream_handle()),
                  "spatial_conv(bf16)")
            ctx.save_for_backward(pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, b1, w2, b2, w3, b3, *sx)
            ctx.state = None
            ctx.packed_ref = weakref.ref(packedNeighs if pk is packedNeighs else pk)
            ctx.attrs = (numOutFeatures, bool(combin), batchSize, float(radius), bool(scaleInv), bool(avg))
            return out
        out = torch.empty((m, outF), dtype=torch.float32, device=pts.device)
        # what the backward pass can reuse: per-edge records (16 B per edge) and, for layers with one input feature, the
        # per-centre sums; only kept when a gradient will be asked for
        state = None
        sbytes = lib.mccnn_spatial_conv_state_bytes(m, e, fin, numOutFeatures, int(bool(combin)))
        if KEEP_CONV_STATE and sbytes and e > 0 and any(t.requires_grad for t in (inFeatures, w1, b1, w2, b2, w3, b3)):
            state = torch.empty(sbytes, dtype=torch.uint8, device=pts.device)
        check(lib.mccnn_spatial_conv_fwd(ptr(pts), ptr(feats), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk),
                                         ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), n, m,
                                         e, fin, numOutFeatures, int(bool(combin)), batchSize, float(radius),
                                         int(bool(scaleInv)), int(bool(avg)), ptr(out), ptr(state), ptr(ws), ws.numel(),
                                         stream_handle()), "spatial_conv")
        ctx.save_for_backward(pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, b1, w2, b2, w3, b3, *sx)
        ctx.state = state
        # the builder's cached tensor OBJECT carries the transposed list (see _transposed_neighbors). A weak reference: the
        # graph object outlives its backward pass (as long as the caller keeps the output or the loss), and a strong one
        # would keep the list (and its transposed form and plans) alive for that long
        ctx.packed_ref = weakref.ref(packedNeighs if pk is packedNeighs else pk)
        ctx.attrs = (numOutFeatures, bool(combin), batchSize, float(radius), bool(scaleInv), bool(avg))
        return out

    @staticmethod
    def backward(ctx, outGrad):
        # _spatial_conv_grad (MCConvModuleSrc:74-81): grads for features and the 6 MLP tensors only
        saved = ctx.saved_tensors
        pts, feats, bids, pdfs, smp, st, pk, mn, mx, w1, b1, w2, b2, w3, b3 = saved[:15]
        sort_index = saved[15] if len(saved) > 15 else None
        feat_index = saved[16] if len(saved) > 16 else None  # ctx.unsorted: `feats` are the rows of the unsorted points
        numOutFeatures, combin, batchSize, radius, scaleInv, avg = ctx.attrs
        bf16 = feats.dtype == torch.bfloat16
        og = _feat(outGrad, "out_features_grad")
        if bf16 and og.dtype != torch.bfloat16:
            og = og.to(torch.bfloat16)
        lib = _lib.load()
        n, fin = feats.shape
        m, e = smp.shape[0], pk.shape[0]
        fg = torch.empty_like(feats)
        # the six MLP gradients are consecutive slices of ONE buffer, in the order the builder creates the variables:
        # a data-parallel step all-reduces that buffer as it is (dist.GradBucket), with no packing kernel in between
        gflat = torch.empty(w1.numel() + b1.numel() + w2.numel() + b2.numel() + w3.numel() + b3.numel(),
                            dtype=w1.dtype, device=w1.device)
        dw1, db1, dw2, db2, dw3, db3 = gflat.split([w1.numel(), b1.numel(), w2.numel(), b2.numel(), w3.numel(), b3.numel()])
        dw1, dw2, dw3 = dw1.view_as(w1), dw2.view_as(w2), dw3.view_as(w3)
        ws2, bs2, ws3, bs3 = ctx.wshapes  # gradients in the shapes the tensors came in
        packed_obj = ctx.packed_ref()
        if packed_obj is None:  # the list object is gone (its cache entry was dropped): the saved tensor has the same rows
            packed_obj = pk
        rows_bwd = _rows_shape(combin, fin, feats, n, e, backward=True) and m > 0 and (og.data_ptr() & 15) == 0
        if feat_index is not None and not rows_bwd:
            # (an out-gradient the row kernel cannot take, e.g. a misaligned view: the streaming kernels want sorted rows)
            feats = _scatter_rows(feats, sort_index, n, False)
            feat_index = None
        if rows_bwd:
            # depth-wise layer: ONE sweep over the transposed row plan finishes the feature gradient and the six
            # parameter gradients (the edge-major kernels evaluate the kernel MLP twice for that)
            plan = _row_plan(packed_obj, True, pts, bids, pdfs, smp, st, pk, mn, mx, n, m, e, batchSize, radius, scaleInv, avg)
            ws = _ws(lib.mccnn_spatial_conv_bwd_rows_workspace_bytes(n, e, fin), pts.device)
            scratch = torch.empty((plan.scratch_rows, fin), dtype=torch.float32, device=pts.device)
            check(lib.mccnn_spatial_conv_bwd_rows(ptr(pts), ptr(feats), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk),
                                                  ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                                  ptr(og), n, m, e, fin, batchSize, radius, int(scaleInv), int(avg),
                                                  int(bf16), ptr(plan.row_start), plan.vrow, plan.vcode,
                                                  plan.slice_off, plan.vpos_row, plan.rec, plan.other,
                                                  ptr(fg), ptr(scratch), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(dw3),
                                                  ptr(db3), ptr(feat_index), ptr(ws), ws.numel(), stream_handle()),
                  "spatial_conv_grad(rows)")
            if feat_index is not None:  # the gradient rows already lie in the order the features arrived in
                sort_index = None
            return (None, _unsort_grad(sort_index, fg), None, None, None, None, None, None, None, dw1, db1, dw2.view(ws2), db2.view(bs2), dw3.view(ws3), db3.view(bs3),
                    None, None, None, None, None, None, None, None, None)
        ws = _ws(lib.mccnn_spatial_conv_bwd_workspace_bytes(n, m, e, fin, numOutFeatures, int(combin)), pts.device)
        start_t = perm_t = None
        if not combin and e > 0:
            start_t, perm_t, _ = _transposed_neighbors(packed_obj, n)
        elif combin and 2 <= fin <= 4 and e > 0 and getattr(packed_obj, "_mccnn_transposed", None) is not None:
            # the transposed list exists already (prefetched with the geometry): the feature gradient is then gathered
            # through it instead of added with float atomics (deterministic, and cheaper than the atomics)
            start_t, perm_t, _ = _transposed_neighbors(packed_obj, n)
        if bf16:
            check(lib.mccnn_spatial_conv_bwd_bf16(ptr(pts), ptr(feats), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk),
                                                  ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                                  ptr(og), n, m, e, fin, batchSize, radius, int(scaleInv), int(avg),
                                                  ptr(start_t), ptr(perm_t), ptr(fg), ptr(dw1), ptr(db1), ptr(dw2),
                                                  ptr(db2), ptr(dw3), ptr(db3), ptr(ws), ws.numel(), stream_handle()),
                  "spatial_conv_grad(bf16)")
            return (None, _unsort_grad(sort_index, fg), None, None, None, None, None, None, None, dw1, db1, dw2.view(ws2), db2.view(bs2), dw3.view(ws3), db3.view(bs3),
                    None, None, None, None, None, None, None, None, None)
        check(lib.mccnn_spatial_conv_bwd(ptr(pts), ptr(feats), ptr(bids), ptr(pdfs), ptr(smp), ptr(st), ptr(pk),
                                         ptr(mn), ptr(mx), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                                         ptr(og), n, m, e, fin, numOutFeatures, int(combin), batchSize, radius,
                                         int(scaleInv), int(avg), ptr(ctx.state), ptr(start_t), ptr(perm_t), ptr(fg),
                                         ptr(dw1),
                                         ptr(db1), ptr(dw2), ptr(db2),
                                         ptr(dw3), ptr(db3), ptr(ws), ws.numel(), stream_handle()),
              "spatial_conv_grad")
        return (None, _unsort_grad(sort_index, fg), None, None, None, None, None, None, None, dw1, db1, dw2.view(ws2), db2.view(bs2), dw3.view(ws3), db3.view(bs3),
                None, None, None, None, None, None, None, None, None)


def _unsort_grad(idx, fg):
    """Feature gradient back in the order the features arrived in (sortIndex: in[i] = sorted[index_new_pos[i]])."""
    return fg if idx is None else _gather_rows(fg, idx, idx.shape[0])


def spatial_conv(inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs, aabbMin, aabbMax,
                 weights1, weights2, weightsOut, biases1, biases2, biasesOut, numOutFeatures, combin, batchSize, radius,
                 scaleInv, avg, sortIndex=None, _trusted=False, featIndex=None):
    """SpatialConv (MCConvModuleSrc:70-81). Note the reference's argument order (weights first, then biases);
    the op itself takes (w1, b1, w2, b2, w3, b3). sortIndex (extension): inFeatures are the rows of the UNSORTED points and
    sortIndex the grid's index_new_pos -- sort_features(inFeatures, sortIndex) happens inside this op. featIndex (with
    sortIndex): the inverse permutation (sorted row -> unsorted row, build_grid's fifth output); with it the row kernels of a
    small level read the unsorted rows in place."""
    return _SpatialConv.apply(inPts, inFeatures, inBatchIds, inPDFs, inSamplePts, neighStartIndexs, packedNeighs,
                              aabbMin, aabbMax, weights1, biases1, weights2, biases2, weightsOut, biasesOut,
                              numOutFeatures, combin, batchSize, radius, scaleInv, avg, sortIndex, _trusted, featIndex)
