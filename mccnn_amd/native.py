"""Native step executor (csrc/exec.hip, include/mccnn.h "NATIVE STEP EXECUTOR"): what
ConvolutionBuilder.create_convolution does around the kernels -- grid, neighbour list, PDFs, feature sort, kernel
choice, row plans -- behind one library call per convolution GEOMETRY and one per layer and direction.

A step of a network is bound by the host issuing its launches: ~12 us of Python per launch against 3 us for
the launch itself. This module is the thin remainder: it owns the device buffers the library asks for (the library
never allocates), the pinned word the edge total arrives in, and the autograd node of a layer.
"""
import os
import ctypes as C
import threading
import weakref

import torch

from . import _env, _lib
from ._lib import check, ptr, stream_handle

def _load_ext():
    """mccnn_amd/lib/_mccnn_torch.so (csrc/torch_ext.cpp): the same calls as below from C++, with the autograd node on the
    C++ side -- a convolution then costs the host one Python -> C++ call forward and none backward. MCCNN_TORCH_EXT=0 (or a
    tree without the built module) keeps the ctypes form."""
    import importlib.util
    import os
    from ._env import flag
    if not flag("TORCH_EXT"):
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_mccnn_torch.so")
    if not os.path.exists(path):
        return None
    _lib.load()  # libmccnn_hip.so first: the module links against it (a missing HIP library raises: no fallback for that)
    try:
        spec = importlib.util.spec_from_file_location("_mccnn_torch", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    except (ImportError, OSError) as err:  # built against another torch: the ctypes binding of the SAME library takes over
        import sys
        print("mccnn_amd.native: lib/_mccnn_torch.so does not load (%s); using the ctypes binding of libmccnn_hip.so" % err,
              file=sys.stderr)
        return None
    return mod


_EXT = _load_ext()
if _EXT is not None:
    # The extension's helper threads hold tensors that have Python objects; dropping the last reference to one takes the
    # GIL, and a non-main thread that asks for it while the interpreter finalises is ended by Python with a forced unwind
    # (std::terminate: "terminate called without an active exception" at the end of a script that stops with prefetched
    # work in flight). Before finalisation starts the queued jobs are run to the end and the threads joined.
    import atexit
    atexit.register(_EXT.shutdown_helpers)

E_CAPACITY = -6
NEED_PLAN_FWD, NEED_PLAN_TR, NEED_TLIST, NEED_RECORDS = 1, 2, 4, 8

_TLS = threading.local()
_EDGE_GUESS = {}   # (device, n, m, radius, B, scaleInv) -> capacity to try first
_EDGE_RATIO = {}   # (device, radius, scaleInv) -> edges per centre of the last search with this radius


def _slot():
    """A pinned int32 the count pass stores the edge total into (device-accessible host memory: no copy is enqueued);
    pooled per host thread."""
    _poll_parked()
    pool = getattr(_TLS, "slots", None)
    if pool is None:
        pool = _TLS.slots = []
    if pool:
        return pool.pop()
    return torch.empty(1, dtype=torch.int32).pin_memory()


def _release_slot(t):
    pool = getattr(_TLS, "slots", None)
    if pool is not None and len(pool) < 64:
        pool.append(t)


_PARKED = []   # (slot, event): words of geometries dropped before their total arrived -- the count pass may still write them
_PARKED_LOCK = threading.Lock()   # Geometry.__del__ parks from any thread, and from GC runs inside _poll_parked's own loop


def _park_slot(t):
    """Keeps a pinned word alive (neither pooled nor returned to the host allocator) until everything the current stream
    holds -- the build that writes it among it -- has retired."""
    try:
        ev = torch.cuda.Event()
        ev.record()
    except Exception:   # interpreter shutdown, no device: the word simply stays referenced
        ev = None
    _PARKED.append((t, ev))   # (list.append is atomic; _poll_parked never overwrites the list, see there)


def _poll_parked():
    if not _PARKED:
        return
    # Take the entries present NOW out of the list in one atomic step and work on the private copy: an entry parked while
    # this loop runs (another thread, or a GC run triggered by the allocations below calling Geometry.__del__) stays in
    # _PARKED and is never dropped -- `_PARKED[:] = keep` used to discard it and hand a word the count pass could still
    # write back to the host allocator. Non-blocking: a second caller simply skips its poll.
    if not _PARKED_LOCK.acquire(False):
        return
    try:
        k = len(_PARKED)
        items = _PARKED[:k]
        del _PARKED[:k]
        keep = []
        for t, ev in items:
            try:
                done = int(t[0]) >= 0 or ev is None or ev.query()
            except Exception:
                done = False
            if done:
                _release_slot(t)
            else:
                keep.append((t, ev))
        _PARKED.extend(keep)
    finally:
        _PARKED_LOCK.release()


def _ws(nbytes, device):
    from .MCConvModule import _ws as pool_ws
    return pool_ws(nbytes, device)


_ECAP_SCALE = _env.debug("ecap_scale", 1.0)   # (debugging: generous / starved capacity guesses)


def _capacity_guess(gkey, m):
    if _ECAP_SCALE != 1.0:
        return int(_capacity_guess_(gkey, m) * _ECAP_SCALE) + 1024
    return _capacity_guess_(gkey, m)


def _capacity_guess_(gkey, m):
    g = _EDGE_GUESS.get(gkey, 0)
    if g <= 0:
        ratio = _EDGE_RATIO.get((gkey[0], gkey[3], gkey[5]), 0.0)
        g = int(ratio * m * 1.25) + 1024 if ratio > 0.0 else 48 * m + 1024  # first search of a radius: a plain guess
    return g


def _remember(gkey, m, e):
    if len(_EDGE_GUESS) > 256:
        _EDGE_GUESS.clear()
    _EDGE_GUESS[gkey] = e + e // 16 + 64
    if m > 0:
        _EDGE_RATIO[(gkey[0], gkey[3], gkey[5])] = e / float(m)


class Geometry:
    """One convolution geometry: the grid of the input level at the convolution radius, the neighbour list of the
    output level's points in it and its PDFs, in ONE device buffer (mccnn_geometry_t). Tensor views of its arrays are
    made on demand (grid(), neighbors(), pdfs()) -- the layers themselves only hand the handle to the library."""

    def __init__(self):
        self.core = None        # the C++ object of the torch extension (owns handle / buffers), when that is in use
        self.handle = None
        self.buf = None
        self.slot = None
        self.keep = None        # the input tensors the library borrowed pointers of
        self.attached = []      # plan / list buffers handed to the library
        self.grid_owner = None
        self.n = self.m = self.nc = self.B = self.e_cap = 0
        self.e = -1
        self.gkey = None
        self.args = None
        self.uses = 0           # layers convolved over this geometry so far (the builder counts)

    def __del__(self):
        h, self.handle = self.handle, None
        if h is not None:
            try:
                _lib.load().mccnn_geometry_destroy(h)
            except Exception:
                pass
        if self.slot is not None:
            if self.e >= 0:
                _release_slot(self.slot)
            else:   # the total never arrived here: the count pass may still write the word (see _park_slot)
                try:
                    _park_slot(self.slot)
                except Exception:
                    pass

    # ------------------------------------------------------------------ sizes
    def edges(self):
        """E (waits for the count pass of the build)."""
        if self.e < 0:
            e = self.core.edges(-1) if self.core is not None else _lib.load().mccnn_geometry_edges(self.handle, -1)
            if e < 0:
                raise _lib.MCCNNError("geometry: edge total not available")
            self.e = e
            _remember(self.gkey, self.m, e)
            if e > self.e_cap:
                self._rebuild(e + e // 16 + 64)
        return self.e

    def _rebuild(self, capacity):
        """The neighbour list was longer than the guess: exact repeat (first batch of a shape)."""
        inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF = self.args
        _build_into(self, inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF, capacity,
                    self.grid_owner)
        e = self.core.edges(-1) if self.core is not None else _lib.load().mccnn_geometry_edges(self.handle, -1)
        if e < 0 or e > self.e_cap:
            raise _lib.MCCNNError("geometry: rebuilt list still does not fit (%d > %d)" % (e, self.e_cap))
        self.e = e

    def prebuild(self, what, avg, side, like):
        """Row plans / transposed list ahead of the layers that need them, on a side stream (torch extension only);
        what: NEED_* mask. Waits for the edge total."""
        if self.core is not None:
            self.edges()
            self.core.prebuild(int(what), bool(avg), int(side), like)

    @property
    def have(self):
        """Mask of the pieces attached so far (NEED_*)."""
        return int(self.core.have) if self.core is not None else 0

    def prebuild_async(self, what, avg):
        """Pieces of a geometry whose build has just been queued on a side stream: buffers allocated now (bounds over
        the capacity), attached and issued by the extension's helper thread once the edge total has arrived -- the
        calling thread does not wait."""
        if self.core is not None:
            _EXT.prebuild_async(self.core, int(what), bool(avg))

    # ------------------------------------------------------------------ views (tests, the builder's cache tuples)
    def _info(self):
        if self.core is not None:
            return self.core.info()
        out = (C.c_longlong * 16)()
        check(_lib.load().mccnn_geometry_info(self.handle, out), "geometry_info")
        return list(out)

    def _view(self, owner_buf, addr, nbytes, dtype, shape):
        off = addr - owner_buf.data_ptr()
        return owner_buf[off:off + nbytes].view(dtype).view(shape)

    def _join(self):
        """A build issued on a side stream: whoever READS the arrays through tensor views does so on the current stream."""
        for g in (self, self.grid_owner):
            if g is not None and g.core is not None and g.buf is not None:
                g.core.join(g.buf)

    def grid(self):
        """(sortPts [n,3], sortBatchs [n,1], cellIndexs [B,nc,nc,nc,2], index_new_pos [n], inverse [n])"""
        self._join()
        i = self._info()
        ob = (self.grid_owner or self).buf
        n, nc, B = self.n, self.nc, self.B
        return (self._view(ob, i[0], n * 12, torch.float32, (n, 3)), self._view(ob, i[1], n * 4, torch.int32, (n, 1)),
                self._view(ob, i[2], B * nc ** 3 * 8, torch.int32, (B, nc, nc, nc, 2)),
                self._view(ob, i[3], n * 4, torch.int32, (n,)), self._view(ob, i[4], n * 4, torch.int32, (n,)))

    def neighbors(self):
        """(startIndexs [m,1], packedNeighs [E,2])"""
        e = self.edges()
        self._join()
        i = self._info()
        return (self._view(self.buf, i[5], self.m * 4, torch.int32, (self.m, 1)),
                self._view(self.buf, i[6], e * 8, torch.int32, (e, 2)))

    def pdfs(self):
        e = self.edges()
        self._join()
        i = self._info()
        return self._view(self.buf, i[7], e * 4, torch.float32, (e, 1))


def _build_into(g, inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF, capacity, grid_from,
                side=-1, fork=False, background=False, after=None):
    n, m = inPts.shape[0], centres.shape[0]
    if _EXT is not None:
        uses = g.core.uses if g.core is not None else 0
        g.core = _EXT.build_geometry(inPts, inBids, centres, cbids, mn, mx, B, nc, float(radius), bool(scaleInv), float(window),
                                     bool(usePDF), capacity, grid_from.core if grid_from is not None else None, side, fork,
                                     background, after)
        g.core.uses = uses
        g.buf = g.core.buf
        g.grid_owner = grid_from
        g.n, g.m, g.nc, g.B, g.e_cap, g.e = n, m, nc, B, capacity, -1
        g.args = (inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF)
        return
    lib = _lib.load()
    dev = inPts.device
    with_grid = 0 if grid_from is not None else 1
    nbytes = lib.mccnn_geometry_bytes(n, m, B, nc, capacity, with_grid)
    if nbytes == 0:
        raise _lib.MCCNNError("geometry: batch_size * num_cells^3 does not fit 32-bit keys")
    if g.handle is None:
        g.handle = lib.mccnn_geometry_create()
        if not g.handle:
            raise MemoryError("mccnn_geometry_create")
    if g.slot is None:
        g.slot = _slot()
    g.buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g.attached = []
    g.keep = (inPts, inBids, centres, cbids, mn, mx)
    g.grid_owner = grid_from
    g.n, g.m, g.nc, g.B, g.e_cap, g.e = n, m, nc, B, capacity, -1
    g.args = (inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF)
    check(lib.mccnn_geometry_build(g.handle, ptr(inPts), ptr(inBids), n, ptr(centres), ptr(cbids), m, ptr(mn), ptr(mx), B, nc,
                                   float(radius), int(bool(scaleInv)), float(window), int(bool(usePDF)), capacity,
                                   grid_from.handle if grid_from is not None else None, g.buf.data_ptr(), nbytes,
                                   g.slot.data_ptr(), stream_handle()), "geometry_build")


def begin_batch():
    """Geometries requested until end_batch() that go to a side stream are issued as ONE batch -- one launch per kernel kind
    over all of them (mccnn_geometry_build_batch) -- on one side stream (torch extension only; otherwise a no-op)."""
    if _EXT is not None:
        _EXT.begin_geometry_batch()


def end_batch():
    if _EXT is not None:
        _EXT.end_geometry_batch()


def side_streams_available():
    """Geometry builds on side streams need the torch extension (events and streams live on its side)."""
    return _EXT is not None


def build_geometry(inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF, grid_from=None,
                   side=-1, fork=False, background=False, after=None):
    """Enqueues grid + search + KDE of one convolution geometry; no host wait. nc: cells per axis
    (MCConvModule._num_cells). grid_from: a Geometry over the same points / radius whose grid is shared. side >= 0
    (torch extension only): the build runs on side stream `side` -- behind everything the current stream holds at the
    first call that says fork=True -- and the first layer that uses the geometry orders its stream behind it.
    after: the future of the ADOPTED prefetched hierarchy that every input of this build belongs to; a side-stream build
    then waits for that hierarchy only and takes its memory from its own stream's pool (torch_ext.cpp, Geo::own_pool)."""
    n, m = inPts.shape[0], centres.shape[0]
    gkey = (inPts.device.index, n, m, float(radius), int(B), bool(scaleInv))
    g = Geometry()
    g.gkey = gkey
    if grid_from is not None and grid_from.grid_owner is not None:
        grid_from = grid_from.grid_owner
    _build_into(g, inPts, inBids, centres, cbids, mn, mx, B, nc, radius, scaleInv, window, usePDF,
                _capacity_guess(gkey, m), grid_from, side if _EXT is not None else -1, fork, background,
                after if _EXT is not None else None)
    return g


_I0 = C.c_int
_LL = C.c_longlong


def _prepare(geo, feats, fin, fout, combin, bf16, backward, flags):
    """-> (ws bytes, saved bytes); attaches whatever the geometry lacks for this call."""
    lib = _lib.load()
    mask, edges = _I0(0), _I0(0)
    need = (_LL * 4)()
    wsb, svb = _LL(0), _LL(0)
    rc = lib.mccnn_conv_prepare(geo.handle, feats.data_ptr(), fin, fout, combin, bf16, backward, flags, C.byref(mask), need,
                                C.byref(wsb), C.byref(svb), C.byref(edges))
    if rc == E_CAPACITY:
        geo.edges()  # rebuilds with the exact size
        rc = lib.mccnn_conv_prepare(geo.handle, feats.data_ptr(), fin, fout, combin, bf16, backward, flags, C.byref(mask),
                                    need, C.byref(wsb), C.byref(svb), C.byref(edges))
    check(rc, "conv_prepare")
    if geo.e < 0:
        geo.e = edges.value
        _remember(geo.gkey, geo.m, geo.e)
    if mask.value:
        for k, bit in enumerate((NEED_PLAN_FWD, NEED_PLAN_TR, NEED_TLIST, NEED_RECORDS)):
            if mask.value & bit:
                t = torch.empty(max(int(need[k]), 256), dtype=torch.uint8, device=feats.device)
                check(lib.mccnn_geometry_attach(geo.handle, bit, t.data_ptr(), t.numel()), "geometry_attach")
                geo.attached.append(t)
    return wsb.value, svb.value


E_WORKSPACE = -4
_SIZES = {}   # (direction, fin, fout, combin, bf16, n, m) -> (scratch bytes, saved bytes) of the last call of this shape


class _Conv(torch.autograd.Function):
    """SpatialConv with sort_features folded in (MCConvModuleSrc:35-45,70-81) over a native Geometry. The calls are made
    OPTIMISTICALLY with the buffer sizes of the last call of the same shape; only when the library says something is
    missing (first batch of a shape, a longer list, a plan to attach) mccnn_conv_prepare is asked and the call repeated --
    nothing has been launched by then."""

    @staticmethod
    def forward(ctx, feats, w1, b1, w2, b2, w3, b3, geo, fout, combin, avg, deterministic):
        lib = _lib.load()
        fin = feats.shape[1]
        bf16 = 1 if feats.dtype == torch.bfloat16 else 0
        need_grad = feats.requires_grad or w1.requires_grad or w3.requires_grad or w2.requires_grad or b1.requires_grad \
            or b2.requires_grad or b3.requires_grad
        flags = (1 if need_grad else 0) | (2 if deterministic else 0)
        combin = 1 if combin else 0
        avg = 1 if avg else 0
        dev = feats.device
        out = torch.empty((geo.m, fout if combin else fin), dtype=feats.dtype, device=dev)
        skey = (flags & 1, fin, fout, combin, bf16, geo.n, geo.m)
        wsb, svb = _SIZES.get(skey, (0, -1))
        if svb < 0:
            wsb, svb = _prepare(geo, feats, fin, fout, combin, bf16, 0, flags)
            svb += svb // 16   # a little head room: the lists of a shape vary from batch to batch
        tried = False
        while True:
            saved = torch.empty(svb, dtype=torch.uint8, device=dev) if svb else None
            ws = _ws(wsb, dev)
            rc = lib.mccnn_conv_forward(geo.handle, feats.data_ptr(), fin, fout, combin, avg, bf16, flags, w1.data_ptr(),
                                        b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(), b3.data_ptr(),
                                        out.data_ptr(), ptr(saved), svb, ws.data_ptr(), ws.numel(), stream_handle())
            if rc == 0:
                break
            if rc not in (E_WORKSPACE, E_CAPACITY):
                check(rc, "conv_forward")
            w2_, s2_ = _prepare(geo, feats, fin, fout, combin, bf16, 0, flags)
            if s2_ <= svb and rc == E_WORKSPACE and ws.numel() >= w2_ and tried:
                check(rc, "conv_forward")  # the library asked twice for what it was given
            tried = True
            wsb, svb = w2_, s2_ + s2_ // 16
        if geo.e < 0:
            geo.e = lib.mccnn_geometry_edges(geo.handle, 0)
            _remember(geo.gkey, geo.m, geo.e)
        if len(_SIZES) > 1024:
            _SIZES.clear()
        _SIZES[skey] = (wsb, svb)
        if need_grad:
            ctx.save_for_backward(feats, w1, b1, w2, b2, w3, b3)
            ctx.saved_buf = saved
            ctx.geo = geo
            ctx.attrs = (fin, fout, combin, avg, bf16, flags)
        return out

    @staticmethod
    def backward(ctx, outGrad):
        feats, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        geo, saved = ctx.geo, ctx.saved_buf
        fin, fout, combin, avg, bf16, flags = ctx.attrs
        if geo.uses > 1:
            # several layers share this neighbour list: its transposed form is built once and the feature gradient of
            # combin layers with 2..4 input features is gathered through it in a fixed order (bit-reproducible) instead
            # of added with float atomics; a bare single call keeps the atomics (the list would cost more than they do)
            flags |= 2
        lib = _lib.load()
        og = outGrad if outGrad.is_contiguous() else outGrad.contiguous()
        if og.dtype != feats.dtype:
            og = og.to(feats.dtype)
        dev = feats.device
        fg = torch.empty_like(feats)
        # the six MLP gradients: consecutive slices of ONE buffer in the order the builder creates the variables (a
        # data-parallel step all-reduces that buffer as it is, dist.GradBucket)
        n1, n2, n3, n4, n5, n6 = w1.numel(), b1.numel(), w2.numel(), b2.numel(), w3.numel(), b3.numel()
        gflat = torch.empty(n1 + n2 + n3 + n4 + n5 + n6, dtype=torch.float32, device=dev)
        dw1, db1, dw2, db2, dw3, db3 = gflat.split((n1, n2, n3, n4, n5, n6))
        base = gflat.data_ptr()
        skey = (2, flags & 2, fin, fout, combin, bf16, geo.n, geo.m)   # (deterministic and atomic backward passes differ in scratch)
        wsb = _SIZES.get(skey, (-1, 0))[0]
        if wsb < 0:
            wsb, _ = _prepare(geo, feats, fin, fout, combin, bf16, 1, flags)
        svn = saved.numel() if saved is not None else 0
        tried = False
        while True:
            ws = _ws(wsb, dev)
            rc = lib.mccnn_conv_backward(geo.handle, feats.data_ptr(), ptr(saved), svn, og.data_ptr(), fin, fout, combin, avg,
                                         bf16, flags, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), w3.data_ptr(),
                                         b3.data_ptr(), fg.data_ptr(), base, base + 4 * n1, base + 4 * (n1 + n2),
                                         base + 4 * (n1 + n2 + n3), base + 4 * (n1 + n2 + n3 + n4),
                                         base + 4 * (n1 + n2 + n3 + n4 + n5), ws.data_ptr(), ws.numel(), stream_handle())
            if rc == 0:
                break
            if rc not in (E_WORKSPACE, E_CAPACITY):
                check(rc, "conv_backward")
            w2_, _ = _prepare(geo, feats, fin, fout, combin, bf16, 1, flags)
            if tried and ws.numel() >= w2_:
                check(rc, "conv_backward")
            tried = True
            wsb = w2_
        _SIZES[skey] = (wsb, 0)
        return (fg, dw1.view_as(w1), db1, dw2.view_as(w2), db2.view_as(b2), dw3.view_as(w3), db3.view_as(b3),
                None, None, None, None, None)


def conv(geo, feats, w1, b1, w2, b2, w3, b3, numOutFeatures, combin, avg, deterministic=False):
    """One MC convolution over `geo`: feats are the rows of the UNSORTED input points ([n, Fin] f32, or bf16 for
    depth-wise layers); the kernel-MLP tensors in any shape over the reference's flat layout."""
    core = geo.core
    if core is not None:
        core.uses = geo.uses
        try:
            out = _EXT.conv(core, feats, w1, b1, w2, b2, w3, b3, numOutFeatures, bool(combin), bool(avg), bool(deterministic))
        except _EXT.CapacityError:
            geo.edges()   # the list did not fit the guess (first batch of a shape): exact rebuild, then the layer
            geo.core.uses = geo.uses
            out = _EXT.conv(geo.core, feats, w1, b1, w2, b2, w3, b3, numOutFeatures, bool(combin), bool(avg), bool(deterministic))
        if geo.e < 0:
            geo.e = geo.core.e
            _remember(geo.gkey, geo.m, geo.e)
        return out
    return _Conv.apply(feats, w1, b1, w2, b2, w3, b3, geo, numOutFeatures, combin, avg, deterministic)
