"""PointHierarchy / ConvolutionBuilder -- the builder API of the reference's utils/MCConvBuilder.py
on top of the HIP op surface (mccnn_amd.MCConvModule).

Same constructor / create_convolution signatures, cache keys (MCConvBuilder.py:203-238) and variable
names / shapes (MCConvBuilder.py:394-419). The reference builds a TF1 graph once; here the builder
is called every forward pass (eager), so:
  * the kernel-MLP variables live in the builder (`ConvolutionBuilder.parameters()`), are created on
    first use of a `convName` and re-used afterwards (tf.get_variable semantics);
  * `reset()` (MCConvBuilder.py:241) drops the per-forward caches (grids, neighbours, pdfs) -- call it
    (or build a new PointHierarchy) at the start of each forward pass.
"""
import math

import os
import sys
import time
import torch

from . import _env


# `from . import x` inside a function is ~1 us per execution (importlib's _handle_fromlist): 60-70 of them per step of a
# 17-layer graph. The two modules are imported once, on first use (not at import time: they load the libraries).
_NATIVE = None
_HIPOPS = None


def _native_mod():
    global _NATIVE
    if _NATIVE is None:
        from . import native
        _NATIVE = native
    return _NATIVE


def _hip_ops_mod():
    global _HIPOPS
    if _HIPOPS is None:
        from . import MCConvModule
        _HIPOPS = MCConvModule
    return _HIPOPS

from .MCConvModule import (compute_aabb, sort_points_step1, sort_points_step2, sort_features, sort_features_back,
                           compute_pdf, poisson_sampling, get_sampled_features, spatial_conv, get_block_size,
                           transform_indexs, find_neighbors)

_OP_NAMES = ("compute_aabb", "sort_points_step1", "sort_points_step2", "sort_features", "sort_features_back",
             "compute_pdf", "poisson_sampling", "get_sampled_features", "spatial_conv", "get_block_size",
             "transform_indexs", "find_neighbors")


class _Ops:
    """The twelve names the reference builder imports from MCConvModule (MCConvBuilder.py:20-21). By default they are
    this module's own imports of the HIP op surface, looked up at call time; a caller may hand the builder classes
    another object with the same names (`ops=`) -- the parity tests run the identical graph through the CPU checker."""

    def __init__(self, ops=None):
        self._ops = ops

    def __getattr__(self, name):
        if name not in _OP_NAMES:
            raise AttributeError(name)
        return getattr(self._ops, name) if self._ops is not None else globals()[name]


_VERBOSE = False
#: build all levels of a PointHierarchy with ONE host read-back (MCConvModule.point_hierarchy_levels) instead of one per
#: level; results are identical. Only with the HIP op surface on CUDA tensors.
FUSED_HIERARCHY = True


def _log(msg):
    if _VERBOSE:
        print(msg)


_GEO_TRACE = _env.debug("geo_trace", False)


class _LazyEntry:
    """A cache entry of the native path (mccnn_amd.native.Geometry): behaves like the tuple / tensor the reference's
    cache holds (MCConvBuilder.py:349-391), but the tensor views are only made when somebody looks -- the layers
    themselves hand the geometry's handle to the library."""
    __slots__ = ("geo", "_val", "_make")

    def __init__(self, geo, make):
        self.geo, self._val, self._make = geo, None, make

    def value(self):
        if self._val is None:
            self._val = self._make()
        return self._val

    def __getitem__(self, i):
        return self.value()[i]

    def __iter__(self):
        return iter(self.value())

    def __len__(self):
        return len(self.value())

    def __getattr__(self, name):  # (cachePDFs_ entries are tensors: .shape, .data_ptr() ...)
        return getattr(self.value(), name)


_GEO_PREFETCH_MIN = _env.debug("geo_prefetch_min", 5)
_PLAN_PREFETCH_MAX_E = _env.debug("plan_prefetch_max_e", 10 ** 12)


class _PrefetchedHierarchy:
    """Handle of PointHierarchy.prefetch(): the future of the extension and what it was requested for."""

    def __init__(self, future, points, batchIds, radiusList, batchSize, relativeRadius, features=None):
        self.future, self.points, self.batchIds = future, points, batchIds
        # (feature rows handed to prefetch(): the levels' rows were gathered with the hierarchy when these are still the
        # same tensor, unmodified)
        self.features, self.featuresVersion = features, (features._version if features is not None else None)
        self.radiusList, self.batchSize, self.relativeRadius = [float(r) for r in radiusList], int(batchSize), bool(relativeRadius)
        self.version = (points._version, batchIds._version)

    def done(self):
        return self.future.done()

    def check(self, points, batchIds, radiusList, batchSize, relativeRadius):
        same = (points is self.points or (points.data_ptr() == self.points.data_ptr() and points.shape == self.points.shape)) \
            and (batchIds is self.batchIds or (batchIds.data_ptr() == self.batchIds.data_ptr()
                                              and batchIds.shape == self.batchIds.shape)) \
            and (self.points._version, self.batchIds._version) == self.version \
            and [float(r) for r in radiusList] == self.radiusList and int(batchSize) == self.batchSize \
            and bool(relativeRadius) == self.relativeRadius
        if not same:
            from .MCConvModule import InvalidArgumentError
            raise InvalidArgumentError("PointHierarchy: `prefetched` was requested for other points, batch ids, radii, "
                                       "batch size or radius mode (or the tensors were modified since)")


def _record_event(device_index=None):
    """A torch.cuda.Event recorded on the current stream of `device_index` (default: the current device). With the index
    spelled out torch.cuda.current_stream() skips its device look-up chain (~9 us of the ~12 an Event().record() costs;
    a step records three)."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(torch.cuda.current_device() if device_index is None else device_index))
    return ev


# The op-by-op prefetch (prefetch_geometry() without the native executor) orders its side stream behind two events: the one a
# PointHierarchy records when its last level is enqueued and the one of the last reset(). An event record is ~5 us of host time and
# ~4 us of the calling QUEUE's -- two or three per step of a loop that never takes that path (the native executor forks its side
# streams itself). They are recorded from the first op-by-op prefetch on; that first call waits for the calling stream instead.
_SIDE_EVENTS = [False]

_CUDA_OK = []   # torch.cuda.is_available(), asked once (reset() runs every step)


def _cuda_ok():
    if not _CUDA_OK:
        _CUDA_OK.append(bool(torch.cuda.is_available()))
    return _CUDA_OK[0]


class _PlainState:
    """Attributes whose names end in '_' are plain state (caches, lists, tensors the class merely holds): they bypass
    torch.nn.Module.__setattr__, whose parameter / buffer / sub-module bookkeeping costs ~5 us per assignment -- a reset()
    and a hierarchy make thirty of them per step."""

    def __setattr__(self, name, value):
        if name.endswith("_") and not isinstance(value, (torch.nn.Parameter, torch.nn.Module)):
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)


class PointHierarchy(_PlainState, torch.nn.Module):
    """Point hierarchy built by successive Poisson-disk sampling (MCConvBuilder.py:24-131).

    Attributes (same names as the reference): points_, features_, batchIds_, sampledIndexs_,
    radiusList_, batchSize_, relativeRadius_, hierarchyName_, aabbMin_, aabbMax_. A torch.nn.Module without parameters
    (the reference's class holds none either); readyEvent_ (extension) is recorded on the stream that built the
    hierarchy once all levels are enqueued -- ConvolutionBuilder.prefetch_geometry() orders its side stream behind it.
    """

    @staticmethod
    def prefetch(inPoints, inBatchIds, radiusList, batchSize=32, relativeRadius=True, after=None, features=None):
        """Extension (no counterpart in the reference): starts the geometry of a hierarchy -- the boxes and every level's
        Poisson-disk samples, which depend on the points only -- on a stream of its own, issued by a helper thread, and
        returns a handle for `PointHierarchy(..., prefetched=handle)`. In a training loop: request the hierarchy of batch
        k + 1 right after the one of batch k has been constructed; its chain of small dependent kernels and its two
        read-backs then run under the convolutions of batch k instead of in front of those of batch k + 1, and the calling
        thread never waits for the device. The build starts behind what the calling stream holds at the moment of the call
        (the upload of the batch) -- or, for a loader that uploads on a stream of its own, behind the torch.cuda.Event it
        recorded there (after=event), or at once (after=True: points and batch ids are complete); the calling stream may
        hold a whole step of convolutions by then, which such a hierarchy no longer queues behind. Returns None when there
        is nothing to run ahead (host tensors, no extension): the constructor then builds inline, as without the argument.
        features (optional): the feature rows the constructor will be given for level 0. Rows that carry no gradient (a
        network's input features) are then gathered for every level together with the hierarchy, on its stream, instead of
        by one launch per level on the calling thread when the hierarchy is adopted; the constructor uses them when it is
        handed the very same, unmodified tensor, and gathers as usual otherwise."""
        _M = _hip_ops_mod()
        if not FUSED_HIERARCHY or not poisson_sampling.__module__.endswith("MCConvModule"):
            return None
        fut = _M.point_hierarchy_prefetch(inPoints, inBatchIds, list(radiusList), batchSize, relativeRadius, after, features)
        if fut is None:
            return None
        return _PrefetchedHierarchy(fut, inPoints, inBatchIds, radiusList, batchSize, relativeRadius, features)

    def __init__(self, inPoints, inFeatures, inBatchIds, radiusList, hierarchyName="Point_Hierarchy", batchSize=32,
                 relativeRadius=True, aabbReduceGroup=None, ops=None, prefetched=None):
        """prefetched (extension): the handle PointHierarchy.prefetch() returned for these very points, batch ids, radii,
        batch size and radius mode.

        aabbReduceGroup (extension, data-parallel shards only): with relativeRadius=False the reference uses ONE box
        for the whole batch (aabb_gpu.cu:104-114). A shard that holds part of the batch passes its process group here
        (`True` = the default group) and the MIN/MAX all-reduce of the box runs between compute_aabb and the first
        sort, so every level of the sharded hierarchy -- cells, keys, Poisson samples -- equals the corresponding
        slice of the single-device hierarchy (mccnn_amd.dist)."""
        super().__init__()
        ops = _Ops(ops)
        # (plain state, written through __dict__: a hierarchy is built every step, see _PlainState)
        self.__dict__.update(readyEvent_=None, prefetchFuture_=None, points_=[inPoints], features_=[inFeatures],
                             batchIds_=[inBatchIds], sampledIndexs_=[], radiusList_=[0.0], batchSize_=batchSize,
                             relativeRadius_=relativeRadius, hierarchyName_=hierarchyName)

        if prefetched is not None:
            prefetched.check(inPoints, inBatchIds, radiusList, batchSize, relativeRadius)
            if aabbReduceGroup is None and ops._ops is None and self.__adopt_prefetched__(prefetched, inFeatures, ops):
                return
        aabbMin, aabbMax = ops.compute_aabb(inPoints, inBatchIds, batchSize, self.relativeRadius_)
        if aabbReduceGroup is not None and not self.relativeRadius_:
            from .dist import allreduce_aabb
            aabbMin, aabbMax = allreduce_aabb(aabbMin, aabbMax, None if aabbReduceGroup is True else aabbReduceGroup)
        self.aabbMin_ = aabbMin
        self.aabbMax_ = aabbMax
        _log("########## Point Hierarchy: %s (Rel: %s)" % (hierarchyName, relativeRadius))

        if FUSED_HIERARCHY and ops._ops is None and len(radiusList) > 0 and getattr(inPoints, "is_cuda", False) \
                and poisson_sampling.__module__.endswith("MCConvModule"):
            _M = _hip_ops_mod()
            fused = _M.point_hierarchy_levels(inPoints, inBatchIds, aabbMin, aabbMax, list(radiusList), batchSize,
                                              self.relativeRadius_)
            if fused is not None:
                currFeatures = inFeatures
                for (sampledPts, sampledBatchsIds, _sortedIdx, transformedIndexs), currRadius in zip(fused, radiusList):
                    # the level's feature rows: sortFeatures[sampledIndexs] == features[transformedIndexs]
                    # (MCConvBuilder.py:112-116), one differentiable gather now that the sizes are known
                    currFeatures = ops.get_sampled_features(transformedIndexs, currFeatures)
                    self.points_.append(sampledPts)
                    self.batchIds_.append(sampledBatchsIds)
                    self.features_.append(currFeatures)
                    self.sampledIndexs_.append(transformedIndexs)
                    self.radiusList_.append(currRadius)
                self.__mark_ready__()
                return

        currPts, currFeatures, currBatchIds = inPoints, inFeatures, inBatchIds
        for level, currRadius in enumerate(radiusList):
            _log("Level: %d | Poisson Disk Radius: %s" % (level + 1, currRadius))
            keys, indexs = ops.sort_points_step1(currPts, currBatchIds, self.aabbMin_, self.aabbMax_, self.batchSize_,
                                             currRadius, self.relativeRadius_)
            sortPts, sortBatchs, sortFeatures, cellIndexs = ops.sort_points_step2(
                currPts, currBatchIds, currFeatures, keys, indexs, self.aabbMin_, self.aabbMax_, self.batchSize_,
                currRadius, self.relativeRadius_)
            sampledPts, sampledBatchsIds, sampledIndexs = ops.poisson_sampling(
                sortPts, sortBatchs, cellIndexs, aabbMin, aabbMax, currRadius, batchSize, self.relativeRadius_)
            sampledFeatures = ops.get_sampled_features(sampledIndexs, sortFeatures)
            transformedIndexs = ops.transform_indexs(sampledIndexs, indexs)

            self.points_.append(sampledPts)
            self.batchIds_.append(sampledBatchsIds)
            self.features_.append(sampledFeatures)
            self.sampledIndexs_.append(transformedIndexs)
            self.radiusList_.append(currRadius)
            currPts, currBatchIds, currFeatures = sampledPts, sampledBatchsIds, sampledFeatures
        self.__mark_ready__()

    def __adopt_prefetched__(self, prefetched, inFeatures, ops):
        """The levels a helper thread built ahead (PointHierarchy.prefetch): the current stream is ordered behind them, the
        feature rows of every level are gathered now. False when the single-launch Poisson form gave up somewhere."""
        _M = _hip_ops_mod()
        aabbMin, aabbMax, extent, levels = prefetched.future.result()   # (its wait is counted by the extension: wait_ns)
        if not levels:
            return False
        # (geometry builds over this hierarchy on side streams wait for THIS, not for the calling stream: native.build_geometry)
        self.__dict__["prefetchFuture_"] = prefetched.future
        self.__dict__.update(aabbMin_=aabbMin, aabbMax_=aabbMax)
        if not self.relativeRadius_:
            _M._seed_num_cells(aabbMin, aabbMax, extent)
        _log("########## Point Hierarchy: %s (Rel: %s, prefetched)" % (self.hierarchyName_, self.relativeRadius_))
        currFeatures = inFeatures
        # the levels' feature rows came with the hierarchy when prefetch() was handed this very tensor (no gradient, unmodified)
        pf = prefetched.features
        gathered = (pf is not None and inFeatures is not None and not getattr(inFeatures, "requires_grad", True)
                    and (inFeatures is pf or (inFeatures.data_ptr() == pf.data_ptr() and inFeatures.shape == pf.shape
                                              and inFeatures.dtype == pf.dtype))
                    and pf._version == prefetched.featuresVersion)
        for lvl, currRadius in zip(levels, prefetched.radiusList):
            sampledPts, sampledBatchsIds, _sortedIdx, transformedIndexs = lvl[:4]
            if gathered and len(lvl) > 4:
                currFeatures = lvl[4]
            else:
                currFeatures = ops.get_sampled_features(transformedIndexs, currFeatures)
            self.points_.append(sampledPts)
            self.batchIds_.append(sampledBatchsIds)
            self.features_.append(currFeatures)
            self.sampledIndexs_.append(transformedIndexs)
            self.radiusList_.append(currRadius)
        self.__mark_ready__()
        return True

    def __mark_ready__(self):
        """Everything the hierarchy holds (points of every level, boxes) has been enqueued on the current stream."""
        if _SIDE_EVENTS[0] and getattr(self.points_[0], "is_cuda", False):
            self.__dict__["readyEvent_"] = _record_event(self.points_[0].device.index)


def _fan_avg_uniform_(t, fan_in, fan_out):
    """tf.contrib.layers.variance_scaling_initializer(factor=1.0, mode='FAN_AVG', uniform=True)
    (MCConvBuilder.py:394): U(-l, l), l = sqrt(3 * factor / ((fan_in + fan_out) / 2))."""
    limit = math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))
    with torch.no_grad():
        t.uniform_(-limit, limit)
    return t


class ConvolutionBuilder(_PlainState, torch.nn.Module):
    """Creates MC convolutions on point hierarchies, caching grids / neighbours / pdfs
    (MCConvBuilder.py:133-427). A torch.nn.Module: the kernel-MLP variables are registered parameters under the
    reference's names (`<convName>_weights`, `_biases`, `_weights2`, ... MCConvBuilder.py:407-419), created on first use
    of a convName (tf.get_variable semantics), so `parameters()`, `state_dict()`, `.to()`, optimisers and DDP wrappers
    work on the builder as on any module; `forward` is create_convolution."""

    def __init__(self, multiFeatureConvs=False, KDEWindow=0.25, relativeRadius=True, usePDF=True, useAVG=True,
                 decayLossCollection='weight_decay_loss', device=None, ops=None, fuseSort=None, native=None):
        super().__init__()
        self.ops_ = _Ops(ops)
        # extension: grids from the points alone (MCConvModule.build_grid), feature rows sorted inside the convolution's
        # node (spatial_conv(sortIndex=)) -- fewer op calls and graph nodes per convolution, same kernels and results
        self.fuseSort_ = _env.debug("fuse_sort", True) if fuseSort is None else bool(fuseSort)
        # extension: the whole geometry of a convolution (grid, search, KDE) in one library call and the layer itself in
        # one per direction (mccnn_amd.native, csrc/exec.hip) -- a step is bound by the host's work per launch
        self.native_ = _env.flag("NATIVE") if native is None else bool(native)
        self.cacheGrids_ = {}
        self.cacheNeighs_ = {}
        self.cachePDFs_ = {}
        self.cacheGeo_ = {}         # keyPDF -> native.Geometry; keyGrid -> the Geometry that owns the grid
        self.cacheGeoGrid_ = {}
        self.layers_ = {}           # convName -> (spec, variables): the fast path of a repeated create_convolution
        # learned prefetch (native path): the geometries a step asked for, in order; the next step builds them ALL at its
        # first create_convolution, round-robin on side streams -- they depend on the points only, their chains of small
        # kernels run side by side and under the first layers instead of one after the other between them
        self.geoPrefetch_ = _env.flag("GEO_PREFETCH")
        self.geoSeen_ = {}
        self.geoPlan_ = []
        self.prefetchedGeo_ = {}    # prefetch_geometry() on the native path: keyPDF -> (Geometry, keyGrid, keyNeighs, usePDF, transposed)
        self.multiFeatureConvs_ = multiFeatureConvs
        self.KDEWindow_ = KDEWindow
        self.relativeRadius_ = relativeRadius
        self.usePDF_ = usePDF
        self.useAVG_ = useAVG
        self.decayLossCollection_ = decayLossCollection
        self.device_ = device
        self.collections_ = {}      # collection name -> list of parameters (tf.add_to_collection)
        self.opTrace_ = None        # optional list the builder appends (op, key) records to (tests)
        # prefetch_geometry(): side stream, parked geometry (grids, neighbours, pdfs, event), the event of the last
        # reset(), neighbour lists to transpose ahead of time, the dummy feature column of the geometry-only sort
        self.sideStream_ = None
        self.prefetched_ = None
        self.resetEvent_ = None
        self.prefetchTransposed_ = {}
        self.prefetchDummy_ = None

    # ------------------------------------------------------------------ variable store
    @property
    def variables_(self):
        """name -> torch.nn.Parameter (the module's registered parameters: the tf.get_variable store)."""
        return self._parameters

    def get_collection(self, name):
        return list(self.collections_.get(name, []))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Variables are created lazily (first create_convolution of a convName): a checkpoint may name variables that
        do not exist yet -- they are adopted as they are (also when a parent module's load_state_dict() recurses here)."""
        for k, v in state_dict.items():
            name = k[len(prefix):] if k.startswith(prefix) else None
            if name and "." not in name and name not in self._parameters:
                self.register_parameter(name, torch.nn.Parameter(v.detach().clone()))
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _get_variable(self, name, shape, device, init):
        p = self._parameters.get(name)
        if p is None:
            p = torch.nn.Parameter(init(torch.empty(shape, dtype=torch.float32, device=device)))
            self.register_parameter(name, p)
        elif tuple(p.shape) != tuple(shape):
            raise RuntimeError("variable %s exists with shape %s, requested %s" % (name, tuple(p.shape), shape))
        return p

    def forward(self, *args, **kwargs):
        return self.create_convolution(*args, **kwargs)

    def _add_to_collection(self, name, p):
        lst = self.collections_.setdefault(name, [])
        seen = self.__dict__.setdefault("_collection_ids", {}).setdefault(name, set())
        if id(p) not in seen:
            seen.add(id(p))
            lst.append(p)

    # ------------------------------------------------------------------ caches
    def __compute_dic_keys__(self, inPointHierarchy, outPointHierarchy, inPointLevel, outPointLevel, convRadius,
                             KDEWindow, relativeRadius, usePDF):
        # MCConvBuilder.py:203-238 (the strings depend on names and numbers only: memoised -- six str() of floats per call)
        memo = self.__dict__.setdefault("_keyMemo", {})
        k = (inPointHierarchy.hierarchyName_, outPointHierarchy.hierarchyName_, inPointLevel, outPointLevel, convRadius, KDEWindow,
             relativeRadius, usePDF)
        try:
            hit = memo.get(k)
        except TypeError:   # (an unhashable argument: a tensor radius)
            hit, k = None, None
        if hit is not None:
            return hit
        keyGrid = inPointHierarchy.hierarchyName_ + '|' + str(inPointLevel) + '|' + str(convRadius) + '|' + \
            str(relativeRadius)
        keyNeighs = keyGrid + '|' + outPointHierarchy.hierarchyName_ + '|' + str(outPointLevel)
        keyPDF = keyNeighs + '|' + str(KDEWindow) + '|' + str(usePDF)
        if k is not None and len(memo) < 4096:
            memo[k] = (keyGrid, keyNeighs, keyPDF)
        return keyGrid, keyNeighs, keyPDF

    def reset(self):
        """Drop the operation caches (MCConvBuilder.py:241-246). Variables are kept. Geometry parked by
        prefetch_geometry() since the last reset() becomes the new cache content.

        hostStepsAhead_ (extension, attribute; None = unbounded): with k, reset() waits until the GPU has finished every
        step but the last k (0: everything enqueued so far -- the steps are issued one at a time). 1 is what a training
        loop wants: memory of at most two steps in flight, and on the pipelined loop (PointHierarchy.prefetch two batches
        ahead + prefetch_step) no slower than running free -- BASELINE cfg2 1.43 ms per step against 1.49 unbounded and
        1.67 one at a time, cfg3 5.90 / 6.01 / 6.10, cfg4 1.81 / 1.83 / 2.04 (round 5, tools/lag_probe.py)."""
        state = self.__dict__   # (plain state is written through __dict__ here: a dozen assignments per step, see _PlainState)
        k = state.get("hostStepsAhead_", None)
        if k is not None and _cuda_ok():
            evs = state.setdefault("stepEvents_", [])
            evs.append(_record_event())      # the end of the step that has just been issued
            del evs[:-8]
            if len(evs) > k:
                _M = _hip_ops_mod()
                t0 = time.perf_counter()
                evs[-(int(k) + 1)].synchronize()
                dt = time.perf_counter() - t0
                _M.HOST_WAIT_S[0] += dt
                _M.HOST_LAG_WAIT_S[0] += dt
        state["cacheGrids_"], state["cacheNeighs_"], state["cachePDFs_"] = {}, {}, {}
        if self.geoSeen_:
            # the geometries the step's layers USED (built by them, prebuilt, or started a step ago by prefetch_step) and
            # the pieces they attached to each (row plans, transposed list): what the next step asks for ahead
            plan = []
            for key, ent in self.geoSeen_.items():
                geo = self.cacheGeo_.get(key)
                plan.append(ent + ((geo.have & 7) if geo is not None else 0, key))
            state["geoPlan_"], state["geoSeen_"] = plan, {}
        state["cacheGeo_"], state["cacheGeoGrid_"] = {}, {}
        if self.prefetchedGeo_:
            self.__install_prefetched_geometries__()
        pf, state["prefetched_"] = self.prefetched_, None
        if pf is not None:
            grids, neighs, pdfs, event = pf
            main = torch.cuda.current_stream()
            main.wait_event(event)  # GPU-side: whatever is launched from here on runs after the side stream's work
            for kN, h in list(neighs.items()):
                if hasattr(h, "finalize"):  # enqueued without a host wait: the edge total has arrived by now
                    st, pk, pdf = h.finalize()
                    neighs[kN] = (st, pk)
                    pdfs[h.keyPDF] = pdf
            for d in (grids, neighs, pdfs):
                for v in d.values():
                    for t in (v if isinstance(v, tuple) else (v,)):
                        # allocated on the side stream, read on this one: the allocator may hand the block out again only
                        # behind this stream's reads. (Rounds 2-4 dropped these tensors here themselves when reference-
                        # count probes said the builder was their last owner -- 8 blocks x 5.5 us of main-queue time per
                        # step saved on this fallback path, for a test that three CPython / torch internals stay as they
                        # are. The native executor, the default path, never needed it.)
                        t.record_stream(main)
            self.cacheGrids_, self.cacheNeighs_, self.cachePDFs_ = grids, neighs, pdfs
        state["resetEvent_"] = _record_event() if (_SIDE_EVENTS[0] and _cuda_ok()) else None
        if pf is not None:
            for kN, (kG, kP, centres, mn, mx, B, radius, rel) in self.prefetchTransposed_.items():
                if kN in neighs and kG in grids and getattr(self.ops_, "_ops", 0) is None:
                    _hip_ops = _hip_ops_mod()
                    self.sideStream_.wait_event(self.resetEvent_)
                    g = grids[kG]
                    _hip_ops.prefetch_transposed(neighs[kN][1], g[0].shape[0], self.sideStream_)
                    if kP in pdfs:
                        # ... and the transposed ROW PLAN the depth-wise backward passes sweep (MCConvModule._row_plan):
                        # under the forward convolutions as well; its consumer waits for the plan's event
                        _hip_ops.prefetch_rowplan(neighs[kN][1], True, self.sideStream_, g[0], g[1], pdfs[kP], centres,
                                                  neighs[kN][0], neighs[kN][1], mn, mx, g[0].shape[0], centres.shape[0],
                                                  neighs[kN][1].shape[0], B, radius, rel, self.useAVG_)
            self.prefetchTransposed_ = {}

    def prefetch_geometry(self, inPointHierarchy, inPointLevel, convRadius, outPointHierarchy=None, outPointLevel=None,
                          KDEWindow=None, relativeRadius=None, usePDF=None, transposed=False):
        """Extension (no counterpart in the reference): computes the grid, the neighbour list and the PDFs that
        create_convolution() with the same arguments looks up in the caches -- for the NEXT batch, on a side stream, and
        parks them until the next reset(). Geometry depends on the points only, not on the network, so in a training
        loop the grid build / search / KDE of batch k + 1 runs under the convolution kernels of batch k: those hold
        two waves per SIMD (VGPR-bound) and leave issue slots and wave slots that the light geometry kernels fill
        (100k room: 0.70 -> 0.58 ms per step). Several calls between two reset()s accumulate.

        When to call it: on the native step executor (the default) the side stream starts behind what the calling stream
        holds at the moment of the call, so call it right after reset() -- BEFORE the current batch's convolutions are
        launched -- once the next batch's hierarchy has been enqueued; called later it is still correct, the build then
        starts later. The op-by-op protocol (MCCNN_NATIVE_PREFETCH=0, a builder without the executor) waits for the
        hierarchies' own events instead and is called after the backward pass has been launched.

        transposed=True (depth-wise layers will convolve over this neighbour list): reset() also starts the list's
        transposition for their backward pass on the side stream, where it runs under the forward convolutions.
        transposed="list" (layers with 2..4 input features and multiFeatureConv=True will): the transposed list alone --
        their feature gradient is then gathered through it instead of scattered with float atomics (bit-reproducible)."""
        currKDEWindow = self.KDEWindow_ if KDEWindow is None else KDEWindow
        currRelativeRadius = self.relativeRadius_ if relativeRadius is None else relativeRadius
        currUsePDF = self.usePDF_ if usePDF is None else usePDF
        outPH = inPointHierarchy if outPointHierarchy is None else outPointHierarchy
        outLevel = inPointLevel if outPointLevel is None else outPointLevel
        keyGrid, keyNeighs, keyPDF = self.__compute_dic_keys__(inPointHierarchy, outPH, inPointLevel, outLevel, convRadius,
                                                               currKDEWindow, currRelativeRadius, currUsePDF)
        pts, bids = inPointHierarchy.points_[inPointLevel], inPointHierarchy.batchIds_[inPointLevel]
        mn, mx, B = inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_, inPointHierarchy.batchSize_
        if not pts.is_cuda:
            return  # host tensors (a CPU checker behind `ops=`): nothing to overlap, create_convolution computes inline
        if self.__prefetch_native__(inPointHierarchy, inPointLevel, convRadius, outPH, outLevel, currKDEWindow,
                                    currRelativeRadius, currUsePDF, keyGrid, keyNeighs, keyPDF, transposed):
            return
        if self.sideStream_ is None:
            self.sideStream_ = torch.cuda.Stream(device=pts.device)
        _SIDE_EVENTS[0] = True   # (hierarchies and reset()s record their events from here on)
        pf = self.prefetched_
        grids, neighs, pdfs = (pf[0], pf[1], pf[2]) if pf is not None else ({}, {}, {})
        side = self.sideStream_
        # the point hierarchies have to be complete before the side stream reads them. Each records an event when its
        # last level has been enqueued (PointHierarchy.readyEvent_): the side stream waits for exactly that -- NOT for the
        # convolutions launched since, which it is meant to overlap. In a training loop reset() comes first and the next
        # batch's hierarchy is built after it, so the event of the last reset() alone would not cover that work. A
        # hierarchy without an event (built by other means) falls back to everything the calling stream holds now.
        waited = False
        for ph in ((inPointHierarchy,) if outPH is inPointHierarchy else (inPointHierarchy, outPH)):
            ev = getattr(ph, "readyEvent_", None)
            if ev is None:
                side.wait_stream(torch.cuda.current_stream())
                waited = True
                break
            side.wait_event(ev)
        if not waited:   # memory retired at the last reset() is reused only behind it
            if self.resetEvent_ is not None:
                side.wait_event(self.resetEvent_)
            else:
                side.wait_stream(torch.cuda.current_stream())
        bg = None
        if getattr(self.ops_, "_ops", 0) is None:  # the HIP surface: this thread's launches are background work for a while
            from . import _lib as _mclib
            bg = _mclib.load()
            bg_prev = bg.mccnn_background_launches(1)
        try:
            self.__prefetch_on_side__(side, grids, neighs, pdfs, keyGrid, keyNeighs, keyPDF, pts, bids, mn, mx, B, convRadius,
                                      currRelativeRadius, currKDEWindow, currUsePDF, outPH, outLevel, transposed)
        finally:
            if bg is not None:
                bg.mccnn_background_launches(bg_prev)

    def __prefetch_on_side__(self, side, grids, neighs, pdfs, keyGrid, keyNeighs, keyPDF, pts, bids, mn, mx, B, convRadius,
                             currRelativeRadius, currKDEWindow, currUsePDF, outPH, outLevel, transposed):
        with torch.cuda.stream(side):
            if keyGrid not in grids and self.fuseSort_ and getattr(self.ops_, "_ops", 0) is None and not pts.requires_grad:
                _hip_ops = _hip_ops_mod()
                grids[keyGrid] = _hip_ops.build_grid(pts, bids, mn, mx, B, convRadius, currRelativeRadius)
            if keyGrid not in grids:
                keys, indexs = self.ops_.sort_points_step1(pts, bids, mn, mx, B, convRadius, currRelativeRadius)
                dummy = self.prefetchDummy_  # geometry only: one zero feature per point, kept
                if dummy is None or dummy.shape[0] != pts.shape[0] or dummy.device != pts.device:
                    dummy = self.prefetchDummy_ = torch.zeros((pts.shape[0], 1), dtype=torch.float32, device=pts.device)
                sortPts, sortBatchs, _, cellIndexs = self.ops_.sort_points_step2(pts, bids, dummy, keys, indexs, mn, mx, B,
                                                                                convRadius, currRelativeRadius)
                grids[keyGrid] = (sortPts, sortBatchs, cellIndexs, indexs)
            g = grids[keyGrid]
            deferred = None
            if getattr(self.ops_, "_ops", 0) is None:  # the HIP op surface (not a checker handed in through `ops=`)
                _hip_ops = _hip_ops_mod()
                deferred = _hip_ops.find_neighbors_pdf_deferred
            if keyNeighs not in neighs and keyPDF not in pdfs and currUsePDF and deferred is not None:
                # search + KDE without a host wait (list sizes from the last total of this shape; None on the first call)
                h = deferred(outPH.points_[outLevel], outPH.batchIds_[outLevel], g[0], g[1], g[2], mn, mx, convRadius, B,
                             currRelativeRadius, currKDEWindow)
                if h is not None:
                    h.keyPDF = keyPDF
                    neighs[keyNeighs] = h
                    pdfs[keyPDF] = h
            if transposed:
                self.prefetchTransposed_[keyNeighs] = (keyGrid, keyPDF, outPH.points_[outLevel], mn, mx, B, convRadius,
                                                       currRelativeRadius)
            if keyNeighs not in neighs:
                neighs[keyNeighs] = tuple(self.ops_.find_neighbors(outPH.points_[outLevel], outPH.batchIds_[outLevel], g[0],
                                                                   g[2], mn, mx, convRadius, B, currRelativeRadius))
            nb = neighs[keyNeighs]
            if keyPDF not in pdfs:
                if currUsePDF:
                    pdfs[keyPDF] = self.ops_.compute_pdf(g[0], g[1], mn, mx, nb[0], nb[1], currKDEWindow, convRadius, B,
                                                         currRelativeRadius)
                else:
                    pdfs[keyPDF] = torch.ones((nb[1].shape[0], 1), dtype=torch.float32, device=nb[1].device)
            event = torch.cuda.Event()
            event.record(side)
        self.prefetched_ = (grids, neighs, pdfs, event)

    def _trace(self, *rec):
        if self.opTrace_ is not None:
            self.opTrace_.append(rec)

    def __layer_variables__(self, convName, inNumFeatures, currNumOutFeatures, currMultiFeatureConv, dev):
        """The six kernel-MLP variables of `convName` (MCConvBuilder.py:394-419), created on first use."""
        blockSize = self.ops_.get_block_size()
        numOutNeurons = inNumFeatures * currNumOutFeatures if currMultiFeatureConv else inNumFeatures
        numBlocks = int(numOutNeurons / blockSize)
        if numOutNeurons % blockSize != 0:
            numBlocks = numBlocks + 1
        zeros = lambda t: t.zero_()
        nn = blockSize * numBlocks
        weights = self._get_variable(convName + '_weights', (3, nn), dev, lambda t: _fan_avg_uniform_(t, 3, nn))
        self._add_to_collection(self.decayLossCollection_, weights)
        biases = self._get_variable(convName + '_biases', (nn,), dev, zeros)
        # TF fans for a rank-3 variable [numBlocks, bs, bs]: receptive field = numBlocks, fan_in = fan_out = bs*numBlocks
        weights2v = self._get_variable(convName + '_weights2', (numBlocks, blockSize, blockSize), dev,
                                       lambda t: _fan_avg_uniform_(t, numBlocks * blockSize, numBlocks * blockSize))
        self._add_to_collection(self.decayLossCollection_, weights2v)
        biases2v = self._get_variable(convName + '_biases2', (numBlocks, blockSize), dev, zeros)
        weights3v = self._get_variable(convName + '_weights3', (numBlocks, blockSize, blockSize), dev,
                                       lambda t: _fan_avg_uniform_(t, numBlocks * blockSize, numBlocks * blockSize))
        self._add_to_collection(self.decayLossCollection_, weights3v)
        biases3v = self._get_variable(convName + '_biases3', (numBlocks, blockSize), dev, zeros)
        return weights, biases, weights2v, biases2v, weights3v, biases3v, nn

    def __prefetch_native__(self, inPH, inLevel, convRadius, outPH, outLevel, KDEWindow, relativeRadius, usePDF, keyGrid,
                            keyNeighs, keyPDF, transposed, fork=True, pieces=0):
        """prefetch_geometry() on the native step executor: the geometry is ONE buffer, allocated on the CALLER's stream and
        written on a side stream that starts behind everything the caller's stream holds at this moment; the layers that
        use it order their stream behind its event. No reference counting decides anything: the buffer goes back to the
        caller's stream's allocator when the geometry dies, after every reader. For the build to run UNDER the current
        batch's convolutions, call prefetch_geometry() BEFORE they are launched (right after reset()); called after them
        it is still correct, the build then simply waits for them. Returns False when this call has to take the op-by-op
        prefetch (no torch extension, points with a gradient, ...)."""
        _native = _native_mod()
        _hip_ops = _hip_ops_mod()
        if not (self.native_ and self.fuseSort_ and getattr(self.ops_, "_ops", 0) is None and _native.side_streams_available()
                and int(_hip_ops.PDF_MODE) == 1 and self.prefetched_ is None
                and _env.debug("native_prefetch", True)):
            return False
        inPts, inBids = inPH.points_[inLevel], inPH.batchIds_[inLevel]
        centres, cBids = outPH.points_[outLevel], outPH.batchIds_[outLevel]
        if inPts.requires_grad or inPts.shape[0] == 0 or centres.shape[0] == 0:
            return False
        for t, dt in ((inPts, torch.float32), (centres, torch.float32), (inBids, torch.int32), (cBids, torch.int32)):
            if t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                return False
        want = 0 if not transposed else (1 if transposed == "list" else 2)   # nothing | transposed list | + transposed row plan
        if keyPDF in self.prefetchedGeo_:
            ent = self.prefetchedGeo_[keyPDF]
            self.prefetchedGeo_[keyPDF] = ent[:4] + (max(ent[4], want),)
            return True
        mn, mx, B = inPH.aabbMin_, inPH.aabbMax_, inPH.batchSize_
        nc = _hip_ops._num_cells(mn, mx, B, convRadius, relativeRadius)
        owners = self.__dict__.setdefault("prefetchedGridOwner_", {})   # keyGrid -> the parked geometry that owns that grid
        owner = owners.get(keyGrid)
        k = len(self.prefetchedGeo_)
        geo = _native.build_geometry(inPts, inBids, centres, cBids, mn, mx, B, nc, convRadius, relativeRadius, KDEWindow,
                                     usePDF, owner, side=k, fork=fork, background=True,
                                     after=(inPH.prefetchFuture_ if inPH is outPH else None))
        geo.uses = 0
        if owner is None:
            owners[keyGrid] = geo
        if pieces:   # row plans / transposed list the layers of the last step used: attached and issued by a helper thread
            geo.prebuild_async(pieces, self.useAVG_)
        self.prefetchedGeo_[keyPDF] = (geo, keyGrid, keyNeighs, usePDF, want)
        return True

    def prefetch_step(self, pointHierarchy):
        """Extension: everything the LAST step built over a hierarchy of this name -- every grid, neighbour list and PDF, and
        the row plans / transposed lists its layers used -- started now for `pointHierarchy`, the NEXT batch's hierarchy, on
        side streams under the current batch's convolutions; parked until the next reset(). The learned form of
        prefetch_geometry(): no argument lists to repeat, the builder remembers the graph. A training loop that has the
        next batch's hierarchy at hand (PointHierarchy.prefetch two batches ahead) calls this right after reset(); the
        step after it then finds its geometry built and runs its convolutions back to back. Returns the number of
        geometries started (0 on the first steps, while there is nothing to replay, and wherever the native path does
        not apply: nothing is lost, the next step builds what it needs itself)."""
        started = 0
        name, levels = pointHierarchy.hierarchyName_, len(pointHierarchy.points_)
        pieces = _env.debug("plan_prefetch", True)
        _native = _native_mod()
        _native.begin_batch()   # the step's geometries go out together: one launch per kernel kind over all of them
        try:
            started = self.__prefetch_step_entries__(pointHierarchy, name, levels, pieces)
        finally:
            _native.end_batch()
        return started

    def __prefetch_step_entries__(self, pointHierarchy, name, levels, pieces):
        started = 0
        for ent in self.geoPlan_:
            hname, inLevel, outLevel, radius, window, rel, usePDF, have = ent[:8]
            if hname != name or inLevel >= levels or outLevel >= levels:
                continue
            keyGrid, keyNeighs, keyPDF = self.__compute_dic_keys__(pointHierarchy, pointHierarchy, inLevel, outLevel, radius,
                                                                   window, rel, usePDF)
            if keyPDF in self.prefetchedGeo_:
                continue
            if self.__prefetch_native__(pointHierarchy, inLevel, radius, pointHierarchy, outLevel, window, rel, usePDF, keyGrid,
                                        keyNeighs, keyPDF, False, fork=(started == 0), pieces=(have if pieces else 0)):
                started += 1
        return started

    def __install_prefetched_geometries__(self):
        """reset(): the geometries prefetch_geometry() built since the last reset() become the cache content; for those
        asked for with transposed=True the transposed list and the transposed row plan of the depth-wise backward pass are
        started on their side stream now (their edge totals arrived long ago), under the forward passes to come."""
        _native = _native_mod()
        parked, self.prefetchedGeo_ = self.prefetchedGeo_, {}
        self.__dict__["prefetchedGridOwner_"] = {}
        for keyPDF, (geo, keyGrid, keyNeighs, usePDF, transposed) in parked.items():
            geo.unverified = True   # (checked against the hierarchy's tensors at its first use)
            self.cacheGeo_[keyPDF] = geo
            if geo.grid_owner is None:
                self.cacheGeoGrid_[keyGrid] = geo
                self.cacheGrids_[keyGrid] = _LazyEntry(geo, geo.grid)
            self.cacheNeighs_[keyNeighs] = _LazyEntry(geo, geo.neighbors)
            self.cachePDFs_[keyPDF] = _LazyEntry(geo, geo.pdfs)
            if transposed:
                # (depth-wise layers: both row plans -- the forward pass then waits for the first stage only)
                geo.prebuild((_native.NEED_PLAN_FWD | _native.NEED_PLAN_TR) if transposed == 2 else _native.NEED_TLIST,
                             self.useAVG_, 0, geo.buf)

    def __prebuild_geometries__(self, ph):
        """Learned prefetch: every geometry the previous step built over a hierarchy of this name, issued now -- before the
        first layer of this step -- on side streams. A step whose graph differs simply builds what is missing when it is
        asked for; a geometry nobody asks for is dropped at the next reset()."""
        _native = _native_mod()
        _hip_ops = _hip_ops_mod()
        plan, name = self.geoPlan_, ph.hierarchyName_
        levels = len(ph.points_)
        mn, mx, B = ph.aabbMin_, ph.aabbMax_, ph.batchSize_
        if int(_hip_ops.PDF_MODE) != 1:
            return
        k = 0
        pieces = _env.debug("plan_prefetch", True)
        for (hname, inLevel, outLevel, radius, window, rel, usePDF, have, _key) in plan:
            if hname != name or inLevel >= levels or outLevel >= levels:
                continue
            keyGrid, keyNeighs, keyPDF = self.__compute_dic_keys__(ph, ph, inLevel, outLevel, radius, window, rel, usePDF)
            if keyPDF in self.cacheGeo_:
                continue
            inPts, inBids = ph.points_[inLevel], ph.batchIds_[inLevel]
            centres, cBids = ph.points_[outLevel], ph.batchIds_[outLevel]
            if inPts.shape[0] == 0 or centres.shape[0] == 0 or inPts.requires_grad:
                continue
            ok = True
            for t, dt in ((inPts, torch.float32), (centres, torch.float32), (inBids, torch.int32), (cBids, torch.int32)):
                if t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                    ok = False
            if not ok:
                continue
            nc = _hip_ops._num_cells(mn, mx, B, radius, rel)
            owner = self.cacheGeoGrid_.get(keyGrid)
            geo = _native.build_geometry(inPts, inBids, centres, cBids, mn, mx, B, nc, radius, rel, window, usePDF, owner,
                                         side=k, fork=(k == 0), after=ph.prefetchFuture_)
            k += 1
            geo.uses = 0
            if have and pieces and geo.e_cap <= _PLAN_PREFETCH_MAX_E:
                geo.prebuild_async(have, self.useAVG_)
            self.cacheGeo_[keyPDF] = geo
            if owner is None:
                self.cacheGeoGrid_[keyGrid] = geo
                self.cacheGrids_[keyGrid] = _LazyEntry(geo, geo.grid)
                self._trace("sort_points_step1", keyGrid)
                self._trace("sort_points_step2", keyGrid)
            self.cacheNeighs_[keyNeighs] = _LazyEntry(geo, geo.neighbors)
            self.cachePDFs_[keyPDF] = _LazyEntry(geo, geo.pdfs)
            self._trace("find_neighbors", keyNeighs)
            if usePDF:
                self._trace("compute_pdf", keyPDF)

    def __native_convolution__(self, convName, inPH, inLevel, inFeatures, inNumFeatures, convRadius, outPH, outLevel,
                               multiFeatureConv, numOutFeatures, KDEWindow, relativeRadius, usePDF, useAVG, keyGrid,
                               keyNeighs, keyPDF):
        """create_convolution on the native step executor (mccnn_amd.native): the geometry of (keyGrid, keyNeighs, keyPDF)
        is ONE library call (no host wait), the layer one call per direction with the feature sort inside. Returns None
        when this call has to take the op-by-op path: a cache entry of that path exists already (prefetch_geometry), the
        level is empty, or the features are not rows the library reads in place."""
        _native = _native_mod()
        _hip_ops = _hip_ops_mod()
        # (a step with a handful of geometries gains nothing: the hops between the streams cost what the overlap saves --
        # measured: BASELINE cfg1, three lists, 0.93 -> 0.97 ms; cfg2, seven, 2.59 -> 2.18)
        if (not self.cacheGeo_ and self.geoPrefetch_ and len(self.geoPlan_) >= _GEO_PREFETCH_MIN and inPH is outPH and self.prefetched_ is None
                and not self.cacheGrids_ and _native.side_streams_available()):
            self.__prebuild_geometries__(inPH)
        geo = self.cacheGeo_.get(keyPDF)
        if geo is not None and getattr(geo, "unverified", False):
            # a geometry started ahead (prefetch_geometry / prefetch_step) is filed under the reference's cache keys -- names,
            # levels, radii -- like everything else; before its first use it is checked against the tensors it was built
            # from, so that a DIFFERENT hierarchy of the same name cannot be served another batch's lists
            a = geo.args
            cen = outPH.points_[outLevel]
            pin = inPH.points_[inLevel]
            if (a[0] is pin or (a[0].data_ptr() == pin.data_ptr() and a[0].shape == pin.shape)) and \
                    (a[2] is cen or (a[2].data_ptr() == cen.data_ptr() and a[2].shape == cen.shape)):
                geo.unverified = False
            else:
                if _GEO_TRACE:
                    print("native conv %s: parked geometry %s not for this hierarchy (built from %s / %s, asked %s / %s)" % (
                        convName, keyPDF, tuple(a[0].shape), tuple(a[2].shape), tuple(pin.shape), tuple(cen.shape)), file=sys.stderr)
                self.cacheGeo_.pop(keyPDF, None)
                self.cachePDFs_.pop(keyPDF, None)
                self.cacheNeighs_.pop(keyNeighs, None)
                if self.cacheGeoGrid_.get(keyGrid) is geo:
                    self.cacheGeoGrid_.pop(keyGrid, None)
                    self.cacheGrids_.pop(keyGrid, None)
                geo = None
        if geo is None:
            if keyGrid in self.cacheGrids_ and keyGrid not in self.cacheGeoGrid_:
                if _GEO_TRACE:
                    print("native conv %s: grid %s owned by the op path" % (convName, keyGrid), file=sys.stderr)
                return None   # the op-by-op path (or a prefetch) owns this grid
            if keyNeighs in self.cacheNeighs_ or keyPDF in self.cachePDFs_:
                if _GEO_TRACE:
                    print("native conv %s: list %s / pdf %s cached without a geometry (%s %s); geometries: %s" % (
                        convName, keyNeighs, keyPDF, keyNeighs in self.cacheNeighs_, keyPDF in self.cachePDFs_,
                        sorted(self.cacheGeo_)), file=sys.stderr)
                return None
            inPts, inBids = inPH.points_[inLevel], inPH.batchIds_[inLevel]
            centres, cBids = outPH.points_[outLevel], outPH.batchIds_[outLevel]
            if inPts.shape[0] == 0 or centres.shape[0] == 0 or int(_hip_ops.PDF_MODE) != 1:
                return None
            for t, dt in ((inPts, torch.float32), (centres, torch.float32), (inBids, torch.int32), (cBids, torch.int32)):
                if t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                    return None
            mn, mx, B = inPH.aabbMin_, inPH.aabbMax_, inPH.batchSize_
            nc = _hip_ops._num_cells(mn, mx, B, convRadius, relativeRadius)
            owner = self.cacheGeoGrid_.get(keyGrid)
            if owner is not None and getattr(owner, "unverified", False):
                # (a grid started ahead for another hierarchy of this name is not shared either)
                a0 = owner.args[0]
                if not (a0 is inPts or (a0.data_ptr() == inPts.data_ptr() and a0.shape == inPts.shape)):
                    self.cacheGeoGrid_.pop(keyGrid, None)
                    self.cacheGrids_.pop(keyGrid, None)
                    owner = None
            geo = _native.build_geometry(inPts, inBids, centres, cBids, mn, mx, B, nc, convRadius, relativeRadius, KDEWindow,
                                         usePDF, owner)
            geo.uses = 0
            self.cacheGeo_[keyPDF] = geo
            if owner is None:
                self.cacheGeoGrid_[keyGrid] = geo
                self.cacheGrids_[keyGrid] = _LazyEntry(geo, geo.grid)
                self._trace("sort_points_step1", keyGrid)
                self._trace("sort_points_step2", keyGrid)
            else:
                self._trace("sort_features", keyGrid)
            self.cacheNeighs_[keyNeighs] = _LazyEntry(geo, geo.neighbors)
            self.cachePDFs_[keyPDF] = _LazyEntry(geo, geo.pdfs)
            self._trace("find_neighbors", keyNeighs)
            if usePDF:
                self._trace("compute_pdf", keyPDF)
        else:
            self._trace("sort_features", keyGrid)
        if inPH is outPH and keyPDF not in self.geoSeen_:
            self.geoSeen_[keyPDF] = (inPH.hierarchyName_, inLevel, outLevel, convRadius, KDEWindow, relativeRadius, usePDF)
        feats = inFeatures
        if _GEO_TRACE and feats.dim() == 2 and feats.shape[0] != geo.n:
            print("native conv %s: %d feature rows for a geometry over %d points (key %s, unverified %s, built from %s, level has %s)" % (
                convName, feats.shape[0], geo.n, keyPDF, getattr(geo, "unverified", None), tuple(geo.args[0].shape),
                tuple(inPH.points_[inLevel].shape)), file=sys.stderr)
        if (not feats.is_cuda or feats.dim() != 2 or feats.shape[0] != geo.n or feats.shape[1] != inNumFeatures
                or feats.dtype not in (torch.float32, torch.bfloat16)):
            return None
        if feats.dtype == torch.bfloat16 and (multiFeatureConv or inNumFeatures % 8 != 0):
            return None
        if not feats.is_contiguous():
            feats = feats.contiguous()
        # the op's shape rules (spatial_conv.cc:290-296)
        neurons = inNumFeatures * numOutFeatures if multiFeatureConv else inNumFeatures
        if ((neurons + 7) // 8 * 8) % inNumFeatures != 0:
            raise _hip_ops.InvalidArgumentError("SpatialConvOp expects a number of output neurons multiple of the number of features.")
        if not multiFeatureConv and inNumFeatures % 8 != 0:
            raise _hip_ops.InvalidArgumentError("SpatialConvOp expects the same number of features in the input and the output")
        lay = self.layers_.get(convName)
        spec = (inNumFeatures, numOutFeatures, bool(multiFeatureConv))
        if lay is None or lay[0] != spec:
            dev = self.device_ or inFeatures.device
            v = self.__layer_variables__(convName, inNumFeatures, numOutFeatures, multiFeatureConv, dev)
            lay = self.layers_[convName] = (spec, v)
        weights, biases, weights2v, biases2v, weights3v, biases3v, nn = lay[1]
        self._trace("spatial_conv", convName, (3, nn), numOutFeatures, bool(multiFeatureConv))
        geo.uses += 1
        return _native.conv(geo, feats, weights, biases, weights2v, biases2v, weights3v, biases3v, numOutFeatures,
                            bool(multiFeatureConv), bool(useAVG))

    # ------------------------------------------------------------------ create_convolution
    def create_convolution(self, convName, inPointHierarchy, inPointLevel, inFeatures, inNumFeatures, convRadius,
                           outPointHierarchy=None, outPointLevel=None, multiFeatureConv=None, outNumFeatures=None,
                           KDEWindow=None, relativeRadius=None, usePDF=None, useAVG=None):
        # defaults: MCConvBuilder.py:299-325
        currMultiFeatureConv = self.multiFeatureConvs_ if multiFeatureConv is None else multiFeatureConv
        currNumOutFeatures = inNumFeatures if outNumFeatures is None else outNumFeatures
        currKDEWindow = self.KDEWindow_ if KDEWindow is None else KDEWindow
        currRelativeRadius = self.relativeRadius_ if relativeRadius is None else relativeRadius
        currUsePDF = self.usePDF_ if usePDF is None else usePDF
        currUseAVG = self.useAVG_ if useAVG is None else useAVG
        currOutPointHierarchy = inPointHierarchy if outPointHierarchy is None else outPointHierarchy
        currOutPointLevel = inPointLevel if outPointLevel is None else outPointLevel

        if currOutPointHierarchy.batchSize_ != inPointHierarchy.batchSize_:
            raise RuntimeError('Different batch size in the input and output point hierarchy')
        if (currMultiFeatureConv == False) and (currNumOutFeatures != inNumFeatures):
            raise RuntimeError('The number of input and output features should be the same '
                               'for multi feature convolutions.')

        keyGrid, keyNeighs, keyPDF = self.__compute_dic_keys__(
            inPointHierarchy, currOutPointHierarchy, inPointLevel, currOutPointLevel, convRadius, currKDEWindow,
            currRelativeRadius, currUsePDF)
        _log("Convolution: %s (KDE: %s | MF: %s | Rel: %s | PDF: %s)" % (convName, currKDEWindow,
                                                                       currMultiFeatureConv, currRelativeRadius,
                                                                       currUsePDF))

        inPts = inPointHierarchy.points_[inPointLevel]
        if (self.native_ and self.fuseSort_ and getattr(self.ops_, "_ops", 0) is None and inPts.is_cuda
                and not inPts.requires_grad):
            out = self.__native_convolution__(convName, inPointHierarchy, inPointLevel, inFeatures, inNumFeatures, convRadius,
                                              currOutPointHierarchy, currOutPointLevel, currMultiFeatureConv,
                                              currNumOutFeatures, currKDEWindow, currRelativeRadius, currUsePDF, currUseAVG,
                                              keyGrid, keyNeighs, keyPDF)
            if out is not None:
                return out

        # grid (MCConvBuilder.py:349-363)
        # HIP op surface, points that carry no gradient: the grid is built from the points alone (one library call) and
        # the feature rows are sorted inside the convolution's own autograd node -- per convolution one op call and one
        # graph node less than sort_points_step2 / sort_features + spatial_conv (same kernels, same results)
        sortIndex = None
        inPts = inPointHierarchy.points_[inPointLevel]
        fused = (self.fuseSort_ and getattr(self.ops_, "_ops", 0) is None and inPts.is_cuda and not inPts.requires_grad)
        if fused:
            if keyGrid in self.cacheGrids_:
                currGridTuple = self.cacheGrids_[keyGrid]
                self._trace("sort_features", keyGrid)
            else:
                _hip_ops = _hip_ops_mod()
                currGridTuple = _hip_ops.build_grid(inPts, inPointHierarchy.batchIds_[inPointLevel],
                                                     inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_,
                                                     inPointHierarchy.batchSize_, convRadius, currRelativeRadius)
                self.cacheGrids_[keyGrid] = currGridTuple
                self._trace("sort_points_step1", keyGrid)
                self._trace("sort_points_step2", keyGrid)
            sortFeatures, sortIndex = inFeatures, currGridTuple[3]
        elif keyGrid in self.cacheGrids_:
            currGridTuple = self.cacheGrids_[keyGrid]
            sortFeatures = self.ops_.sort_features(inFeatures, currGridTuple[3])
            self._trace("sort_features", keyGrid)
        else:
            keys, indexs = self.ops_.sort_points_step1(
                inPointHierarchy.points_[inPointLevel], inPointHierarchy.batchIds_[inPointLevel],
                inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_, inPointHierarchy.batchSize_, convRadius,
                currRelativeRadius)
            sortPts, sortBatchs, sortFeatures, cellIndexs = self.ops_.sort_points_step2(
                inPointHierarchy.points_[inPointLevel], inPointHierarchy.batchIds_[inPointLevel], inFeatures, keys,
                indexs, inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_, inPointHierarchy.batchSize_,
                convRadius, currRelativeRadius)
            currGridTuple = (sortPts, sortBatchs, cellIndexs, indexs)
            self.cacheGrids_[keyGrid] = currGridTuple
            self._trace("sort_points_step1", keyGrid)
            self._trace("sort_points_step2", keyGrid)

        # neighbours (MCConvBuilder.py:366-376)
        if fused and currUsePDF and keyNeighs not in self.cacheNeighs_ and keyPDF not in self.cachePDFs_:
            # search + KDE enqueued back to back (list sizes from the last total of this shape), ONE wait for the edge
            # count at the end instead of a wait between the two ops; None on the first call of a shape
            _hip_ops = _hip_ops_mod()
            h = _hip_ops.find_neighbors_pdf_deferred(
                currOutPointHierarchy.points_[currOutPointLevel], currOutPointHierarchy.batchIds_[currOutPointLevel],
                currGridTuple[0], currGridTuple[1], currGridTuple[2], inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_,
                convRadius, inPointHierarchy.batchSize_, currRelativeRadius, currKDEWindow)
            if h is not None:
                startIndexs, packedNeighs, pdfsNow = h.finalize()
                self.cacheNeighs_[keyNeighs] = (startIndexs, packedNeighs)
                self.cachePDFs_[keyPDF] = pdfsNow
                self._trace("find_neighbors", keyNeighs)
                self._trace("compute_pdf", keyPDF)
        if keyNeighs in self.cacheNeighs_:
            currNeighTuple = self.cacheNeighs_[keyNeighs]
        else:
            startIndexs, packedNeighs = self.ops_.find_neighbors(
                currOutPointHierarchy.points_[currOutPointLevel], currOutPointHierarchy.batchIds_[currOutPointLevel],
                currGridTuple[0], currGridTuple[2], inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_, convRadius,
                inPointHierarchy.batchSize_, currRelativeRadius)
            currNeighTuple = (startIndexs, packedNeighs)
            self.cacheNeighs_[keyNeighs] = currNeighTuple
            self._trace("find_neighbors", keyNeighs)

        # pdf (MCConvBuilder.py:379-391)
        if keyPDF in self.cachePDFs_:
            currPDFs = self.cachePDFs_[keyPDF]
        else:
            if currUsePDF:
                currPDFs = self.ops_.compute_pdf(currGridTuple[0], currGridTuple[1], inPointHierarchy.aabbMin_,
                                       inPointHierarchy.aabbMax_, currNeighTuple[0], currNeighTuple[1], currKDEWindow,
                                       convRadius, inPointHierarchy.batchSize_, currRelativeRadius)
                self._trace("compute_pdf", keyPDF)
            else:
                currPDFs = torch.ones((currNeighTuple[1].shape[0], 1), dtype=torch.float32,
                                      device=currNeighTuple[1].device)
            self.cachePDFs_[keyPDF] = currPDFs

        # variables (MCConvBuilder.py:394-419)
        blockSize = self.ops_.get_block_size()
        dev = self.device_ or inFeatures.device
        weights, biases, weights2v, biases2v, weights3v, biases3v, nn = self.__layer_variables__(
            convName, inNumFeatures, currNumOutFeatures, currMultiFeatureConv, dev)

        self._trace("spatial_conv", convName, (3, nn), currNumOutFeatures, bool(currMultiFeatureConv))
        if sortIndex is None:
            weights2, weights3 = weights2v.reshape(blockSize, nn), weights3v.reshape(blockSize, nn)
            biases2, biases3 = biases2v.reshape(nn), biases3v.reshape(nn)
        if sortIndex is not None:
            _hip_ops = _hip_ops_mod()
            # (the variables in their stored shapes: the op reinterprets them itself, see _SpatialConv.forward)
            return _hip_ops.spatial_conv(currGridTuple[0], sortFeatures, currGridTuple[1], currPDFs,
                                currOutPointHierarchy.points_[currOutPointLevel], currNeighTuple[0], currNeighTuple[1],
                                inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_, weights, weights2v, weights3v, biases,
                                biases2v, biases3v, currNumOutFeatures, currMultiFeatureConv, inPointHierarchy.batchSize_,
                                convRadius, currRelativeRadius, currUseAVG, sortIndex, True,
                                currGridTuple[4] if len(currGridTuple) > 4 else None)
        return self.ops_.spatial_conv(currGridTuple[0], sortFeatures, currGridTuple[1], currPDFs,
                            currOutPointHierarchy.points_[currOutPointLevel], currNeighTuple[0], currNeighTuple[1],
                            inPointHierarchy.aabbMin_, inPointHierarchy.aabbMax_, weights, weights2, weights3, biases,
                            biases2, biases3, currNumOutFeatures, currMultiFeatureConv, inPointHierarchy.batchSize_,
                            convRadius, currRelativeRadius, currUseAVG)
