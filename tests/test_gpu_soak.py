"""The pipelined loop of a whole network under changing batch sizes (tools/soak_network.py, shortened): hierarchy of the next
batch on its helper thread, learned geometry prefetch on side streams, row plans on the third helper thread -- outputs of
every convolution bit-identical to the references computed with nothing running ahead, gradients within float-atomic noise,
memory flat."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("deep", ["0", "1", "fork"], ids=["hierarchy_ahead", "hierarchy_and_geometry_ahead",
                                                          "everything_forked_from_the_calling_stream"])
def test_network_soak_with_everything_running_ahead(mc, deep):
    """deep: the hierarchy two batches ahead and ConvolutionBuilder.prefetch_step() for the next batch's geometry and plans.
    By default the prefetched hierarchy starts at once (after=True) and the geometry streams wait for the hierarchy's event,
    with their memory from their own streams' pools; "fork": both start behind what the calling stream holds (the form
    for inputs produced on the calling stream, and the only one up to round 4)."""
    env = dict(os.environ, SOAK_STEPS="400", SOAK_DEEP="1" if deep == "fork" else deep)
    if deep == "fork":
        env.update(SOAK_HIER_AFTER="0", MCCNN_DEBUG="geo_own_pool=0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_network.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"forward mismatches (\d+), worst relative gradient deviation (\S+), memory now (\d+) MB \(start (\d+)\)", out.stdout)
    assert m, out.stdout
    assert int(m.group(1)) == 0 and float(m.group(2)) < 1e-4
    assert int(m.group(3)) <= int(m.group(4)) + 64   # nothing accumulates over the steps


def test_room_network_soak_with_changing_room_sizes(mc):
    """The MCSegScanNet graph (absolute radii, 17 layers, 5 levels) over rooms of 30 k .. 100 k points and a two-room batch in
    random order, hierarchy two batches ahead + prefetch_step, the prefetched hierarchy on the phased Poisson form (most
    launches on its stream). This is the run that exposed a helper thread's scratch being taken from the caching allocator's
    pool of ANOTHER stream without ordering (level sizes of a prefetched hierarchy overwritten by convolution kernels)."""
    # (also: the steps issued one at a time -- ConvolutionBuilder.hostStepsAhead_ = 0 --, random layer subsets, skipped backward passes)
    env = dict(os.environ, SOAK_STEPS="500", SOAK_DEEP="1", SOAK_CFG="cfg4", MCCNN_DEBUG="hier_pmode=0", SOAK_LAG="0",
               SOAK_VARY_GRAPH="1", SOAK_SKIP_BWD="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_network.py")], env=env, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"forward mismatches (\d+), worst relative gradient deviation (\S+),", out.stdout)
    assert m and int(m.group(1)) == 0 and float(m.group(2)) < 1e-4, out.stdout


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_script_that_ends_with_prefetched_work_in_flight(mc, cfg):
    """A script whose last steps start the next batch's hierarchy, geometry and row plans and never consume them: the helper
    threads still hold tensors with Python objects when the interpreter finalises. Dropping those takes the GIL, which
    Python answers with a forced unwind of the thread (std::terminate, a core dump after the script's last line) -- the
    extension's helpers are drained and joined from atexit instead (native.py, Issuer::retire in csrc/torch_ext.cpp)."""
    env = dict(os.environ, NOCONV="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_phases.py"), cfg, "1"], env=env, capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, "rc %d\n" % out.returncode + out.stdout[-1500:] + out.stderr[-1500:]
    assert "terminate called" not in out.stderr, out.stderr[-1500:]
    assert "ms/step" in out.stdout


def test_geometry_dropped_unconsumed_orders_the_default_stream(mc):
    """Round 6, NOTES item 6c: a prefetched geometry nobody consumes borrows its hierarchy's level tensors; when it dies it has
    to order the CALLER's stream -- the only consumer the allocator knows for them -- behind its build / plan kernels first. The
    handle of the device's default stream is a null pointer and used to be taken for "no caller" (GPU memory faults in a
    prefetch-and-drop loop). Deterministic check: a few prefetch-and-drop steps on the default stream must have enqueued such
    orderings (the extension counts them)."""
    code = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from mccnn_amd import native
from mccnn_amd.workloads import CONFIGS
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
cw = bench.ConfigWorkload(CONFIGS["cfg2"], torch.device("cuda", 0))
assert cw.set_pipeline(True, geometry=True)
for _ in range(8):
    cw.step()                      # (the builder learns the step's geometries and pieces)
torch.cuda.synchronize()
c0 = native._EXT.debug_counters()["caller_orderings"]
assert torch.cuda.current_stream().cuda_stream == 0   # the default stream: a null handle
for _ in range(6):                 # prefetch and drop: reset, adopt, request, prefetch_step -- no layer
    cw.builder.reset()
    cw.ph = cw.ready_ph
    nxt = cw.hierarchy(cw.next_ph)
    cw.request_next()
    cw.builder.prefetch_step(nxt)
    cw.ready_ph = nxt
cw.builder.reset(); cw.builder.reset()
torch.cuda.synchronize()
import gc; gc.collect()
c1 = native._EXT.debug_counters()["caller_orderings"]
print("caller orderings", c1 - c0)
assert c1 - c0 >= 6, (c0, c1)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "caller orderings" in out.stdout
