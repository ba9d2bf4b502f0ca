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


@pytest.mark.parametrize("deep", ["0", "1"], ids=["hierarchy_ahead", "hierarchy_and_geometry_ahead"])
def test_network_soak_with_everything_running_ahead(mc, deep):
    """deep: the hierarchy two batches ahead and ConvolutionBuilder.prefetch_step() for the next batch's geometry and plans."""
    env = dict(os.environ, SOAK_STEPS="400", SOAK_DEEP=deep)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_network.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"forward mismatches (\d+), worst relative gradient deviation (\S+), memory now (\d+) MB \(start (\d+)\)", out.stdout)
    assert m, out.stdout
    assert int(m.group(1)) == 0 and float(m.group(2)) < 1e-4
    assert int(m.group(3)) <= int(m.group(4)) + 64   # nothing accumulates over the steps
