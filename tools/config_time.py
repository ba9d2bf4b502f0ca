"""One BASELINE.json configuration (mccnn_amd.workloads) as bench.py measures it, stand-alone -- for rocprofv3:

    PROF_CMD="python $PWD/tools/config_time.py cfg1 30" tools/prof.sh r03_cfg1

One step = PointHierarchy + forward + backward of every convolution of the model's graph (bench.ConfigWorkload)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
cw = bench.ConfigWorkload(CONFIGS[name], torch.device("cuda", 0))
if os.environ.get("PIPE", "0") == "1":   # the next batch's hierarchy one step ahead (bench.run_config's pipelined mode)
    print("pipelined:", cw.set_pipeline(True, geometry=os.environ.get("DEEP", "1") == "1"))
ms, launches = cw.timed(steps, 5)
print("%s: %d points, %.4f ms/step, %.1f M points/s, %.1f library launches/step" % (
    name, cw.P.shape[0], ms, cw.P.shape[0] / ms / 1e3, launches))
