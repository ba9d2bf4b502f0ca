"""A BASELINE.json configuration's step with the geometry issued FIRST: python tools/config_geofirst.py cfg2 [steps]

Same work as bench.ConfigWorkload.step (PointHierarchy + forward + backward of every convolution), other order: the
grids / neighbour lists / PDFs of every (level, radius) the model convolves over are enqueued through
ConvolutionBuilder.prefetch_geometry right after the hierarchy (no host wait between them), reset() parks them in the
caches, and the convolutions follow without a single edge-count wait in between -- in the reference's order the host
waits at the first use of every neighbour list for everything queued before it. Prints both step times."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mccnn_amd.workloads import CONFIGS  # noqa: E402


class GeometryFirst(bench.ConfigWorkload):
    first = False

    def step(self):
        if not self.first:
            return super().step()
        b = self.builder
        ph = self.ph = self.hierarchy()
        seen = set()
        for c in self.cfg.convs:
            key = (c.lin, c.lout, round(c.radius, 9), c.window)
            if key in seen:
                continue
            seen.add(key)
            b.prefetch_geometry(ph, c.lin, c.radius, ph, c.lout, c.window, transposed=any(
                (not d.combin) and (d.lin, d.lout, round(d.radius, 9), d.window) == key for d in self.cfg.convs))
        b.reset()
        outs = [self.conv(ph, ci) for ci in range(len(self.cfg.convs))]
        inputs = self.feats + list(b.parameters())
        self.grads = torch.autograd.grad(outs, inputs, self.ogs, allow_unused=True)
        return outs


names = [a for a in sys.argv[1:] if not a.isdigit()] or ["cfg1", "cfg2", "cfg3", "cfg4"]
steps = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [20])[0]
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
for name in names:
    cw = GeometryFirst(CONFIGS[name], torch.device("cuda", 0))
    ref = [o.detach().float().clone() for o in cw.step()]
    res = {}
    for rep in range(2):
        for first in (False, True):
            cw.first = first
            ms, launches = cw.timed(steps, 5)
            res.setdefault(first, []).append((ms, cw.host_issue_ms, launches))
    cw.first = True
    outs = cw.step()
    same = all(torch.equal(o.detach().float(), r) for o, r in zip(outs, ref))
    print("%s: reference order %s ms/step (host issue %s) | geometry first %s ms/step (host issue %s), launches %.0f / %.0f, "
          "outputs identical: %s" % (name, ["%.3f" % r[0] for r in res[False]], ["%.3f" % r[1] for r in res[False]],
                                     ["%.3f" % r[0] for r in res[True]], ["%.3f" % r[1] for r in res[True]],
                                     res[False][0][2], res[True][0][2], same))
