#!/usr/bin/env python3
"""End-to-end consumer of the drop-in path: the MCClassS and MCNormS graphs of the reference
(models/MCClassS.py:25-77, models/MCNormS.py:25-58) written against mccnn_amd's builder and dense helpers, plus a
tiny synthetic training loop. The graph builders keep the reference's structure line by line; only `tf.` is gone.

    python examples/mcclass_s.py [--steps 20]
"""
import argparse
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mccnn_amd.MCConvBuilder import PointHierarchy, ConvolutionBuilder
from mccnn_amd.MCNetworkUtils import (MLP_2_hidden, batch_norm_RELU_drop_out, conv_1x1, VariableStore)


class MCClassS(torch.nn.Module):
    """models/MCClassS.py create_network(): 3 Poisson levels, 3 MC convolutions, global-feature MLP. A torch.nn.Module
    whose sub-modules (the convolution builder and the dense helpers' variable store) register every variable under
    the reference's name, so `parameters()` / `state_dict()` / optimisers / DDP see the whole network."""

    def __init__(self, numInputFeatures, batchSize, k, numOutCat, device, ops=None):
        super().__init__()
        self.args = (numInputFeatures, batchSize, k, numOutCat)
        self.store = VariableStore(device)
        self.ops = ops  # None = the HIP op surface; the parity tests pass the CPU checker's
        self.convBuilder = ConvolutionBuilder(KDEWindow=0.2, device=device, ops=ops)

    RADII = [0.1, 0.4, math.sqrt(3.0) + 0.1]

    def prefetch_hierarchy(self, points, batchIds, after=None):
        """Extension: starts the point hierarchy of a batch on a stream of its own (PointHierarchy.prefetch) -- call it for
        batch k + 1 before the forward pass of batch k and hand the result to forward(prefetched=...). after: what the
        build waits for -- None: the calling stream (the batch was uploaded there), a torch.cuda.Event of the loader's
        upload stream, or True (the batch is complete)."""
        return PointHierarchy.prefetch(points, batchIds, self.RADII, self.args[1], after=after)

    def hierarchy(self, points, batchIds, features, prefetched=None):
        """The network's point hierarchy for a batch (prefetched: what prefetch_hierarchy() returned for it)."""
        return PointHierarchy(points, features, batchIds, self.RADII, "MCClassS_PH", self.args[1], ops=self.ops,
                              prefetched=prefetched)

    def forward(self, points, batchIds, features, isTraining, keepProbConv=1.0, keepProbFull=0.5, useConvDropOut=False,
                 useDropOutFull=True, prefetched=None, hierarchy=None, nextHierarchy=None):
        """hierarchy (extension): this batch's hierarchy, built a step ago with self.hierarchy(); nextHierarchy: the NEXT
        batch's -- its grids, neighbour lists, PDFs and row plans are then started under this batch's layers
        (ConvolutionBuilder.prefetch_step) and the next forward pass finds them built."""
        numInputFeatures, batchSize, k, numOutCat = self.args
        st, mConvBuilder = self.store, self.convBuilder
        mConvBuilder.reset()
        if nextHierarchy is not None:
            mConvBuilder.prefetch_step(nextHierarchy)
        mPointHierarchy = hierarchy if hierarchy is not None else self.hierarchy(points, batchIds, features, prefetched)
        convFeatures1 = mConvBuilder.create_convolution(
            convName="Conv_1", inPointHierarchy=mPointHierarchy, inPointLevel=0, outPointLevel=1, inFeatures=features,
            inNumFeatures=numInputFeatures, outNumFeatures=k, convRadius=0.2, multiFeatureConv=True)
        convFeatures1 = batch_norm_RELU_drop_out("Reduce_1_In_BN", convFeatures1, isTraining, useConvDropOut, keepProbConv, st)
        convFeatures1 = conv_1x1("Reduce_1", convFeatures1, k, k * 2, st)
        convFeatures1 = batch_norm_RELU_drop_out("Reduce_1_Out_BN", convFeatures1, isTraining, useConvDropOut, keepProbConv, st)
        convFeatures2 = mConvBuilder.create_convolution(
            convName="Conv_2", inPointHierarchy=mPointHierarchy, inPointLevel=1, outPointLevel=2,
            inFeatures=convFeatures1, inNumFeatures=k * 2, convRadius=0.8)
        convFeatures2 = batch_norm_RELU_drop_out("Reduce_2_In_BN", convFeatures2, isTraining, useConvDropOut, keepProbConv, st)
        convFeatures2 = conv_1x1("Reduce_2", convFeatures2, k * 2, k * 4, st)
        convFeatures2 = batch_norm_RELU_drop_out("Reduce_2_Out_BN", convFeatures2, isTraining, useConvDropOut, keepProbConv, st)
        convFeatures3 = mConvBuilder.create_convolution(
            convName="Conv_3", inPointHierarchy=mPointHierarchy, inPointLevel=2, outPointLevel=3,
            inFeatures=convFeatures2, inNumFeatures=k * 4, convRadius=math.sqrt(3.0) + 0.1)
        finalInput = batch_norm_RELU_drop_out("BNRELUDROP_final", convFeatures3, isTraining, useConvDropOut, keepProbConv, st)
        finalLogits = MLP_2_hidden(finalInput, k * 4, k * 2, k, numOutCat, "Final_Logits", keepProbFull, isTraining,
                                   useDropOutFull, store=st)
        self.lastHierarchy = mPointHierarchy
        return finalLogits


class MCNormS(torch.nn.Module):
    """models/MCNormS.py create_network(): two same-level multi-feature convolutions (normal estimation)."""

    def __init__(self, numInputFeatures, batchSize, k, device, ops=None):
        super().__init__()
        self.args = (numInputFeatures, batchSize, k)
        self.store = VariableStore(device)
        self.ops = ops
        self.convBuilder = ConvolutionBuilder(KDEWindow=0.2, device=device, ops=ops)

    def forward(self, points, batchIds, features, isTraining):
        numInputFeatures, batchSize, k = self.args
        cb = self.convBuilder
        cb.reset()
        ph = PointHierarchy(points, features, batchIds, [], "MCNormS_PH", batchSize, ops=self.ops)
        c1 = cb.create_convolution(convName="Conv_1", inPointHierarchy=ph, inPointLevel=0, inFeatures=features,
                                   inNumFeatures=numInputFeatures, outNumFeatures=k, convRadius=0.15, multiFeatureConv=True)
        c1 = batch_norm_RELU_drop_out("BN_RELU", c1, isTraining, False, False, self.store)
        return cb.create_convolution(convName="Conv_2", inPointHierarchy=ph, inPointLevel=0, inFeatures=c1,
                                     inNumFeatures=k, outNumFeatures=3, convRadius=0.15, multiFeatureConv=True)


def synthetic_batch(batchSize, nPts, numCat, rng, device):
    """Shapes whose class is a geometric property: ellipsoids with class-dependent axis ratios, unit-box normalised."""
    pts, bids, labels = [], [], []
    for b in range(batchSize):
        c = int(rng.integers(0, numCat))
        axes = np.array([1.0, 0.3 + 0.7 * c / max(numCat - 1, 1), 0.3 + 0.7 * ((c * 3) % numCat) / max(numCat - 1, 1)])
        v = rng.normal(size=(nPts, 3))
        p = v / np.linalg.norm(v, axis=1, keepdims=True) * axes
        p = (p - p.min(0)) / (p.max(0) - p.min(0)).max()
        pts.append(p.astype(np.float32))
        bids.append(np.full((nPts, 1), b, np.int32))
        labels.append(c)
    t = lambda a: torch.from_numpy(a).to(device)
    P, Bi = t(np.concatenate(pts)), t(np.concatenate(bids))
    return P, Bi, torch.ones((P.shape[0], 1), dtype=torch.float32, device=device), t(np.array(labels, np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=1024)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    # one process per GPU: run the backward pass on the calling thread -- the autograd engine's per-device worker thread
    # only adds a hand-off per backward() and this network is host-bound (cfg1 step: 4.0 -> 3.3 ms, tools/e2e_time.py)
    torch.autograd.set_multithreading_enabled(False)
    net = MCClassS(1, args.batch, 16, 4, device)
    P, Bi, F, y = synthetic_batch(args.batch, args.points, 4, rng, device)
    net(P, Bi, F, True)  # creates the variables
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    # the loader runs two batches ahead: the hierarchy of batch k + 2 is requested (helper thread, own stream) while batch
    # k trains; the one of batch k + 1 is complete by then, and its geometry is started under batch k's layers
    cur = synthetic_batch(args.batch, args.points, 4, rng, device)
    nxt = synthetic_batch(args.batch, args.points, 4, rng, device)
    ph_cur = net.hierarchy(cur[0], cur[1], cur[2])
    fut_nxt = net.prefetch_hierarchy(nxt[0], nxt[1])
    for step in range(args.steps):
        P, Bi, F, y = cur
        ph_nxt = net.hierarchy(nxt[0], nxt[1], nxt[2], prefetched=fut_nxt)
        after = synthetic_batch(args.batch, args.points, 4, rng, device)
        fut_after = net.prefetch_hierarchy(after[0], after[1])
        logits = net(P, Bi, F, True, hierarchy=ph_cur, nextHierarchy=ph_nxt)
        cur, nxt, ph_cur, fut_nxt = nxt, after, ph_nxt, fut_after
        loss = torch.nn.functional.cross_entropy(logits, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if step % 5 == 0 or step == args.steps - 1:
            print("step %3d  loss %.4f  acc %.2f  level sizes %s" % (
                step, float(loss), float((logits.argmax(1) == y).float().mean()),
                [int(p.shape[0]) for p in net.lastHierarchy.points_]))


if __name__ == "__main__":
    main()
