"""Dense helpers (mccnn_amd.MCNetworkUtils): variable names / shapes of the reference, batch-norm semantics, and
synchronised statistics across two gloo ranks (whole-batch normalisation, utils/MCNetworkUtils.py:140)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mccnn_amd import MCNetworkUtils as NU


def test_variable_names_and_shapes():
    st = NU.VariableStore()
    x = torch.randn(10, 5)
    y = NU.MLP_2_hidden(x, 5, 7, 6, 3, "Final_Logits", 0.5, True, True, store=st)
    z = NU.conv_1x1("Reduce_1", x, 5, 4, st)
    w = NU.MLP_1_hidden(x, 5, 9, 2, "H", 0.5, True, store=st)
    assert y.shape == (10, 3) and z.shape == (10, 4) and w.shape == (10, 2)
    shapes = {k: tuple(v.shape) for k, v in st.named_parameters()}
    assert shapes["Final_Logits_weights1"] == (5, 7) and shapes["Final_Logits_weights2"] == (7, 6)
    assert shapes["Final_Logits_weights3"] == (6, 3) and shapes["Final_Logits_biases3"] == (3,)
    assert shapes["Reduce_1_weights"] == (5, 4) and shapes["H_weights2"] == (9, 2)
    assert len(st.get_collection("weight_decay_loss")) == 3 + 1 + 2
    assert float(st.variables_["Final_Logits_biases1"].abs().sum()) == 0.0


def test_batch_norm_matches_torch_and_tracks_moving_stats():
    st = NU.VariableStore()
    x = torch.randn(64, 4) * 3 + 1
    y = NU.batch_normalization(x, True, "bn", st)
    ref = torch.nn.functional.batch_norm(x, None, None, training=True, eps=1e-3)
    assert torch.allclose(y, ref, atol=1e-5)
    assert torch.allclose(st.buffers_["bn/moving_mean"], 0.01 * x.mean(0), atol=1e-6)
    ye = NU.batch_normalization(x, False, "bn", st)
    assert torch.allclose(ye, (x - st.buffers_["bn/moving_mean"]) / torch.sqrt(st.buffers_["bn/moving_variance"] + 1e-3), atol=1e-5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        full = torch.randn(50, 3) * 2 + 0.5
        part = full[:20] if rank == 0 else full[20:]          # ragged shards: 20 and 30 points
        part = part.clone().requires_grad_(True)
        st = NU.VariableStore()
        y = NU.batch_normalization(part, True, "bn", st)
        (y * y).sum().backward()
        q.put((rank, y.detach().numpy(), part.grad.numpy()))
    finally:
        dist.destroy_process_group()


def test_sync_batch_norm_equals_whole_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (y, g)) for r, y, g in [q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    full = (torch.randn(50, 3) * 2 + 0.5).requires_grad_(True)
    y = NU.batch_normalization(full, True, "bn", NU.VariableStore())
    (y * y).sum().backward()
    got = np.concatenate([res[0][0], res[1][0]])
    gg = np.concatenate([res[0][1], res[1][1]])
    assert np.allclose(got, y.detach().numpy(), atol=1e-5)
    assert np.allclose(gg, full.grad.numpy(), atol=1e-4)
