"""Host logic of the data-parallel path on CPU: flat buffers, the single
gradient all-reduce (gloo, world_size 2) and the flat RMSprop step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def make_net():
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
    net.unused = nn.Parameter(torch.zeros(4))      # never gets a gradient
    return net


def test_flat_parameters_and_rmsprop_match_torch_optim():
    from torch_scae_amd.data_parallel import FlatParameters, RMSpropFlat
    a, b = make_net(), make_net()
    flat = FlatParameters(a)
    opt_a = RMSpropFlat(flat, lr=1e-2, eps=1e-3, momentum=0.9)
    opt_b = torch.optim.RMSprop(b.parameters(), lr=1e-2, eps=1e-3,
                                momentum=0.9)
    assert flat.numel == sum(p.numel() for p in a.parameters())
    g = torch.Generator().manual_seed(1)
    for _ in range(4):
        x = torch.randn(7, 6, generator=g)
        flat.clear_grads()
        a(x).square().sum().backward()
        flat.gather_grads()
        opt_a.step()
        opt_b.zero_grad()
        b(x).square().sum().backward()
        b.unused.grad = torch.zeros(4)
        opt_b.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, atol=1e-6), (pa - pb).abs().max()
    # parameters are views of the flat buffer; state_dict still works
    lo = flat.flat_param.data_ptr()
    assert lo <= a[0].weight.data_ptr() < lo + 4 * flat.numel
    assert set(a.state_dict()) == set(b.state_dict())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_scae_amd.data_parallel import (FlatParameters, RMSpropFlat,
                                              all_reduce_gradients,
                                              broadcast_parameters)
    torch.manual_seed(100 + rank)                  # ranks start different
    net = make_net()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(rank * 0.5)
    flat = FlatParameters(net)
    broadcast_parameters(flat)                     # -> rank 0's weights
    opt = RMSpropFlat(flat, lr=1e-2, eps=1e-3)
    g = torch.Generator().manual_seed(7)
    full = torch.randn(8, 6, generator=g)          # global batch, sharded
    shard = full[rank * 4:(rank + 1) * 4]
    flat.clear_grads()
    net(shard).square().sum(1).mean().backward()
    flat.gather_grads()
    summed = flat.flat_grad.clone()
    all_reduce_gradients(flat)
    out[rank] = (flat.flat_grad.clone(), None)
    # the path TrainStep takes: SUM all-reduce, 1/world folded into the step
    dist.all_reduce(summed)
    twin = make_net()
    twin.load_state_dict(net.state_dict())
    tflat = FlatParameters(twin)
    tflat.flat_grad.copy_(summed)
    topt = RMSpropFlat(tflat, lr=1e-2, eps=1e-3)
    topt.step(grad_scale=1.0 / world)
    opt.step()
    assert torch.allclose(tflat.flat_param, flat.flat_param, atol=1e-7)
    out[rank] = (out[rank][0], flat.flat_param.clone())
    dist.destroy_process_group()


def test_two_rank_gradient_all_reduce_equals_global_batch():
    from torch_scae_amd.data_parallel import FlatParameters
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    g0, p0 = out[0]
    g1, p1 = out[1]
    assert torch.equal(g0, g1) and torch.equal(p0, p1)
    # reference: one process, the whole batch (mean of shard means == mean)
    net = make_net()
    flat = FlatParameters(net)
    g = torch.Generator().manual_seed(7)
    full = torch.randn(8, 6, generator=g)
    net(full).square().sum(1).mean().backward()
    flat.gather_grads()
    assert torch.allclose(flat.flat_grad, g0, atol=1e-6)
    assert float(g0[:4].abs().sum()) == 0.0        # the unused parameter
    assert float(g0[4:].abs().sum()) > 0.0
