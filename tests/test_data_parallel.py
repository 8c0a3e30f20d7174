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


# -- SCAE's own flat layout (aligned groups, the early-gradient front block,
#    GradSlot re-arming) through the bucketed two-all-reduce path ------------
SCAE_CFG = dict(image_shape=(1, 16, 16), n_classes=4, n_part_caps=5,
                n_obj_caps=4,
                pcae_cnn_encoder_params=dict(out_channels=[8, 8],
                                             kernel_sizes=[3, 3],
                                             strides=[2, 1]),
                pcae_template_generator_params=dict(template_size=(5, 5)),
                ocae_encoder_set_transformer_params=dict(dim_hidden=8,
                                                         dim_out=16,
                                                         n_layers=2),
                ocae_decoder_capsule_params=dict(dim_caps=4,
                                                 hidden_sizes=(8,)),
                scae_params=dict(reconstruct_alternatives=False))


def make_scae():
    import numpy as np
    from torch_scae_amd import factory
    np.random.seed(0)
    torch.manual_seed(0)
    return factory.make_scae(SCAE_CFG)


def fake_backward(flat, names, seed, step):
    """What a backward of the HIP ops does to the flat buffers: most
    parameters get their gradient written straight into their slot (adopted
    as p.grad), some get a separate tensor from autograd, two get none."""
    g = torch.Generator().manual_seed(1000 * seed + step)
    want = {}
    for i, (n, p) in enumerate(zip(names, flat.params)):
        if n.endswith("dummy_vote") or n.startswith("posterior_classifier"):
            continue                               # no gradient at all
        val = torch.randn(p.shape, generator=g)
        want[n] = val
        if i % 3 == 0:
            p.grad = val.clone()                   # produced by autograd itself
        else:
            v = p._scae_grad_slot.take()
            assert v is not None, n                # slots were re-armed
            v.copy_(val)
            p.grad = v
            assert p._scae_grad_slot.take() is None    # one taker per step
    return want


def _scae_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_scae_amd.data_parallel import (FlatParameters, RMSpropFlat,
                                              all_reduce_gradients,
                                              broadcast_parameters)
    from torch_scae_amd.train_step import EARLY_PREFIXES
    model = make_scae()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.25 * rank)
    flat = FlatParameters(model, front=lambda n: n.startswith(EARLY_PREFIXES))
    by_id = {id(p): n for n, p in model.named_parameters()}
    names = [by_id[id(p)] for p in flat.params]
    broadcast_parameters(flat)
    opt = RMSpropFlat(flat, lr=1e-2, eps=1e-3)
    wants = []
    for step in range(2):
        flat.clear_grads()
        want = fake_backward(flat, names, rank, step)
        # the order TrainStep uses: bucket 0 leaves while "part B" still runs
        early = {n: v for n, v in want.items() if n.startswith(EARLY_PREFIXES)}
        flat.gather_grads(0)
        w0 = all_reduce_gradients(flat, average=False, which=0, async_op=True)
        flat.gather_grads(1)
        w1 = all_reduce_gradients(flat, average=False, which=1, async_op=True)
        w0.wait()
        w1.wait()
        assert len(early) > 0
        wants.append(want)
        grads = flat.flat_grad.clone()
        opt.step(grad_scale=1.0 / world)
    out[rank] = (names, list(flat.offsets), flat.n_front, wants, grads,
                 flat.flat_param.clone())
    dist.destroy_process_group()


def test_scae_flat_layout_front_block_and_alignment():
    from torch_scae_amd.data_parallel import FlatParameters
    from torch_scae_amd.train_step import EARLY_PREFIXES
    model = make_scae()
    n_params = sum(p.numel() for p in model.parameters())
    flat = FlatParameters(model, front=lambda n: n.startswith(EARLY_PREFIXES))
    by_id = {id(p): n for n, p in model.named_parameters()}
    names = [by_id[id(p)] for p in flat.params]
    early = [n.startswith(EARLY_PREFIXES) for n in names]
    k = sum(early)
    assert 0 < k < len(names) and all(early[:k]) and not any(early[k:])
    assert flat.front_count == k and flat.n_front % 4 == 0
    assert flat.offsets[k] == flat.n_front
    assert flat.block_grad(0).numel() + flat.block_grad(1).numel() \
        == flat.numel >= n_params
    assert flat.block_grad(1).data_ptr() % 16 == flat.flat_grad.data_ptr() % 16
    # the object encoder's trunk group still lies back to back, 16-byte aligned
    groups = [g for m in model.modules()
              for g in getattr(m, "_flat_param_groups", lambda: [])()]
    assert groups
    for group in groups:
        idx = [[id(p) for p in flat.params].index(id(q)) for q in group]
        assert idx == list(range(idx[0], idx[0] + len(idx)))
        assert flat.offsets[idx[0]] % 4 == 0
        for a, b in zip(idx, idx[1:]):
            assert flat.offsets[b] == flat.offsets[a] + flat.params[a].numel()
    # without the hint the module order is kept and there is one block
    plain = FlatParameters(make_scae())
    assert plain.n_front == plain.numel


def test_two_rank_bucketed_all_reduce_on_scae_layout():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_scae_worker, args=(world, _free_port(), out), nprocs=world,
             join=True)
    names, offsets, n_front, wants0, g0, p0 = out[0]
    _, _, _, wants1, g1, p1 = out[1]
    assert torch.equal(g0, g1) and torch.equal(p0, p1)
    # the last step's reduced gradient: the sum of both ranks' gradients in
    # every parameter's slot, zeros for the gradient-less parameters and pads
    expect = torch.zeros_like(g0)
    for n, off in zip(names, offsets):
        if n in wants0[-1]:
            v = wants0[-1][n] + wants1[-1][n]
            expect[off:off + v.numel()] = v.reshape(-1)
    assert torch.allclose(g0, expect, atol=1e-6)
    assert 0 < n_front < g0.numel()


CFG2 = dict(image_shape=(1, 40, 40), n_classes=10, n_part_caps=24, n_obj_caps=24,
            scae_params=dict(reconstruct_alternatives=False))


def _cfg2_step_worker(rank, world, port, out):
    """TrainStep's OWN two-bucket schedule (``TrainStep._run``: part A, bucket
    0 in flight while part B runs, bucket 1, both waited for, fused RMSprop)
    on the real flat layout of BASELINE.json configs[1] -- 2.4 M parameters,
    228 tensors, the decoders' block in front -- with stand-ins for the two
    halves of the backward (no GPU here)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from torch_scae_amd import factory
    from torch_scae_amd.train_step import EARLY_PREFIXES, TrainStep
    np.random.seed(0)
    torch.manual_seed(0)
    model = factory.make_scae(CFG2)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.125 * rank)          # broadcast_parameters has to undo this
    step = TrainStep(model, 128, CFG2["image_shape"], use_graph=False,
                     lr=1e-3, collective_mode="2 buckets")
    assert step.collective and step.split and step.world == world
    flat = step.flat
    by_id = {id(p): n for n, p in model.named_parameters()}
    names = [by_id[id(p)] for p in flat.params]
    lo0, hi0 = flat.block(0)
    lo1, hi1 = flat.block(1)
    assert (lo0, hi0, hi1) == (0, flat.front_count, len(flat.params)) and lo1 == hi0
    assert all(n.startswith(EARLY_PREFIXES) for n in names[:hi0])
    assert not any(n.startswith(EARLY_PREFIXES) for n in names[hi0:])
    order, wants = [], {}

    def half(which):
        lo, hi = flat.block(which)
        g = torch.Generator().manual_seed(100 * rank + which)

        def run():
            if which == 0:
                flat.clear_grads()
            for n, p in zip(names[lo:hi], flat.params[lo:hi]):
                if n.endswith("dummy_vote") or n.startswith("posterior_classifier"):
                    continue                       # no gradient in this configuration
                v = p._scae_grad_slot.take()
                v.copy_(torch.randn(p.shape, generator=g))
                p.grad = v
                wants[n] = v.clone()
            flat.gather_grads(which)
            order.append(("part", which))
        return run

    reduce_real = step._reduce

    def reduce_spy(which=None, async_op=False):
        order.append(("reduce", which))
        return reduce_real(which, async_op=async_op)
    step._reduce = reduce_spy
    before = flat.flat_param.clone()
    step._run(half(0), half(1))
    step.opt.step(grad_scale=1.0 / world)
    assert order == [("part", 0), ("reduce", 0), ("part", 1), ("reduce", 1)], order
    out[rank] = dict(names=names, offsets=list(flat.offsets), n_front=flat.n_front,
                     numel=flat.numel, wants=wants, grads=flat.flat_grad.clone(),
                     before=before, after=flat.flat_param.clone())
    dist.destroy_process_group()


def test_two_rank_train_step_bucket_schedule_on_cfg2_layout():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cfg2_step_worker, args=(world, _free_port(), out), nprocs=world,
             join=True)
    a, b = out[0], out[1]
    # BASELINE.json configs[1]: 2 414 879 parameters; the first
    # bucket -- the two decoders and the classifier heads -- is two thirds of the bytes
    assert 2414879 <= a["numel"] <= 2414879 + 4 * len(a["names"])
    assert 0.6 < a["n_front"] / a["numel"] < 0.72
    assert torch.equal(a["before"], b["before"])          # rank 0's weights everywhere
    assert torch.equal(a["grads"], b["grads"]) and torch.equal(a["after"], b["after"])
    expect = torch.zeros_like(a["grads"])
    for n, off in zip(a["names"], a["offsets"]):
        if n in a["wants"]:
            v = a["wants"][n] + b["wants"][n]
            expect[off:off + v.numel()] = v.reshape(-1)
    assert torch.allclose(a["grads"], expect, atol=1e-6)
    assert float((a["after"] - a["before"]).abs().max()) > 0
