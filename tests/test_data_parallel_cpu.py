"""N > 1 path on CPU: world_size-2 gloo processes exercise the data-parallel host logic of
usip_amd.step (pair sharding, flat gradient bucket, all-reduce-mean) -- the only exchange the
path has.  The HIP forward/backward itself needs a GPU and is covered by -m gpu tests;
here the 'backward' is a stand-in that writes rank-dependent gradients through the same bucket
views autograd accumulates into."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions, build_detector
    from usip_amd.step import FlatGradBucket, shard_pairs

    torch.manual_seed(0)                                   # identical replicas, as bench.py does
    net = build_detector("som", DetectorOptions(surface_normal_len=3))
    bucket = FlatGradBucket(net)
    n_params = sum(p.numel() for p in net.parameters())
    assert bucket.flat.numel() == n_params == 1198660       # Cs=3 detector
    # gradients are views into the one flat buffer (a single all-reduce moves all 46 tensors)
    for p in net.parameters():
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()
        assert p.grad.data_ptr() < bucket.flat.data_ptr() + 4 * n_params
    # stand-in backward: rank-dependent gradient, accumulated in place like autograd does
    bucket.zero()
    for i, p in enumerate(net.parameters()):
        p.grad.add_(torch.full_like(p, float(rank + 1) * (i + 1)))
    bucket.all_reduce_mean()
    want = [(1 + 2) / 2.0 * (i + 1) for i in range(len(list(net.parameters())))]
    ok = all(torch.allclose(p.grad, torch.full_like(p, w)) for p, w in zip(net.parameters(), want))
    # pair sharding: contiguous, disjoint, complete
    batch = synth.make_pair_batch(7, 4, 64, 8, 3, "sphere")
    mine = shard_pairs(batch, rank, world)
    assert mine["src_pc"].shape[0] == 2
    ok = ok and np.array_equal(mine["src_pc"], batch["src_pc"][rank * 2:(rank + 1) * 2])
    # parameters stay identical across ranks after an identical update of identical gradients
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    opt.step()
    flat_p = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat_p) for _ in range(world)]
    dist.all_gather(gathered, flat_p)
    ok = ok and torch.equal(gathered[0], gathered[1])
    # replicas that drifted apart (what a captured all-reduce that did not reduce would leave behind) are put back
    # together from rank 0: parameters, BatchNorm buffers, optimizer state -- usip_amd.step._resync_from_rank0
    import types
    from usip_amd import step as step_mod
    if rank == 1:
        with torch.no_grad():
            for p_ in net.parameters():
                p_.add_(0.5)
            for b_ in net.buffers():
                if b_.dtype.is_floating_point:
                    b_.add_(1.0)
            for st_ in opt.state.values():
                st_["exp_avg"].add_(2.0)
    owner = next(c for c in vars(step_mod).values() if isinstance(c, type) and "_resync_from_rank0" in vars(c))
    owner._resync_from_rank0(types.SimpleNamespace(module=net, optimizer=opt), None)
    for t in [p_.detach() for p_ in net.parameters()] + [b_ for b_ in net.buffers() if b_.dtype.is_floating_point] + \
            [st_["exp_avg"] for st_ in opt.state.values()]:
        flat_t = t.reshape(-1).float()
        both = [torch.empty_like(flat_t) for _ in range(world)]
        dist.all_gather(both, flat_t)
        ok = ok and torch.equal(both[0], both[1])
    # the two agreement primitives of the one-graph form (usip_amd/step.py): capture success is a logical AND over the
    # ranks, and reduced gradients are compared as bit patterns -- a NaN everywhere is agreement, not a reason to re-sync
    dev = torch.device("cpu")
    ns = types.SimpleNamespace(device=dev, bucket=types.SimpleNamespace(flat=torch.full((8,), 3.0)))
    ok = ok and owner._all_ranks_agree(ns, True, None) is True
    ok = ok and owner._all_ranks_agree(ns, rank == 0, None) is False          # one rank's capture failed: nobody fuses
    ok = ok and owner._gradients_agree(ns, None) is True
    ns.bucket.flat = torch.full((8,), float("nan"))
    ok = ok and owner._gradients_agree(ns, None) is True
    ns.bucket.flat = torch.full((8,), float(rank))
    ok = ok and owner._gradients_agree(ns, None) is False
    # random point dropout (keypoint_detector.py:160-168): ONE keep ratio and index set per step for the whole batch --
    # rank 0 draws, the others receive (host RNGs are deliberately seeded differently here)
    import random
    random.seed(100 + rank)
    np.random.seed(200 + rank)
    opt_d = DetectorOptions(surface_normal_len=3)
    opt_d.random_pc_dropout_lower_limit, opt_d.input_pc_num = 0.5, 64
    pre = types.SimpleNamespace(opt=opt_d, device=dev, use_graph=False, _SIAMESE=step_mod.DetectorStep._SIAMESE)
    b = {k: torch.from_numpy(v) for k, v in mine.items()}
    kept = step_mod.DetectorStep._prepare(pre, b)
    n_kept = torch.tensor([kept["src_pc"].shape[2]])
    both = [torch.empty_like(n_kept) for _ in range(world)]
    dist.all_gather(both, n_kept)
    ok = ok and int(both[0]) == int(both[1]) and 32 <= int(both[0]) <= 64 and pre._replay_this_call is False
    # the points kept are the same columns on every rank: rank 1's first cloud gathered with rank 0's choice
    col0 = kept["src_pc"][0, 0, :4].clone()
    probe = torch.from_numpy(mine["src_pc"])[0, 0]
    pos = torch.tensor([int((probe == v).nonzero()[0]) for v in col0])
    both = [torch.empty_like(pos) for _ in range(world)]
    dist.all_gather(both, pos)
    ok = ok and torch.equal(both[0], both[1])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_detector_state_dict_keys_match_reference_checkpoints():
    """46 parameter tensors, reference key names, 1,198,724 parameters at Cs=4."""
    from usip_amd.networks import DetectorOptions, build_detector
    net = build_detector("som", DetectorOptions(surface_normal_len=4))
    keys = dict(net.named_parameters())
    assert len(keys) == 46 and sum(p.numel() for p in keys.values()) == 1198724
    assert tuple(keys["first_pointnet.layers.0.conv.weight"].shape) == (64, 7, 1)
    assert tuple(keys["knnlayer_1.layers_before.0.conv.weight"].shape) == (256, 131, 1, 1)
    assert tuple(keys["knnlayer_1.layers_after.1.conv.weight"].shape) == (512, 512, 1, 1)
    assert tuple(keys["mlp1.conv.weight"].shape) == (512, 640, 1)
    assert tuple(keys["mlp3.conv.weight"].shape) == (4, 256, 1)
    sd = net.state_dict()
    assert "first_pointnet.layers.0.norm.running_mean" in sd
    assert "mlp1.norm.num_batches_tracked" in sd
    assert "first_pointnet.layers.2.norm.weight" not in sd        # last PointNet layer: no BN
    ball = build_detector("ball", DetectorOptions(surface_normal_len=4))
    bk = dict(ball.named_parameters())
    assert tuple(bk["conv1.conv.weight"].shape) == (64, 7, 1, 1)
    assert tuple(bk["conv5.conv.weight"].shape) == (128, 128, 1, 1)
    assert len(bk) == 50 and sum(p.numel() for p in bk.values()) == 1199108   # conv3/conv5 keep their BN


def test_product_forward_refuses_host_tensors():
    """No CPU fallback in the product path."""
    import pytest
    from usip_amd.networks import DetectorOptions, build_detector
    net = build_detector("ball", DetectorOptions())
    x = torch.randn(1, 3, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(x, torch.randn(1, 4, 64), x[:, :, :8].contiguous())


def test_flat_adam_takes_a_per_parameter_adam_checkpoint():
    """the CPU / gloo path trains with torch.optim.Adam (one state per parameter), the GPU path with FlatAdam
    (one state over the flat buffer).  A checkpoint of the former loads into the latter: moments end to end in
    parameter order, the step count, the (decayed) learning rate; FlatAdam's own checkpoint round-trips."""
    from usip_amd.networks import DetectorOptions, build_detector
    from usip_amd.step import FlatAdam
    torch.manual_seed(1)
    net = build_detector("som", DetectorOptions(surface_normal_len=3))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    opt.param_groups[0]["lr"] = 2.5e-4
    sd = opt.state_dict()
    n = sum(p.numel() for p in net.parameters())
    flat = torch.nn.Parameter(torch.zeros(n))
    fa = FlatAdam(flat, lr=1e-3)
    fa.load_state_dict(sd)
    st = fa.state[flat]
    want = torch.cat([opt.state[p]["exp_avg_sq"].reshape(-1) for p in net.parameters()])
    assert torch.equal(st["exp_avg_sq"], want)
    assert torch.equal(st["exp_avg"], torch.cat([opt.state[p]["exp_avg"].reshape(-1) for p in net.parameters()]))
    assert float(st["step"]) == 1.0 and fa.param_groups[0]["lr"] == 2.5e-4
    own = fa.state_dict()
    fb = FlatAdam(torch.nn.Parameter(torch.zeros(n)), lr=1e-3)
    fb.load_state_dict(own)
    assert torch.equal(fb.state[fb.param]["exp_avg"], st["exp_avg"]) and fb.param_groups[0]["lr"] == 2.5e-4
    assert own["state"][0]["step"].dim() == 0


def test_flat_adam_refuses_checkpoint_options_it_does_not_implement():
    """FlatAdam is plain Adam (what models/keypoint_detector.py:42-45 constructs); a checkpoint that asks for
    weight decay or amsgrad is refused instead of being trained without it."""
    import pytest
    from usip_amd.step import FlatAdam
    p = torch.nn.Parameter(torch.zeros(16))
    p.grad = torch.zeros(16)
    opt = FlatAdam(p, lr=1e-3)
    sd = opt.state_dict()
    sd["param_groups"][0]["lr"] = 5e-4
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 5e-4
    for key, val in (("weight_decay", 1e-4), ("amsgrad", True)):
        bad = opt.state_dict()
        bad["param_groups"][0][key] = val
        with pytest.raises(ValueError, match="not implemented"):
            opt.load_state_dict(bad)
