"""tests/golden/make_golden.py -- regenerates the golden fixtures in this directory.

Runs ONLY in the build container, where the reference is mounted at /root/reference:
it imports the reference's own Python modules (models.networks / layers / losses,
util.som) and its own index_max C++ (built unmodified into oracle/_ref/, see
oracle/build_ref.py), feeds them seeded inputs + closed-form weights
(usip_amd.synth.fill_parameters) and stores inputs and outputs as small .npz files.
Fixtures are data (numbers); no reference source text is stored.

Shims needed to import the reference here: an empty `torchvision`
module (util/som.py:12 imports it, never uses it on this path); `index_max` = the
reference's own CPU entry point; `ball_query` = oracle/usip_oracle.c (the reference has
no CPU ball_query; the oracle itself is pinned by ball_query_ancestor_cases.npz, rows
produced by the reference's commented-out numba kernel, see gen_ball_query_ancestor).

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import native                      # noqa: E402
from oracle.build_ref import load_ref_index_max  # noqa: E402
from usip_amd import synth                     # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def import_reference():
    import matplotlib
    matplotlib.use("Agg")
    ref_im = load_ref_index_max()
    assert ref_im is not None, "reference index_max could not be built"
    sys.modules["torchvision"] = types.ModuleType("torchvision")
    im = types.ModuleType("index_max")
    im.forward_cpu = ref_im.forward_cpu
    im.forward_multi_thread_cpu = ref_im.forward_multi_thread_cpu
    im.forward_cuda_shared_mem = lambda d, i, K: ref_im.forward_cpu(d.contiguous(), i.contiguous(), K)
    im.forward_cuda = im.forward_cuda_shared_mem
    sys.modules["index_max"] = im
    bq = types.ModuleType("ball_query")
    bq.forward_cuda_shared_mem = lambda dist, r, K: torch.from_numpy(
        native.ball_query(dist.detach().contiguous().numpy(), float(r), int(K)))
    bq.forward_cuda = bq.forward_cuda_shared_mem
    sys.modules["ball_query"] = bq
    sys.path.insert(0, "/root/reference")
    from models import networks, losses, layers   # noqa
    from util import som                           # noqa
    return ref_im, networks, losses, layers, som


class Opt:
    """Attribute bag standing in for the argparse namespace (kitti/options_detector.py:14-60)."""
    def __init__(self, **kw):
        self.activation = "relu"
        self.normalization = "batch"
        self.bn_momentum = 0.1
        self.bn_momentum_decay_step = None
        self.bn_momentum_decay = 0.6
        self.k = 1
        self.__dict__.update(kw)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    if os.path.exists(path) and "--allow-changes" not in sys.argv:
        # regenerating must not move what earlier rounds pinned: every key already in the fixture stays bit-equal
        old = np.load(path, allow_pickle=False)
        for k in old.files:
            new = np.asarray(arrays[k])
            assert k in arrays and np.array_equal(old[k], new, equal_nan=new.dtype.kind == "f"), (name, k)
    np.savez_compressed(path, **arrays)
    print("%-34s %8.1f KB" % (name, os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- native ops
def gen_index_max(ref_im):
    rng = np.random.default_rng(101)
    cases = {}

    def add(tag, data, index, K):
        d, i = torch.from_numpy(data), torch.from_numpy(index)
        out = ref_im.forward_cpu(d, i, K).numpy()
        out_mt = ref_im.forward_multi_thread_cpu(d, i, K, 3).numpy()
        assert np.array_equal(out, out_mt)       # the reference's only parity statement
        cases[tag + "_data"], cases[tag + "_index"] = data, index
        cases[tag + "_K"], cases[tag + "_out"] = np.int32(K), out

    B, C, N, K = 3, 7, 1000, 37
    add("random", rng.normal(0, 1, (B, C, N)).astype(np.float32),
        rng.integers(0, K, (B, N)).astype(np.int32), K)
    # ties: quantised values -> lowest n among the maxima must win
    add("ties", rng.integers(-3, 4, (B, C, N)).astype(np.float32),
        rng.integers(0, K, (B, N)).astype(np.int32), K)
    # floor: values at / below -1000 never win (strict >), all-below-floor node -> 0
    d = rng.normal(-1000, 2, (B, C, N)).astype(np.float32)
    d[:, :, ::7] = -1000.0
    add("floor", d, rng.integers(0, K, (B, N)).astype(np.int32), K)
    # empty nodes: only even nodes are ever assigned
    add("empty", rng.normal(0, 1, (B, C, N)).astype(np.float32),
        (2 * rng.integers(0, K // 2, (B, N))).astype(np.int32), K)
    # NaN never wins
    d = rng.normal(0, 1, (2, 3, 200)).astype(np.float32)
    d[:, :, 5::11] = np.nan
    add("nan", d, rng.integers(0, 8, (2, 200)).astype(np.int32), 8)
    # ragged / tiny
    add("tiny", rng.normal(0, 1, (1, 1, 1)).astype(np.float32), np.zeros((1, 1), np.int32), 1)
    add("n_lt_wave", rng.normal(0, 1, (2, 2, 13)).astype(np.float32),
        rng.integers(0, 5, (2, 13)).astype(np.int32), 5)
    save("index_max_cases.npz", **cases)


def gen_dist_ball():
    """torch.norm distance matrices (pinned: torch CPU) + ball_query rows from the C
    restatement (UNPINNED: no executable reference exists for ball_query)."""
    rng = np.random.default_rng(202)
    B, M, N, K = 2, 48, 1500, 16
    x = synth.make_cloud(rng, N, "slab:12")[None].repeat(B, 0)
    x[1] = synth.make_cloud(rng, N, "slab:12")
    node = np.stack([x[b][:, rng.permutation(N)[:M]] for b in range(B)])
    node[0, :, 0] = 1000.0                        # a node with an empty ball
    tx, tn = torch.from_numpy(x), torch.from_numpy(node)
    dist = torch.norm(tn.unsqueeze(3) - tx.unsqueeze(2), dim=1).numpy()
    out, prefix = native.ball_query(dist, 2.0, K, return_prefix=True)
    hits = (dist <= 2.0).sum(-1)
    assert (hits == 0).any() and ((hits > 0) & (hits < K)).any() and (hits >= K).any()
    save("dist_ball_cases.npz", x=x, node=node, dist=dist, radius=np.float32(2.0), K=np.int32(K),
         ball_idx_unpinned=out, prefix_len_unpinned=prefix)


def _numba_ancestor_ball_query():
    """The reference's only host-runnable statement of ball_query: the numba-CUDA kernel it replaced with the
    C++/CUDA extension and left, commented out, at models/operations.py:295-329.  Its text is read from the
    reference checkout, un-commented in memory and executed with a thread-index shim standing in for
    numba.cuda (one Python call per (block m, thread b)); nothing of it is stored -- only the numbers it
    produces.  It divides by the hit count (`i % unique_idx`), so it is undefined for an empty ball; those rows
    are covered by the CUDA kernel's rule (ball_query_cuda.cu:40-45: all zeros) and flagged in the fixture."""
    src = open("/root/reference/models/operations.py").read().split("\n")
    start = next(i for i, ln in enumerate(src) if ln.startswith("# def ball_query("))
    end = next(i for i, ln in enumerate(src) if ln.startswith("# def ball_query_wrapper("))
    body = "\n".join(ln[2:] if ln.startswith("# ") else ln.lstrip("#") for ln in src[start:end])
    cuda = types.SimpleNamespace(blockIdx=types.SimpleNamespace(x=0), threadIdx=types.SimpleNamespace(x=0),
                                 shared=types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype)))
    numba = types.SimpleNamespace(cuda=cuda, int32=np.int32, float32=np.float32)
    scope = {"numba": numba}
    exec(compile(body, "operations.py:ball_query (numba ancestor)", "exec"), scope)
    kernel = scope["ball_query"]

    def run(dist, radius, nsamples):
        B, M, _ = dist.shape
        assert B <= 32                                        # the kernel's shared array has 32 slots
        out = np.zeros((B, M, nsamples), np.int32)
        empty = np.zeros((B, M), bool)
        for m in range(M):
            for b in range(B):
                if not (dist[b, m] <= np.float32(radius)).any():
                    empty[b, m] = True                        # undefined in the ancestor (modulo by zero)
                    continue
                cuda.blockIdx.x, cuda.threadIdx.x = m, b
                kernel(dist, out, np.float32(radius), nsamples)
        return out, empty
    return run


def gen_ball_query_ancestor():
    """ball_query rows produced by the reference's own (numba) kernel text on seeded distance matrices:
    partially filled balls (cyclic padding), full balls (first K hits), ties at exactly the radius."""
    run = _numba_ancestor_ball_query()
    rng = np.random.default_rng(909)
    cases = {}
    seen = np.zeros(3, bool)                                  # empty / partially filled / full balls
    for name, (B, M, N, K, r) in dict(a=(2, 40, 700, 16, 2.0), b=(3, 17, 333, 64, 2.5), c=(1, 9, 64, 5, 3.0)).items():
        x = np.stack([synth.make_cloud(rng, N, "slab:10") for _ in range(B)])
        node = np.stack([x[b][:, rng.permutation(N)[:M]] for b in range(B)])
        node[0, :, 0] = 500.0                                 # an empty ball
        dist = torch.norm(torch.from_numpy(node).unsqueeze(3) - torch.from_numpy(x).unsqueeze(2), dim=1).numpy()
        dist[0, 1, 7] = np.float32(r)                         # a hit exactly on the radius (<= is inclusive)
        out, empty = run(dist, r, K)
        hits = (dist <= np.float32(r)).sum(-1)
        seen = seen | np.array([empty.any(), ((hits > 0) & (hits < K)).any(), (hits >= K).any()])
        cases.update({name + "_dist": dist, name + "_radius": np.float32(r), name + "_K": np.int32(K),
                      name + "_idx": out, name + "_empty": empty})
    assert seen.all(), seen
    save("ball_query_ancestor_cases.npz", **cases)


def gen_som(som):
    rng = np.random.default_rng(303)
    B, N, M = 3, 900, 40
    x = np.stack([synth.make_cloud(rng, N, "sphere") for _ in range(B)])
    node = np.stack([x[b][:, rng.permutation(N)[:M]] for b in range(B)])
    node[1, :, 3] = 50.0                           # a node no point is nearest to
    mask, mask_row_max, min_idx = som.query_topk(torch.from_numpy(node), torch.from_numpy(x), M, k=1)
    save("som_cases.npz", x=x, node=node, min_idx=min_idx.numpy().astype(np.int32),
         mask_row_max=mask_row_max.numpy().astype(np.int32),
         count=mask.sum(dim=1).numpy().astype(np.int32))


# --------------------------------------------------------------------------- modules
def load_filled(module, head_std=0.05):
    sd = module.state_dict()
    filled = synth.fill_parameters({k: tuple(v.shape) for k, v in sd.items()}, head_std)
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(sd[k].shape)
                            for k, v in filled.items()})


def grad_digest(module):
    """Per-parameter gradient digests: first 48 flat entries, L2 norm, sum, and four seeded +-1 projections
    (usip_amd.synth.grad_projections: every entry of the gradient enters each of them, so a digest of 4 numbers
    pins the whole tensor without storing 1.2 M floats per fixture)."""
    d = {}
    for k, p in module.named_parameters():
        g = p.grad.detach().numpy().ravel()
        d["grad_head/" + k] = g[:48].copy()
        d["grad_norm/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        d["grad_sum/" + k] = np.float64(g.astype(np.float64).sum())
        d["grad_proj/" + k] = synth.grad_projections(k, g)
    return d


def gen_layers(layers):
    rng = np.random.default_rng(404)
    out = {}
    # MyConv2d 1x1 + BN + ReLU, train mode (layers.py:172-216)
    conv = layers.MyConv2d(7, 12, kernel_size=(1, 1), stride=1, padding=0, bias=True,
                           activation="relu", normalization="batch", momentum=0.1)
    load_filled(conv)
    conv.train()
    x = torch.from_numpy(rng.normal(0, 1, (3, 7, 10, 6)).astype(np.float32)).requires_grad_(True)
    gy = torch.from_numpy(rng.normal(0, 1, (3, 12, 10, 6)).astype(np.float32))
    y = conv(x)
    y.backward(gy)
    out.update(conv2d_x=x.detach().numpy(), conv2d_gy=gy.numpy(), conv2d_y=y.detach().numpy(),
               conv2d_gx=x.grad.numpy(), conv2d_gw=conv.conv.weight.grad.numpy(),
               conv2d_gb=conv.conv.bias.grad.numpy(), conv2d_ggamma=conv.norm.weight.grad.numpy(),
               conv2d_gbeta=conv.norm.bias.grad.numpy(),
               conv2d_running_mean=conv.norm.running_mean.numpy(),
               conv2d_running_var=conv.norm.running_var.numpy())
    conv.eval()
    out["conv2d_y_eval"] = conv(x.detach()).detach().numpy()
    # EquivariantLayer (layers.py:248-303)
    eq = layers.EquivariantLayer(6, 9, activation="relu", normalization="batch", momentum=0.1)
    load_filled(eq)
    eq.train()
    x1 = torch.from_numpy(rng.normal(0, 1, (4, 6, 50)).astype(np.float32)).requires_grad_(True)
    g1 = torch.from_numpy(rng.normal(0, 1, (4, 9, 50)).astype(np.float32))
    y1 = eq(x1)
    y1.backward(g1)
    out.update(eq_x=x1.detach().numpy(), eq_gy=g1.numpy(), eq_y=y1.detach().numpy(),
               eq_gx=x1.grad.numpy(), eq_gw=eq.conv.weight.grad.numpy())
    # GeneralKNNFusionModule (layers.py:375-440), small widths
    knn = layers.GeneralKNNFusionModule(3 + 8, (16, 16), (24, 24), activation="relu",
                                        normalization="batch", momentum=0.1)
    load_filled(knn)
    knn.train()
    q = torch.from_numpy(np.stack([synth.make_cloud(rng, 30, "sphere") for _ in range(2)]))
    f = torch.from_numpy(rng.normal(0, 1, (2, 8, 30)).astype(np.float32)).requires_grad_(True)
    gk = torch.from_numpy(rng.normal(0, 1, (2, 24, 30)).astype(np.float32))
    yk = knn(query=q, database=q, x=f, K=5)
    yk.backward(gk)
    out.update(knn_q=q.numpy(), knn_f=f.detach().numpy(), knn_gy=gk.numpy(), knn_y=yk.detach().numpy(),
               knn_gf=f.grad.numpy(), knn_gw0=knn.layers_before[0].conv.weight.grad.numpy(),
               knn_gw_after0=knn.layers_after[0].conv.weight.grad.numpy())
    save("layers_cases.npz", **out)


def gen_bn_decay(layers):
    """a-13: epoch-driven BatchNorm momentum (layers.py:61-71, :112-121): momentum = momentum0 *
    decay ** (epoch // step) once epoch >= 1, clamped below at 0.01.  Two training forwards per epoch value
    through MyConv2d and EquivariantLayer with decay_step=2, decay=0.6; the fixture holds the momentum the
    reference module ends up with and its running statistics after every call."""
    rng = np.random.default_rng(1313)
    out = {}
    x2 = rng.normal(0.3, 1.2, (2, 3, 5, 7, 4)).astype(np.float32)        # [call, B, C, M, K]
    x1 = rng.normal(-0.2, 0.8, (2, 3, 6, 33)).astype(np.float32)         # [call, B, C, N]
    epochs = [None, 0, 1, 2, 5, 9, 40]
    out.update(x2=x2, x1=x1, epochs=np.asarray([-1 if e is None else e for e in epochs], dtype=np.int32))
    for e in epochs:
        tag = "none" if e is None else str(e)
        conv = layers.MyConv2d(5, 8, kernel_size=(1, 1), stride=1, padding=0, bias=True, activation="relu",
                               normalization="batch", momentum=0.1, bn_momentum_decay_step=2, bn_momentum_decay=0.6)
        eq = layers.EquivariantLayer(6, 10, activation="relu", normalization="batch", momentum=0.1,
                                     bn_momentum_decay_step=2, bn_momentum_decay=0.6)
        load_filled(conv)
        load_filled(eq)
        conv.train()
        eq.train()
        for call in range(2):
            y2 = conv(torch.from_numpy(x2[call]), e)
            y1 = eq(torch.from_numpy(x1[call]), e)
            out["conv_rm_%s_%d" % (tag, call)] = conv.norm.running_mean.numpy().copy()
            out["conv_rv_%s_%d" % (tag, call)] = conv.norm.running_var.numpy().copy()
            out["eq_rm_%s_%d" % (tag, call)] = eq.norm.running_mean.numpy().copy()
            out["eq_rv_%s_%d" % (tag, call)] = eq.norm.running_var.numpy().copy()
        out["conv_y_%s" % tag] = y2.detach().numpy()
        out["eq_y_%s" % tag] = y1.detach().numpy()
        out["momentum_%s" % tag] = np.float64(conv.norm.momentum)
        assert conv.norm.momentum == eq.norm.momentum
    assert out["momentum_40"] == 0.01 and out["momentum_none"] == 0.1 and out["momentum_1"] == 0.1
    save("bn_decay_cases.npz", **out)


def gen_losses(losses):
    rng = np.random.default_rng(505)
    opt = Opt()
    B, M, N = 3, 40, 300
    src = torch.from_numpy(rng.normal(0, 1, (B, 3, M)).astype(np.float32)).requires_grad_(True)
    dst = torch.from_numpy(rng.normal(0, 1, (B, 3, M + 7)).astype(np.float32)).requires_grad_(True)
    ss = torch.from_numpy(rng.uniform(0.05, 1.5, (B, M)).astype(np.float32)).requires_grad_(True)
    sd = torch.from_numpy(rng.uniform(0.05, 1.5, (B, M + 7)).astype(np.float32)).requires_grad_(True)
    loss, pure, weighted = losses.ChamferLoss_Brute(opt)(src, dst, ss, sd)
    loss.backward()
    out = dict(pc_src=src.detach().numpy(), pc_dst=dst.detach().numpy(), pc_ss=ss.detach().numpy(),
               pc_sd=sd.detach().numpy(), pc_loss=loss.detach().numpy(), pc_pure=pure.numpy(),
               pc_weighted=weighted.numpy(), pc_gsrc=src.grad.numpy(), pc_gdst=dst.grad.numpy(),
               pc_gss=ss.grad.numpy(), pc_gsd=sd.grad.numpy())
    kp = torch.from_numpy(rng.normal(0, 1, (B, 3, M)).astype(np.float32))
    pc = torch.from_numpy(rng.normal(0, 1, (B, 3, N)).astype(np.float32))
    kp[0, :, 0] = pc[0, :, 5]                       # exact coincidence: zero sub-gradient of norm
    kp.requires_grad_(True)
    d = losses.KeypointOnPCLoss(opt)(kp, pc, None)
    gd = torch.from_numpy(rng.normal(0, 1, (B, M)).astype(np.float32))
    d.backward(gd)
    out.update(ss_kp=kp.detach().numpy(), ss_pc=pc.numpy(), ss_d=d.detach().numpy(),
               ss_gd=gd.numpy(), ss_gkp=kp.grad.numpy())
    save("losses_cases.npz", **out)


def gen_point_to_plane(losses):
    """Round 4: the keypoint-on-surface form of KeypointOnPCLoss (models/losses.py:146-187, selected by
    opt.keypoint_on_pc_type == 'point_to_plane' at keypoint_detector.py:197-201): (n . (kp - p) / (|kp - p| + 1e-7))^2 for
    the nearest cloud point p with normal n.  sn carries 4 channels as in KITTI (normal + curvature); the reference
    gathers the first three."""
    rng = np.random.default_rng(606)
    opt = Opt()
    B, M, N = 3, 40, 300
    kp = torch.from_numpy(rng.normal(0, 1, (B, 3, M)).astype(np.float32))
    pc = torch.from_numpy(rng.normal(0, 1, (B, 3, N)).astype(np.float32))
    nrm = rng.normal(0, 1, (B, 3, N))
    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    sn = torch.from_numpy(np.concatenate((nrm, rng.uniform(0, 1, (B, 1, N))), 1).astype(np.float32))
    kp[0, :, 0] = pc[0, :, 5]                       # exact coincidence: 0 / (0 + 1e-7)
    kp.requires_grad_(True)
    loss = losses.KeypointOnPCLoss(opt)(kp, pc, sn)            # B x M x 1 x 1
    g = torch.from_numpy(rng.normal(0, 1, tuple(loss.shape)).astype(np.float32))
    loss.backward(g)
    save("point_to_plane_cases.npz", kp=kp.detach().numpy(), pc=pc.numpy(), sn=sn.numpy(), loss=loss.detach().numpy(),
         g=g.numpy(), gkp=kp.grad.numpy())


def run_step(net, losses_mod, opt, batch, alpha):
    """ModelDetector.optimize without the Adam update, driven on the reference's modules
    directly (ModelDetector itself calls torch.cuda.synchronize(), keypoint_detector.py:134)."""
    t = {k: torch.from_numpy(v) for k, v in batch.items()}
    B = t["src_pc"].shape[0]
    net.train()
    node_r, kp, sg, _ = net(torch.cat((t["src_pc"], t["dst_pc"]), 0),
                            torch.cat((t["src_sn"], t["dst_sn"]), 0),
                            torch.cat((t["src_node"], t["dst_node"]), 0), True, None)
    kp_s, kp_d = kp[:B], kp[B:]
    kp_t = torch.matmul(t["R"], kp_s)
    kp_t = kp_t * t["scale"].unsqueeze(1).unsqueeze(2)
    kp_t = kp_t + t["shift"]
    net.zero_grad()
    lc, pure, weighted = losses_mod.ChamferLoss_Brute(opt)(kp_t, kp_d, sg[:B], sg[B:])
    crit = losses_mod.KeypointOnPCLoss(opt)
    l_src = torch.mean(crit(kp_s, t["src_pc"], None)) * alpha
    l_dst = torch.mean(crit(kp_d, t["dst_pc"], None)) * alpha
    loss = lc + l_src + l_dst
    loss.backward()
    out = dict(node=node_r.detach().numpy(), keypoints=kp.detach().numpy(), sigmas=sg.detach().numpy(),
               loss=loss.detach().numpy(), loss_chamfer=lc.detach().numpy(), chamfer_pure=pure.numpy(),
               chamfer_weighted=weighted.numpy(), loss_on_pc_src=l_src.detach().numpy(),
               loss_on_pc_dst=l_dst.detach().numpy())
    out.update(grad_digest(net))
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = v.numpy().copy()
    return out


RELU_NEAR = 1e-4       # |BatchNorm output| below which a ReLU decision is stored in the fixture


def capture_indices(networks_mod, som_mod):
    """Record the index tensors the forward computes, by wrapping the shim entry points."""
    rec = {}
    im, bq = sys.modules["index_max"], sys.modules["ball_query"]
    orig_im, orig_bq = im.forward_cuda_shared_mem, bq.forward_cuda_shared_mem
    orig_topk = torch.topk

    def im_wrap(d, i, K):
        r = orig_im(d, i, K)
        rec["index_max_%d" % sum(k.startswith("index_max_") for k in rec)] = r.numpy().copy()
        rec["min_idx"] = i.numpy().copy()
        return r

    def bq_wrap(d, r_, K):
        r = orig_bq(d, r_, K)
        rec["ball_idx"] = r.numpy().copy()
        return r

    def topk_wrap(*a, **kw):
        r = orig_topk(*a, **kw)
        if kw.get("sorted", True) and kw.get("largest", True) is False:
            rec["knn_I"] = r[1].numpy().astype(np.int32).copy()
        elif kw.get("largest", True) is False and kw.get("k", 1) > 1 and a[0].dim() == 3:
            # RPN_Detector_KNN's point neighbourhoods (networks.py:581): topk(sorted=False) leaves the ORDER of the
            # k picks unspecified, so the fixture pins the SET, written nearest first with ties towards the lower
            # index; `_nn_pos` maps a position in the reference's order to the position in that canonical order
            # (used below to re-express the reference's max-pool / ReLU decisions, which are positions in ITS order)
            k = kw["k"]
            canon = torch.sort(a[0], dim=2, stable=True)[1][:, :, :k]
            ref = r[1]
            assert torch.equal(torch.sort(ref, dim=2)[0], torch.sort(canon, dim=2)[0]), "tie at the k-th distance"
            rec["nn_idx"] = canon.numpy().astype(np.int32).copy()
            rec["_nn_pos"] = (ref.unsqueeze(3) == canon.unsqueeze(2)).float().argmax(3).numpy().astype(np.int64)
        return r

    orig_max = torch.max

    def max_wrap(*a, **kw):
        # the four max-pools over the K neighbours (networks.py:706,710, layers.py:433,438): torch.max(x4d, dim=3)
        r = orig_max(*a, **kw)
        if kw.get("dim") == 3 and len(a) == 1 and a[0].dim() == 4:
            arg = r[1].reshape(a[0].shape[0], a[0].shape[1], a[0].shape[2])
            assert a[0].shape[3] <= 127
            rec["pool_arg_%d" % sum(k.startswith("pool_arg_") for k in rec)] = arg.numpy().astype(np.int8).copy()
        return r
    orig_relu = torch.nn.ReLU.forward

    def relu_wrap(self, x):
        # every ReLU of the path sits behind a BatchNorm (layers.py:167,215,301): x is O(1).  Record the decisions
        # that another correct fp32 forward could take differently: pre-activations within 1e-4 of zero
        # (flat index, on/off).  Everything farther from zero is unambiguous (forwards agree to ~1e-6).
        i = sum(k.startswith("relu_near_idx_") for k in rec)
        flat = x.detach().reshape(-1)
        idx = torch.nonzero(flat.abs() < RELU_NEAR, as_tuple=False).reshape(-1)
        rec["relu_near_idx_%d" % i] = idx.numpy().astype(np.int32)
        rec["relu_near_on_%d" % i] = (flat[idx] > 0).numpy()
        rec["relu_numel_%d" % i] = np.int64(flat.numel())
        return orig_relu(self, x)
    im.forward_cuda_shared_mem, bq.forward_cuda_shared_mem = im_wrap, bq_wrap
    torch.topk = topk_wrap
    torch.max = max_wrap
    torch.nn.ReLU.forward = relu_wrap

    def restore():
        im.forward_cuda_shared_mem, bq.forward_cuda_shared_mem = orig_im, orig_bq
        torch.topk = orig_topk
        torch.max = orig_max
        torch.nn.ReLU.forward = orig_relu
    return rec, restore


def reorder_neighbour_decisions(rec, n_layers):
    """RPN_Detector_KNN: the reference's decisions over its own (unspecified) neighbour order, re-expressed over the
    canonical order of idx/nn_idx: the first two max-pools (over the point neighbourhoods) and the near-zero ReLU
    lists of conv1..conv5."""
    pos = rec.pop("_nn_pos")                                  # [B,M,K]: canonical position of the reference's k-th pick
    B, M, K = pos.shape
    for i in range(2):
        arg = rec["pool_arg_%d" % i].astype(np.int64)        # [B,C,M]
        rec["pool_arg_%d" % i] = np.take_along_axis(pos[:, None].repeat(arg.shape[1], 1), arg[..., None], 3)[..., 0] \
            .astype(np.int8)
    for i in range(n_layers):
        flat = rec["relu_near_idx_%d" % i].astype(np.int64)
        C = int(rec["relu_numel_%d" % i]) // (B * M * K)
        b, c, m, k = np.unravel_index(flat, (B, C, M, K))
        new = np.ravel_multi_index((b, c, m, pos[b, m, k]), (B, C, M, K))
        order = np.argsort(new, kind="stable")
        rec["relu_near_idx_%d" % i] = new[order].astype(np.int32)
        rec["relu_near_on_%d" % i] = rec["relu_near_on_%d" % i][order]


def gen_detectors_r3(networks, losses, som):
    """Round 3: the reference's other two detector call sequences on this surface -- RPN_DetectorLite
    (networks.py:165-307: what every indoor train_detector.py builds) and RPN_Detector_KNN (:482-608)."""
    opt = Opt(surface_normal_len=3, node_knn_k_1=16, loss_sigma_lower_bound=1e-4)
    batch = synth.make_pair_batch(seed=3131, pairs=2, n=1280, m=48, cs=3, kind="sphere")
    net = networks.RPN_DetectorLite(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=1.0)
    restore()
    save("detector_lite_micro.npz", cfg_model="lite", cfg_knn=np.int32(16), cfg_sigma_lb=np.float32(1e-4),
         cfg_alpha=np.float32(1.0), **{"in/" + k: v for k, v in batch.items()},
         **{"idx/" + k: v for k, v in rec.items()}, **out)

    opt = Opt(surface_normal_len=4, node_knn_k_1=16, loss_sigma_lower_bound=1e-3)
    batch = synth.make_pair_batch(seed=5151, pairs=2, n=2048, m=64, cs=4, kind="slab:14")
    net = networks.RPN_Detector_KNN(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=0.01)
    restore()
    reorder_neighbour_decisions(rec, 5)
    save("detector_knn_micro.npz", cfg_model="knn", cfg_knn=np.int32(16), cfg_sigma_lb=np.float32(1e-3),
         cfg_alpha=np.float32(0.01), **{"in/" + k: v for k, v in batch.items()},
         **{"idx/" + k: v for k, v in rec.items()}, **out)


def gen_detectors(networks, losses, som):
    # config 1: RPN_Detector, ModelNet plumbing size (BASELINE.json configs[0])
    opt = Opt(surface_normal_len=3, node_knn_k_1=32, loss_sigma_lower_bound=1e-4)
    batch = synth.make_pair_batch(seed=1234, pairs=2, n=1024, m=64, cs=3, kind="sphere")
    net = networks.RPN_Detector(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=1.0)
    restore()
    save("detector_som_cfg1.npz", cfg_model="som", cfg_knn=np.int32(32), cfg_sigma_lb=np.float32(1e-4),
         cfg_alpha=np.float32(1.0), **{"in/" + k: v for k, v in batch.items()},
         **{"idx/" + k: v for k, v in rec.items()}, **out)

    # micro config: RPN_Detector_Ball, KITTI-like parameters at small size (ball_idx unpinned)
    opt = Opt(surface_normal_len=4, node_knn_k_1=16, loss_sigma_lower_bound=1e-3)
    batch = synth.make_pair_batch(seed=4321, pairs=2, n=2048, m=64, cs=4, kind="slab:14")
    net = networks.RPN_Detector_Ball(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=0.01)
    restore()
    save("detector_ball_micro.npz", cfg_model="ball", cfg_knn=np.int32(16), cfg_sigma_lb=np.float32(1e-3),
         cfg_alpha=np.float32(0.01), **{"in/" + k: v for k, v in batch.items()},
         **{"idx/" + k: v for k, v in rec.items()}, **out)

    # micro config: RPN_Detector at KITTI-like parameters (Cs=4, Kn=16)
    opt = Opt(surface_normal_len=4, node_knn_k_1=16, loss_sigma_lower_bound=1e-3)
    batch = synth.make_pair_batch(seed=777, pairs=1, n=1536, m=48, cs=4, kind="slab:14")
    net = networks.RPN_Detector(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=0.01)
    restore()
    save("detector_som_micro.npz", cfg_model="som", cfg_knn=np.int32(16), cfg_sigma_lb=np.float32(1e-3),
         cfg_alpha=np.float32(0.01), **{"in/" + k: v for k, v in batch.items()},
         **{"idx/" + k: v for k, v in rec.items()}, **out)


def gen_detectors_options(networks, losses, som):
    """Round 6: the non-default values of --activation / --normalization (models/layers.py:181-193, :262-272; every
    options file of the reference defaults to relu / batch): RPN_Detector_Ball with ELU + instance normalisation and
    RPN_Detector with Swish + batch normalisation, through the same optimize() step."""
    opt = Opt(surface_normal_len=4, node_knn_k_1=8, loss_sigma_lower_bound=1e-3, activation="elu",
              normalization="instance")
    batch = synth.make_pair_batch(seed=6161, pairs=1, n=1024, m=32, cs=4, kind="slab:14")
    net = networks.RPN_Detector_Ball(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=0.01)
    restore()
    rec = {k: v for k, v in rec.items() if not k.startswith("relu_")}
    save("detector_ball_elu_instance.npz", cfg_model="ball", cfg_knn=np.int32(8), cfg_sigma_lb=np.float32(1e-3),
         cfg_alpha=np.float32(0.01), cfg_activation="elu", cfg_normalization="instance",
         **{"in/" + k: v for k, v in batch.items()}, **{"idx/" + k: v for k, v in rec.items()}, **out)

    opt = Opt(surface_normal_len=3, node_knn_k_1=8, loss_sigma_lower_bound=1e-4, activation="swish",
              normalization="batch")
    batch = synth.make_pair_batch(seed=6262, pairs=1, n=1024, m=32, cs=3, kind="sphere")
    net = networks.RPN_Detector(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=1.0)
    restore()
    rec = {k: v for k, v in rec.items() if not k.startswith("relu_")}
    save("detector_som_swish.npz", cfg_model="som", cfg_knn=np.int32(8), cfg_sigma_lb=np.float32(1e-4),
         cfg_alpha=np.float32(1.0), cfg_activation="swish", cfg_normalization="batch",
         **{"in/" + k: v for k, v in batch.items()}, **{"idx/" + k: v for k, v in rec.items()}, **out)

    # --k 2 (util/som.py:49-50, networks.py:85-92): every point goes to its TWO nearest nodes, the cloud is stacked twice.
    # torch.topk(sorted=False) leaves the order of a point's picks unspecified: idx/min_idx is compared as per-point SETS
    # and the index_max positions (positions in the stacked cloud) are not stored.
    opt = Opt(surface_normal_len=3, node_knn_k_1=8, loss_sigma_lower_bound=1e-4, k=2)
    batch = synth.make_pair_batch(seed=6363, pairs=1, n=1024, m=32, cs=3, kind="sphere")
    net = networks.RPN_Detector(opt)
    load_filled(net)
    rec, restore = capture_indices(networks, som)
    out = run_step(net, losses, opt, batch, alpha=1.0)
    restore()
    rec = {k: v for k, v in rec.items() if k in ("min_idx", "knn_I")}
    save("detector_som_k2.npz", cfg_model="som", cfg_knn=np.int32(8), cfg_sigma_lb=np.float32(1e-4),
         cfg_alpha=np.float32(1.0), cfg_k=np.int32(2),
         **{"in/" + k: v for k, v in batch.items()}, **{"idx/" + k: v for k, v in rec.items()}, **out)


def gen_descriptor(networks, losses):
    """DescriptorLiteOld + DescPairScanLoss as ModelDescriptor.optimize drives them
    (models/keypoint_descriptor.py:126-159), with the random point permutation of
    networks.py:345-347 recorded (ball_query picks the FIRST K points inside the ball, so the
    permutation is observable)."""
    rng = np.random.default_rng(606)
    B, N, M, Cs = 2, 2048, 32, 4
    opt = Opt(surface_normal_len=Cs, descriptor_len=128, ball_radius=2, ball_nsamples=64,
              triple_loss_gamma=0.5, sigma_max=3.0)
    anc = np.stack([synth.make_cloud(rng, N, "slab:14") for _ in range(B)])
    pos = np.stack([synth.make_cloud(rng, N, "slab:14") for _ in range(B)])
    anc_sn = np.stack([synth.make_normals(rng, N, Cs) for _ in range(B)])
    pos_sn = np.stack([synth.make_normals(rng, N, Cs) for _ in range(B)])
    anc_kp = np.ascontiguousarray(np.stack([anc[b][:, rng.permutation(N)[:M]] for b in range(B)]))
    pos_kp = np.ascontiguousarray(np.stack([pos[b][:, rng.permutation(N)[:M]] for b in range(B)]))
    anc_kp[0, :, 0] = 500.0                         # a keypoint with an empty ball
    anc_sigmas = rng.uniform(0.1, 3.5, (B, M)).astype(np.float32)
    neg_idx = np.array([1, 0], dtype=np.int64)
    perm = rng.permutation(N).astype(np.int64)
    net = networks.DescriptorLiteOld(opt)
    load_filled(net)
    net.train()
    orig = np.random.permutation
    np.random.permutation = lambda n: perm.copy()
    rec, restore = capture_indices(networks, None)           # round 3: the two max-pool routings + near-zero ReLUs
    try:
        desc, x_feat = net(torch.from_numpy(np.concatenate([anc, pos])), torch.from_numpy(np.concatenate([anc_sn, pos_sn])),
                           torch.from_numpy(np.concatenate([anc_kp, pos_kp])), True, None)
    finally:
        np.random.permutation = orig
        restore()
    anc_d, pos_d = desc[:B], desc[B:]
    net.zero_grad()
    trip, active = losses.DescPairScanLoss(opt)(anc_d, pos_d, anc_d[torch.from_numpy(neg_idx), :, :],
                                                torch.from_numpy(anc_sigmas))
    loss = torch.mean(trip)
    loss.backward()
    out = dict(anc_pc=anc, pos_pc=pos, anc_sn=anc_sn, pos_sn=pos_sn, anc_kp=anc_kp, pos_kp=pos_kp,
               anc_sigmas=anc_sigmas, neg_idx=neg_idx, perm=perm, descriptors=desc.detach().numpy(),
               x_features=x_feat.detach().numpy(), triplet=trip.detach().numpy(), active=active.numpy(),
               loss=loss.detach().numpy())
    out.update(grad_digest(net))
    out.update({"idx/" + k: v for k, v in rec.items() if k.startswith(("pool_arg_", "relu_n"))})
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["buf/" + k] = v.numpy().copy()
    save("descriptor_micro.npz", **out)


def _extract(path, name):
    """Compile ONE top-level function / class of a reference file (found with ast) into a namespace, so the
    reference's own code runs without importing a module whose other imports are absent here."""
    import ast
    src = open(path).read()
    node = next(n for n in ast.parse(src).body if getattr(n, "name", None) == name)
    ns = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def gen_pre_post(networks):
    """f-3 farthest-point sampling and f-4 NMS / top-k: outputs of the reference's own FarthestSampler
    (data/kitti_detector_loader.py:69-83) and nms() (evaluation/save_keypoints.py:180-216), plus an eval-mode
    forward of RPN_Detector (keypoint_detector.py:247-251)."""
    rng = np.random.default_rng(707)
    Sampler = _extract("/root/reference/data/kitti_detector_loader.py", "FarthestSampler")
    nms = _extract("/root/reference/evaluation/save_keypoints.py", "nms")
    out = {}
    # FPS: 3 clouds, n = 1800 points, k = 96 nodes; the reference draws the first index with np.random.randint
    B, n, k = 3, 1800, 96
    pts = np.stack([synth.make_cloud(rng, n, "slab:20").T.copy() for _ in range(B)])       # [B,n,3]
    first = rng.integers(0, n, B)
    orig = np.random.randint
    sel = []
    for b in range(B):
        np.random.randint = lambda *a, **kw: int(first[b])
        try:
            far = Sampler().sample(pts[b], k)                                              # [k,3] float64
        finally:
            np.random.randint = orig
        idx = [int(np.nonzero((pts[b].astype(np.float64) == far[i]).all(1))[0][0]) for i in range(k)]
        sel.append(idx)
    out.update(fps_pts=pts, fps_first=first.astype(np.int32), fps_idx=np.asarray(sel, dtype=np.int32))
    # NMS + top-k: 2 clouds of 256 keypoints
    M = 256
    kp = np.stack([synth.make_cloud(rng, M, "slab:12").T.copy() for _ in range(2)])        # [2,M,3]
    sg = rng.uniform(0.01, 2.0, (2, M)).astype(np.float32)
    sg[0, 10] = sg[0, 3]                                                                   # a sigma tie
    for b in range(2):
        vk, vs = nms(kp[b], sg[b], 2.0)
        out["nms_kept_%d" % b] = vk.astype(np.float32)
        out["nms_sigma_%d" % b] = vs.astype(np.float32)
        order = np.argsort(vs)[:40]
        out["nms_top40_%d" % b] = vk[order].astype(np.float32)
    out.update(nms_kp=kp, nms_sigma=sg, nms_radius=np.float32(2.0))
    # eval-mode forward
    opt = Opt(surface_normal_len=3, node_knn_k_1=8, loss_sigma_lower_bound=1e-3)
    batch = synth.make_pair_batch(seed=808, pairs=1, n=1024, m=32, cs=3, kind="sphere")
    net = networks.RPN_Detector(opt)
    load_filled(net)
    net.eval()
    with torch.no_grad():
        _, kpt, sig, _ = net(torch.from_numpy(batch["src_pc"]), torch.from_numpy(batch["src_sn"]),
                             torch.from_numpy(batch["src_node"]), False, None)
    out.update(eval_pc=batch["src_pc"], eval_sn=batch["src_sn"], eval_node=batch["src_node"],
               eval_keypoints=kpt.numpy(), eval_sigmas=sig.numpy())
    save("pre_post_cases.npz", **out)


if __name__ == "__main__":
    ref_im, networks, losses, layers, som = import_reference()
    if "--only-ball-ancestor" in sys.argv:
        gen_ball_query_ancestor()
        sys.exit(0)
    if "--only-prepost" in sys.argv:
        gen_pre_post(networks)
        sys.exit(0)
    if "--only-descriptor" in sys.argv:
        gen_descriptor(networks, losses)
        sys.exit(0)
    if "--only-detectors" in sys.argv:
        gen_detectors(networks, losses, som)
        sys.exit(0)
    if "--only-detectors-options" in sys.argv:
        gen_detectors_options(networks, losses, som)
        sys.exit(0)
    if "--only-detectors-r3" in sys.argv:
        gen_detectors_r3(networks, losses, som)
        sys.exit(0)
    if "--only-point-to-plane" in sys.argv:
        gen_point_to_plane(losses)
        sys.exit(0)
    if "--only-bn-decay" in sys.argv:
        gen_bn_decay(layers)
        sys.exit(0)
    gen_index_max(ref_im)
    gen_dist_ball()
    gen_som(som)
    gen_layers(layers)
    gen_bn_decay(layers)
    gen_losses(losses)
    gen_point_to_plane(losses)
    gen_detectors(networks, losses, som)
    gen_detectors_r3(networks, losses, som)
    gen_detectors_options(networks, losses, som)
    gen_descriptor(networks, losses)
    gen_pre_post(networks)
