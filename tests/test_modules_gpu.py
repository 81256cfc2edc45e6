"""GPU parity of the nn.Module surface and of the whole detector step against the golden
vectors captured from the reference and against the oracle.

Float bar: 1e-5 relative (conftest.assert_close).  Index tensors: bit-exact.
Whole-step GRADIENTS: test_detector_step_gradients_match_reference_with_pinned_decisions asserts
every parameter gradient at 1e-5 with the forward's discrete decisions (max-pool arg-max, ReLU
on/off) taken from the reference; test_detector_step_matches_reference keeps the free-running
comparison with a flip-tolerant bound, for the reason DESIGN.md ("gradient parity") documents
with numbers: a rounding-level change anywhere in the forward flips a few of those decisions and
moves gradient entries by ~1e-2 -- the reference's own fp32 run sits that far from its fp64 run."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_(True) if grad else t


def _load_filled(module):
    from usip_amd import synth
    sd = module.state_dict()
    filled = synth.fill_parameters({k: tuple(v.shape) for k, v in sd.items()})
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(sd[k].shape) for k, v in filled.items()})
    return module.to(DEV)


def test_myconv2d_matches_reference_fwd_bwd_and_buffers():
    from usip_amd import layers
    g = load_golden("layers_cases.npz")
    conv = _load_filled(layers.MyConv2d(7, 12, kernel_size=(1, 1), stride=1, padding=0, bias=True,
                                        activation="relu", normalization="batch", momentum=0.1))
    conv.train()
    x = _t(g["conv2d_x"], True)
    y = conv(x)
    y.backward(_t(g["conv2d_gy"]))
    assert_close(y.detach().cpu(), g["conv2d_y"], name="y")
    assert_close(x.grad.cpu(), g["conv2d_gx"], name="gx")
    assert_close(conv.conv.weight.grad.cpu(), g["conv2d_gw"], name="gw")
    assert_close(conv.norm.weight.grad.cpu(), g["conv2d_ggamma"], name="ggamma")
    assert_close(conv.norm.bias.grad.cpu(), g["conv2d_gbeta"], name="gbeta")
    assert_close(conv.norm.running_mean.cpu(), g["conv2d_running_mean"], name="running_mean")
    assert_close(conv.norm.running_var.cpu(), g["conv2d_running_var"], name="running_var")
    conv.eval()
    assert_close(conv(x.detach()).detach().cpu(), g["conv2d_y_eval"], name="y_eval")


def test_equivariant_layer_matches_reference():
    from usip_amd import layers
    g = load_golden("layers_cases.npz")
    eq = _load_filled(layers.EquivariantLayer(6, 9, activation="relu", normalization="batch", momentum=0.1))
    eq.train()
    x = _t(g["eq_x"], True)
    y = eq(x)
    y.backward(_t(g["eq_gy"]))
    assert_close(y.detach().cpu(), g["eq_y"], name="y")
    assert_close(x.grad.cpu(), g["eq_gx"], name="gx")
    assert_close(eq.conv.weight.grad.cpu(), g["eq_gw"], name="gw")


def test_knn_fusion_module_matches_reference():
    from usip_amd import layers
    g = load_golden("layers_cases.npz")
    knn = _load_filled(layers.GeneralKNNFusionModule(3 + 8, (16, 16), (24, 24), activation="relu",
                                                     normalization="batch", momentum=0.1))
    knn.train()
    q = _t(g["knn_q"])
    f = _t(g["knn_f"], True)
    y = knn(query=q, database=q, x=f, K=5)
    y.backward(_t(g["knn_gy"]))
    assert_close(y.detach().cpu(), g["knn_y"], name="y")
    assert_close(f.grad.cpu(), g["knn_gf"], rel=5e-5, name="gf")
    assert_close(knn.layers_before[0].conv.weight.grad.cpu(), g["knn_gw0"], rel=5e-5, name="gw0")
    assert_close(knn.layers_after[0].conv.weight.grad.cpu(), g["knn_gw_after0"], rel=5e-5, name="gw_after0")


def test_losses_match_reference():
    from usip_amd import losses
    from usip_amd.networks import DetectorOptions
    g = load_golden("losses_cases.npz")
    opt = DetectorOptions()
    src, dst = _t(g["pc_src"], True), _t(g["pc_dst"], True)
    ss, sd = _t(g["pc_ss"], True), _t(g["pc_sd"], True)
    loss, pure, weighted = losses.ChamferLoss_Brute(opt)(src, dst, ss, sd)
    loss.backward()
    assert_close(loss.detach().cpu(), g["pc_loss"], name="loss")
    assert_close(pure.detach().cpu(), g["pc_pure"], name="pure")
    assert_close(weighted.detach().cpu(), g["pc_weighted"], name="weighted")
    for got, key in ((src.grad, "pc_gsrc"), (dst.grad, "pc_gdst"), (ss.grad, "pc_gss"), (sd.grad, "pc_gsd")):
        assert_close(got.cpu(), g[key], name=key)
    kp = _t(g["ss_kp"], True)
    d = losses.KeypointOnPCLoss(opt)(kp, _t(g["ss_pc"]), None)
    d.backward(_t(g["ss_gd"]))
    assert np.array_equal(d.detach().cpu().numpy(), g["ss_d"])           # same arithmetic order: bit-exact
    assert_close(kp.grad.cpu(), g["ss_gkp"], name="ss_gkp")


def test_point_to_plane_loss_matches_reference():
    """KeypointOnPCLoss(kp, pc, sn) -- the point-to-plane form (losses.py:146-187) on the fused nearest-neighbour kernel:
    values and the keypoint gradient against the reference's fixture (incl. a keypoint ON a cloud point: 0 / (0 + 1e-7))."""
    from usip_amd.losses import KeypointOnPCLoss
    from usip_amd.networks import DetectorOptions
    g = load_golden("point_to_plane_cases.npz")
    kp = _t(g["kp"], grad=True)
    crit = KeypointOnPCLoss(DetectorOptions())
    loss = crit(kp, _t(g["pc"]), _t(g["sn"]))
    assert tuple(loss.shape) == tuple(g["loss"].shape)
    loss.backward(_t(g["g"]))
    assert_close(loss.detach().cpu().numpy(), g["loss"], name="point-to-plane loss")
    assert_close(kp.grad.cpu().numpy(), g["gkp"], name="point-to-plane d/dkp")


def test_step_with_point_to_plane_and_point_dropout_matches_oracle():
    """ModelDetector.optimize with its two non-default switches (keypoint_detector.py:160-168 random point dropout,
    :197-201 point_to_plane): the chosen point indices travel with the batch (the reference draws them with numpy on the
    host), so HIP step and oracle drop the same points; indices bit-exact, floats 1e-5, flip-tolerant gradient norms."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions, detector_param_shapes
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8, keypoint_on_pc_type="point_to_plane",
                          random_pc_dropout_lower_limit=0.7, input_pc_num=2048)
    batch_np = synth.make_pair_batch(77, 2, 2048, 48, 4, "sphere")
    keep = np.sort(np.random.default_rng(5).choice(2048, 1600, replace=False)).astype(np.int64)
    filled = synth.fill_parameters(detector_param_shapes("ball", 4))
    st = DetectorStep("ball", opt, DEV)
    st.load_numpy_state(filled)
    dev_batch = batch_to_device(batch_np, DEV)
    dev_batch["keep_idx"] = torch.from_numpy(keep).to(DEV)
    st.step(dev_batch)
    torch.cuda.synchronize()
    from oracle import detector as od
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in filled.items() if not ("running_" in k or "num_batches" in k)}
    bufs = {k: torch.from_numpy(v.copy()) for k, v in filled.items() if "running_" in k}
    ob = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    ob["keep_idx"] = torch.from_numpy(keep)
    ref = od.detector_step(P, bufs, ob, "ball", opt.node_knn_k_1, opt.loss_sigma_lower_bound, opt.keypoint_on_pc_alpha,
                           on_pc_type="point_to_plane")
    for k, v in st.detector.last_indices.items():
        assert np.array_equal(v.cpu().numpy(), ref[k].numpy()), k
    for k in ("keypoints", "sigmas", "loss", "loss_chamfer"):
        assert_close(st.last[k].detach().cpu().numpy(), ref[k].detach().numpy(), name=k)
    # the plane term normalises kp - p, a difference ~100x smaller than the coordinates: keypoints equal to 1e-7 of their
    # scale give unit vectors equal to ~1e-5 (the loss itself meets 1e-5 on equal inputs: the fixture test above)
    for k in ("loss_on_pc_src", "loss_on_pc_dst"):
        assert_close(st.last[k].detach().cpu().numpy(), ref[k].detach().numpy(), rel=2e-4, name=k)
    # the unset switch draws its own indices: a different number of points every call, never more than the cloud has
    st2 = DetectorStep("ball", opt, DEV)
    st2.load_numpy_state(filled)
    st2.step(batch_to_device(batch_np, DEV))
    assert 0.7 * 2048 - 1 <= st2.last_keep <= 2048 and st.last_keep == 1600
    assert torch.isfinite(st2.last["loss"])


def test_som_front_end_matches_reference():
    from usip_amd import ops, som
    from oracle import detector as od
    g = load_golden("som_cases.npz")
    x, node = _t(g["x"]), _t(g["node"])
    mask, mask_row_max, min_idx = som.query_topk(node, x, node.shape[2], k=1)
    assert np.array_equal(min_idx.cpu().numpy(), g["min_idx"])
    assert np.array_equal(mask_row_max.cpu().numpy(), g["mask_row_max"])
    assert np.array_equal(mask.sum(1).cpu().numpy(), g["count"])
    mean, count, dec = ops.som_cluster(x, min_idx.int(), node.shape[2])
    assert np.array_equal(count.cpu().numpy(), g["count"])
    om, _, odec = od.som_cluster(torch.from_numpy(g["x"]), torch.from_numpy(g["min_idx"]).long(),
                                 torch.from_numpy(g["count"]).long())
    assert_close(mean.cpu(), om, name="cluster_mean")
    assert_close(dec.cpu(), odec, name="x_decentered")


def test_bn_momentum_decay_matches_reference():
    """a-13 (models/layers.py:61-71, :112-121): MyConv2d / EquivariantLayer forwards with an epoch, decay_step=2,
    decay=0.6, against the reference's own modules: output, the momentum the module ends up with, and the running
    statistics after each of two training calls -- epochs None, 0 (no decay), 1, 2, 5, 9 and 40 (0.01 clamp)."""
    from usip_amd import layers
    g = load_golden("bn_decay_cases.npz")
    for e in g["epochs"]:
        epoch, tag = (None, "none") if e < 0 else (int(e), str(int(e)))
        conv = _load_filled(layers.MyConv2d(5, 8, kernel_size=(1, 1), stride=1, padding=0, bias=True, activation="relu",
                                            normalization="batch", momentum=0.1, bn_momentum_decay_step=2,
                                            bn_momentum_decay=0.6))
        eq = _load_filled(layers.EquivariantLayer(6, 10, activation="relu", normalization="batch", momentum=0.1,
                                                  bn_momentum_decay_step=2, bn_momentum_decay=0.6))
        conv.train()
        eq.train()
        for call in range(2):
            y2 = conv(_t(g["x2"][call]), epoch)
            y1 = eq(_t(g["x1"][call]), epoch)
            for mod, pre in ((conv, "conv"), (eq, "eq")):
                assert_close(mod.norm.running_mean.cpu(), g["%s_rm_%s_%d" % (pre, tag, call)], name="%s rm %s" % (pre, tag))
                assert_close(mod.norm.running_var.cpu(), g["%s_rv_%s_%d" % (pre, tag, call)], name="%s rv %s" % (pre, tag))
        assert conv.norm.momentum == eq.norm.momentum == float(g["momentum_" + tag])
        assert_close(y2.detach().cpu(), g["conv_y_" + tag], name="conv y")
        assert_close(y1.detach().cpu(), g["eq_y_" + tag], name="eq y")


@pytest.mark.parametrize("model", ["ball", "som"])
def test_graph_replay_follows_bn_momentum_decay(model):
    """A captured step froze the BatchNorm momentum it was captured with (a kernel argument).  With
    bn_momentum_decay_step set, the graph cache is keyed on the momentum every BatchNorm runs with -- not on the
    epoch -- so a new momentum re-captures, an unchanged one replays, the clamp at 0.01 stops the re-captures,
    and the cache stays bounded; running statistics follow the eager step exactly."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8, bn_momentum_decay_step=2, bn_momentum_decay=0.6)
    batch = batch_to_device(synth.make_pair_batch(61, 2, 1024, 32, 4, "sphere"), DEV)
    torch.manual_seed(3)
    eager = DetectorStep(model, opt, DEV)
    graph = DetectorStep(model, opt, DEV, graph=True)
    graph.detector.load_state_dict(eager.detector.state_dict())
    epochs = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 5, 9, 9, 10, 11, 40, 41, 60, 60]
    captured = []
    for e in epochs:
        le, lg = eager.step(batch, epoch=e).detach(), graph.step(batch, epoch=e).detach()
        assert_close(lg.cpu().numpy(), le.cpu().numpy(), rel=1e-6, name="loss at epoch %d" % e)
        assert len(graph._graphs) <= graph.max_graphs
        captured.append(len(graph._graphs))
        for (k, a), (_, b) in zip(graph.detector.named_modules(), eager.detector.named_modules()):
            if hasattr(a, "momentum_original"):
                assert a.momentum == b.momentum, (k, e, a.momentum, b.momentum)
    # epochs 10, 11 share momentum 0.1*0.6^5 ; 40, 41, 60 are all clamped to 0.01: no new capture for them
    keys = list(graph._graphs.keys())
    assert len({k[1] for k in keys}) == len(keys)
    moms = {m for k in keys for m in k[1]}
    assert 0.01 in moms
    sg = graph.detector.state_dict()
    for k, v in eager.detector.state_dict().items():
        if v.dtype.is_floating_point:
            assert_close(sg[k].cpu().numpy(), v.cpu().numpy(), rel=1e-6, name=k)
        else:
            assert torch.equal(sg[k], v), k
    # mlp1 / mlp2 (and conv1-5 of the ball detector) are called without the epoch in the reference
    # (networks.py:147-148, :705-709): their momentum never decays
    assert graph.detector.mlp1.norm.momentum == 0.1
    assert graph.detector.knnlayer_1.layers_before[0].norm.momentum == 0.01


def _run_step(fix, pinned=False):
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    from usip_amd import synth
    g = load_golden(fix)
    model = str(g["cfg_model"])
    cs = g["in/src_sn"].shape[1]
    opt = DetectorOptions(surface_normal_len=cs, node_knn_k_1=int(g["cfg_knn"]),
                          loss_sigma_lower_bound=float(g["cfg_sigma_lb"]),
                          keypoint_on_pc_alpha=float(g["cfg_alpha"]))
    if "cfg_activation" in g:                          # round 6: the non-default layer options
        opt.activation, opt.normalization = str(g["cfg_activation"]), str(g["cfg_normalization"])
    if "cfg_k" in g:
        opt.k = int(g["cfg_k"])
    st = DetectorStep(model, opt, DEV)
    st.allow_pinned_decisions = bool(pinned)           # a step refuses to run with pinned decisions otherwise
    sd = st.detector.state_dict()
    st.load_numpy_state(synth.fill_parameters({k: tuple(v.shape) for k, v in sd.items()}))
    batch = batch_to_device({k[3:]: v for k, v in g.items() if k.startswith("in/")}, DEV)
    st.step(batch)
    torch.cuda.synchronize()
    return g, st


def test_a_training_step_refuses_pinned_decisions():
    """The decision-pinning hooks are test instruments: a step object that was not built for such a test raises
    instead of training with them, and the context manager removes them even when the body fails."""
    from usip_amd import functional as Fh
    with pytest.raises(RuntimeError, match="pinned_decisions is active"):
        with Fh.pinned_decisions(pools=[], relu_fix=[]):
            _run_step("detector_som_micro.npz")
    assert not Fh.pins_active()


@pytest.mark.parametrize("fix", ["detector_som_cfg1.npz", "detector_ball_micro.npz", "detector_som_micro.npz",
                                 "detector_lite_micro.npz", "detector_knn_micro.npz"])
def test_detector_step_matches_reference(fix):
    g, st = _run_step(fix)
    idx = st.detector.last_indices
    if "idx/min_idx" in g:
        assert np.array_equal(idx["min_idx"].cpu().numpy(), g["idx/min_idx"])
        assert np.array_equal(idx["first_idx"].cpu().numpy(), g["idx/index_max_0"])
        assert np.array_equal(idx["second_idx"].cpu().numpy(), g["idx/index_max_1"])
    if "idx/ball_idx" in g:
        assert np.array_equal(idx["ball_idx"].cpu().numpy(), g["idx/ball_idx"])
    if "idx/nn_idx" in g:      # RPN_Detector_KNN: the set torch.topk(sorted=False) picked, nearest first
        assert np.array_equal(idx["nn_idx"].cpu().numpy(), g["idx/nn_idx"])
    assert np.array_equal(idx["knn_I"].cpu().numpy(), g["idx/knn_I"])
    for k in ("node", "keypoints", "sigmas", "loss", "loss_chamfer", "chamfer_pure", "chamfer_weighted",
              "loss_on_pc_src", "loss_on_pc_dst"):
        assert_close(st.last[k].detach().cpu().numpy(), g[k], name=k)
    for k, v in st.detector.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.cpu().numpy(), g["buf/" + k], name=k)
    # gradients: flip-tolerant whole-step bound (see module docstring)
    num = den = 0.0
    biggest = max(float(v) for k, v in g.items() if k.startswith("grad_norm/"))
    for k, p in st.detector.named_parameters():
        if float(g["grad_norm/" + k]) < 1e-5 * biggest:
            continue      # analytically zero gradient (a bias whose effect a later BatchNorm removes):
            #               the reference's value is rounding noise, not a number to match
        gr = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        ref_head = g["grad_head/" + k].astype(np.float64)
        scale = max(np.abs(gr).max(), 1e-30)
        assert np.abs(gr[:48] - ref_head).max() / scale <= 5e-2, k
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["grad_norm/" + k])) <= 2e-2 * float(g["grad_norm/" + k]), k
        num += ((gr[:48] - ref_head) ** 2).sum()
        den += (ref_head ** 2).sum()
    assert np.sqrt(num / den) <= 2e-2


@pytest.mark.parametrize("fix", ["detector_ball_elu_instance.npz", "detector_som_swish.npz", "detector_som_k2.npz"])
def test_detector_step_with_non_default_layer_options_matches_reference(fix, matmul_mode):
    """VERDICT r5 missing #4: --activation elu | swish (| leakyrelu | selu) and --normalization instance
    (models/layers.py:181-193, :262-272) no longer raise: such layers run their convolution (and BatchNorm statistics) on
    the HIP kernels and the rest as device-tensor operations, the expand + cat + max sequences literally
    (usip_amd/layers.py::_generic_forward, pooled_concat_layer); --k 2 (util/som.py:49-50, networks.py:85-92) runs the k = 1
    computation over the twice-stacked cloud with the top-2 assignment (usip_amd/som.py::topk_assign).  Fixtures from the
    reference's own networks.py: indices bit-exact (for k = 2 the per-point SETS of nodes: topk(sorted=False) leaves
    their order unspecified), node / keypoints / sigmas / losses / BatchNorm buffers 1e-5; gradients: every parameter's
    norm and first entries at the free-running bound."""
    g, st = _run_step(fix)
    idx = st.detector.last_indices
    if "cfg_k" in g:
        k = int(g["cfg_k"])
        mine, ref = idx["min_idx"].cpu().numpy(), g["idx/min_idx"]
        B = mine.shape[0]
        assert mine.shape == ref.shape
        assert np.array_equal(np.sort(mine.reshape(B, k, -1), axis=1), np.sort(ref.reshape(B, k, -1), axis=1))
    elif "idx/min_idx" in g:
        assert np.array_equal(idx["min_idx"].cpu().numpy(), g["idx/min_idx"])
        assert np.array_equal(idx["first_idx"].cpu().numpy(), g["idx/index_max_0"])
        assert np.array_equal(idx["second_idx"].cpu().numpy(), g["idx/index_max_1"])
    if "idx/ball_idx" in g:
        assert np.array_equal(idx["ball_idx"].cpu().numpy(), g["idx/ball_idx"])
    assert np.array_equal(idx["knn_I"].cpu().numpy(), g["idx/knn_I"])
    for k in ("node", "keypoints", "sigmas", "loss", "loss_chamfer", "chamfer_pure", "chamfer_weighted",
              "loss_on_pc_src", "loss_on_pc_dst"):
        assert_close(st.last[k].detach().cpu().numpy(), g[k], name=k)
    for k, v in st.detector.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.cpu().numpy(), g["buf/" + k], name=k)
    biggest = max(float(v) for k, v in g.items() if k.startswith("grad_norm/"))
    seen = 0
    for k, p in st.detector.named_parameters():
        assert p.grad is not None, k
        if float(g["grad_norm/" + k]) < 1e-5 * biggest:
            continue
        gr = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        ref_head = g["grad_head/" + k].astype(np.float64)
        scale = max(np.abs(gr).max(), 1e-30)
        assert np.abs(gr[:48] - ref_head).max() / scale <= 5e-2, k
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["grad_norm/" + k])) <= 2e-2 * float(g["grad_norm/" + k]), k
        seen += 1
    assert seen >= 20


@pytest.mark.parametrize("fix", ["detector_som_cfg1.npz", "detector_ball_micro.npz", "detector_som_micro.npz",
                                 "detector_lite_micro.npz", "detector_knn_micro.npz"])
def test_detector_step_gradients_match_reference_with_pinned_decisions(fix, matmul_mode):
    """a-11 at the north star's bar.  The step ends in loss.backward() (keypoint_detector.py:205).  Its gradient is
    a smooth function of the parameters only BETWEEN changes of the forward's discrete decisions: which neighbour
    every max-pool over K picks (networks.py:706,710, layers.py:433,438) and which pre-activations every ReLU lets
    through.  Two correct fp32 forwards (different summation order) take a handful of those decisions differently
    -- values within rounding distance of a tie or of zero -- and with a few hundred positions per channel in the
    head each flip moves gradients by O(1e-2): the reference's own fp32 run sits 6e-2 from its fp64 run on these
    fixtures for exactly that reason (tools/grad_conditioning.py).  So gradients are compared at EQUAL
    decisions, all taken from the REFERENCE (the fixture): the arg-max its torch.max returned for every pool and its
    on/off choice for every pre-activation within 1e-4 of zero (all others are unambiguous, which is asserted).

      hip    the HIP step with those decisions
      ref32  the oracle (PyTorch-CPU fp32, bit-identical to the reference on the pinned platform) with them
      truth  the oracle in float64, replaying every decision of ref32 (indices, pools, masks, arg-mins)

    Every parameter gradient of `hip` must lie, entry by entry, within 1e-5 of ref32 (of the tensor's scale) or
    within the distance two independent fp32 evaluations can have: ref32's own fp32 distance from the truth is one
    evaluation's noise e, `hip` is a second evaluation, so |hip - ref32| <= ~2e.  Bars: per tensor 3x ITS reference
    noise (measured <= 2.9x), and over the whole fixture 2.5x the fixture's worst reference noise (measured <= 2.0x;
    profiles/r03zh_pinned_grad_errors_*.json; round 2 allowed 4x per tensor).  Three of the five fixtures pass at 1e-5
    outright.  The fixture's digests of the reference gradient (first 48 entries, norm, four whole-tensor
    projections) must hold too.  Runs in BOTH fp32-accurate arithmetic modes: fp32 MFMA, and the split-product mode bench.py times
    (forced onto every launch its tile supports, see conftest.matmul_mode)."""
    from oracle import detector as od
    from usip_amd import functional as Fh
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions, detector_param_shapes
    g = load_golden(fix)
    model = str(g["cfg_model"])
    cs = g["in/src_sn"].shape[1]
    opt = DetectorOptions(surface_normal_len=cs, node_knn_k_1=int(g["cfg_knn"]),
                          loss_sigma_lower_bound=float(g["cfg_sigma_lb"]), keypoint_on_pc_alpha=float(g["cfg_alpha"]))
    filled = synth.fill_parameters(detector_param_shapes(model, cs))
    batch_np = {k[3:]: v for k, v in g.items() if k.startswith("in/")}
    n_pools = sum(k.startswith("idx/pool_arg_") for k in g)
    n_relu = sum(k.startswith("idx/relu_near_idx_") for k in g)
    pools = [torch.from_numpy(g["idx/pool_arg_%d" % i].astype(np.int64)) for i in range(n_pools)]
    relu_fix = [(torch.from_numpy(g["idx/relu_near_idx_%d" % i].astype(np.int64)), torch.from_numpy(g["idx/relu_near_on_%d" % i]))
                for i in range(n_relu)]

    def oracle(dtype, tape):
        P = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in filled.items()
             if not ("running_" in k or "num_batches" in k)}
        bufs = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in filled.items() if "running_" in k}
        od.TAPE = tape
        try:
            res = od.detector_step(P, bufs, {k: torch.from_numpy(v).to(dtype) for k, v in batch_np.items()}, model,
                                   opt.node_knn_k_1, opt.loss_sigma_lower_bound, opt.keypoint_on_pc_alpha)
        finally:
            od.TAPE = None
        return P, res

    tape = od.DecisionTape(pools=pools, relu_fix=relu_fix)
    ref32, res32 = oracle(torch.float32, tape)
    assert tape.pools == [] and tape.relu_fix == []
    truth, _ = oracle(torch.float64, od.DecisionTape(replay=tape.rec))

    with Fh.pinned_decisions(pools=[p.int() for p in pools], relu_fix=relu_fix) as pins:
        g, st = _run_step(fix, pinned=True)
    assert pins.leftover == (0, 0)                         # every pool and every layer took its decisions
    assert not Fh.pins_active()
    flips = pins.flips
    for k in ("keypoints", "sigmas", "loss"):
        assert_close(st.last[k].detach().cpu().numpy(), g[k], name=k)
    # the remaining decisions are index tensors: equal to the oracle's (and to the fixture's, asserted elsewhere)
    for k, v in st.detector.last_indices.items():
        assert np.array_equal(v.cpu().numpy(), res32[k].numpy()), k
    biggest = max(float(v) for k, v in g.items() if k.startswith("grad_norm/"))
    report, bad = {}, {}
    for k, p in st.detector.named_parameters():
        gn = float(g["grad_norm/" + k])
        if gn < 1e-5 * biggest:
            continue                                   # analytically zero (conv bias in front of a BatchNorm)
        hip = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        r32 = ref32[k].grad.numpy().ravel().astype(np.float64)
        tru = truth[k].grad.numpy().ravel()
        scale = max(np.abs(tru).max(), 1e-30)
        ref_noise = np.abs(r32 - tru).max() / scale            # the reference's own fp32 distance from the truth
        bar = max(1e-5, 3 * ref_noise)
        e = dict(hip_vs_ref32=np.abs(hip - r32).max() / scale, hip_vs_truth=np.abs(hip - tru).max() / scale,
                 ref32_vs_truth=ref_noise,
                 digests=max(np.abs(hip[:48] - g["grad_head/" + k]).max() / scale,
                             abs(np.sqrt((hip ** 2).sum()) - gn) / gn,
                             np.abs(synth.grad_projections(k, hip) - g["grad_proj/" + k]).max() / scale))
        report[k] = e
        if e["hip_vs_ref32"] > bar or e["digests"] > bar or e["hip_vs_truth"] > bar:
            bad[k] = e
    fixture_noise = max(v["ref32_vs_truth"] for v in report.values())
    fixture_worst = max(max(v["hip_vs_ref32"], v["hip_vs_truth"], v["digests"]) for v in report.values())
    if fixture_worst > max(1e-5, 2.5 * fixture_noise):
        bad["(whole fixture)"] = dict(worst=fixture_worst, reference_noise=fixture_noise)
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):                         # evidence for DESIGN.md: the per-parameter errors
        with open(os.path.join(out_dir, "pinned_grad_errors_%s_%s.json" % (fix.replace(".npz", ""), matmul_mode)), "w") as f:
            json.dump(dict(matmul_mode=matmul_mode, worst={q: max(v[q] for v in report.values()) for q in
                                                          ("hip_vs_ref32", "hip_vs_truth", "ref32_vs_truth", "digests")},
                           errors=report, relu_decisions_nudged_per_layer=flips,
                           relu_decisions_listed=int(sum(i.numel() for i, _ in relu_fix)),
                           relu_decisions_total=int(sum(int(g["idx/relu_numel_%d" % i]) for i in range(n_relu)))), f, indent=1)
    assert not bad, "gradients off with the decisions pinned: %s" % bad
    assert sum(flips) <= 64                                # a handful of decisions, not a different forward


def _oracle_step(model, opt, batch_np, filled, return_params=False, return_both=False):
    from oracle import detector as od
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in filled.items()
         if not ("running_" in k or "num_batches" in k)}
    bufs = {k: torch.from_numpy(v.copy()) for k, v in filled.items() if "running_" in k}
    res = od.detector_step(P, bufs, {k: torch.from_numpy(v) for k, v in batch_np.items()}, model,
                           opt.node_knn_k_1, opt.loss_sigma_lower_bound, opt.keypoint_on_pc_alpha)
    if return_both:
        return P, res
    return P if return_params else res


def test_config2_shape_parity_vs_oracle():
    """BASELINE.json configs[1] shapes (ModelNet40 detector: N=5000, M=64, node_knn_k_1=32, Cs=3; the
    reference default RPN_Detector) at a reduced batch (3 pairs instead of 24 so the CPU oracle finishes in
    seconds), fp32: indices bit-exact, floats 1e-5.  The config's bf16 is a perf mode, not the parity mode."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=3, node_knn_k_1=32, loss_sigma_lower_bound=1e-4, keypoint_on_pc_alpha=1.0)
    batch_np = synth.make_pair_batch(2024, 3, 5000, 64, 3, "sphere")
    st = DetectorStep("som", opt, DEV)
    filled = synth.fill_parameters({k: tuple(v.shape) for k, v in st.detector.state_dict().items()})
    st.load_numpy_state(filled)
    st.step(batch_to_device(batch_np, DEV))
    torch.cuda.synchronize()
    ref = _oracle_step("som", opt, batch_np, filled)
    for k, v in st.detector.last_indices.items():
        assert np.array_equal(v.cpu().numpy(), ref[k].numpy()), k
    for k in ("node", "keypoints", "sigmas", "loss", "loss_chamfer", "chamfer_pure", "chamfer_weighted"):
        assert_close(st.last[k].detach().cpu().numpy(), ref[k].detach().numpy(), name=k)


_CFG3 = {}


def _config3_case():
    """BASELINE.json configs[2] shapes at a batch the CPU oracle handles in seconds: 2 pairs = 4 clouds, N=16384, M=512,
    Kb=64, Kn=16, Cs=4, "slab" clouds -- the same per-cloud sizes bench.py times, and enough clouds that the dispatcher
    picks the kernels it picks there (256-row tiles, the direct f32x2 GEMM, the persistent narrow / fused layer
    kernels).  One free-running oracle step (fwd + losses + bwd), cached for the three arithmetic modes."""
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 1))    # ATen's strided reductions crawl with 256 threads (bench.py)
    if not _CFG3:
        from usip_amd import synth
        from usip_amd.networks import DetectorOptions, detector_param_shapes
        opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
        batch_np = synth.make_pair_batch(4321, 2, 16384, 512, 4, "slab")
        filled = synth.fill_parameters(detector_param_shapes("ball", 4))
        from oracle import detector as od
        P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in filled.items()
             if not ("running_" in k or "num_batches" in k)}
        bufs = {k: torch.from_numpy(v.copy()) for k, v in filled.items() if "running_" in k}      # updated in place
        res = od.detector_step(P, bufs, {k: torch.from_numpy(v) for k, v in batch_np.items()}, "ball",
                               opt.node_knn_k_1, opt.loss_sigma_lower_bound, opt.keypoint_on_pc_alpha)
        _CFG3.update(opt=opt, batch_np=batch_np, filled=filled, P=P, res=res, bufs=bufs)
    return _CFG3


# per-channel bounds of test_config3_shape_parity_vs_oracle (see there)
REL_BY_CHANNEL_KEYPOINTS = 1e-5          # measured on MI355X (profiles/r06_rel_by_channel_*.json): 2.0e-6, the y row; x, z 1.9e-7
REL_BY_CHANNEL_RUNNING_VAR = 1e-5        # measured: 2.2e-7 per element


def test_config3_shape_parity_vs_oracle(matmul_mode_natural, monkeypatch):
    """oracle parity AT THE SIZE THE BENCH TIMES, natural dispatch, all three fp32-accurate modes.
    Every index tensor bit-exact; node / keypoints / sigmas / the three losses / BatchNorm buffers within 1e-5 of the
    oracle (oracle/detector.py = the reference's ATen calls); gradients free-running with the flip-tolerant bound of
    test_detector_step_matches_reference (a handful of max-pool / ReLU decisions within rounding distance of a tie move
    gradient entries by O(1e-2): DESIGN.md 3)."""
    from usip_amd import ops
    from usip_amd.step import DetectorStep, batch_to_device
    c = _config3_case()
    st = DetectorStep("ball", c["opt"], DEV)
    st.load_numpy_state(c["filled"])
    # the library's own (strict) rule for BatchNorm bounds: every coefficient tensor the step hands to a two-plane
    # kernel -- forward, and out of ctx.saved_tensors in backward -- must carry its recorded sample count
    monkeypatch.delenv("USIP_ASSUME_LAUNCH_SAMPLES", raising=False)
    unknown = ops.UNKNOWN_SAMPLE_LOOKUPS
    st.step(batch_to_device(c["batch_np"], DEV))
    torch.cuda.synchronize()
    assert ops.UNKNOWN_SAMPLE_LOOKUPS == unknown
    ref = c["res"]
    for k, v in st.detector.last_indices.items():
        assert np.array_equal(v.cpu().numpy(), ref[k].numpy()), k
    for k in ("node", "keypoints", "sigmas", "loss", "loss_chamfer", "chamfer_pure", "chamfer_weighted"):
        assert_close(st.last[k].detach().cpu().numpy(), ref[k].detach().numpy(), name=k)
    # BatchNorm buffers: the oracle updated its copies in place from the same starting values
    for k, v in st.detector.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.cpu().numpy(), c["bufs"][k].numpy(), name=k)
    # The same comparison PER CHANNEL: every coordinate row of node / keypoints [B,3,M] against its own
    # maximum (x, z ~ +-50, y ~ N(0,1)), every cloud's sigmas, and the BatchNorm buffers per element for the running variances
    # (positive, no cancellation) -- the running MEANS cross zero, there a per-element ratio has no scale and the per-tensor bar
    # above is the meaningful one.  Recorded (gpurun_out/ -> profiles/r06_rel_by_channel_<mode>.json) and held to the bounds
    # measured on MI355X, stated below where they exceed 1e-5.
    from conftest import rel_by_channel
    fig = {}
    for k, ax in (("node", 1), ("keypoints", 1), ("sigmas", 0)):
        fig[k], per = rel_by_channel(st.last[k].detach().cpu().numpy(), ref[k].detach().numpy(), ax)
        fig[k + "_slices"] = [float("%.3e" % v) for v in per[:8]]
    worst_var = 0.0
    for k, v in st.detector.state_dict().items():
        if k.endswith("running_var"):
            worst_var = max(worst_var, rel_by_channel(v.cpu().numpy(), c["bufs"][k].numpy(), 0)[0])
    fig["running_var_per_element"] = worst_var
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "r06_rel_by_channel_%s.json" % matmul_mode_natural), "w") as f:
        json.dump({k: (float("%.3e" % v) if isinstance(v, float) else v) for k, v in fig.items()}, f)
    assert fig["node"] <= 1e-5 and fig["sigmas"] <= 1e-5, fig
    assert fig["keypoints"] <= REL_BY_CHANNEL_KEYPOINTS, fig
    assert fig["running_var_per_element"] <= REL_BY_CHANNEL_RUNNING_VAR, fig
    gmax = max(float(p.grad.abs().max()) for p in c["P"].values() if p.grad is not None)
    num = den = 0.0
    for k, p in st.detector.named_parameters():
        r = c["P"][k].grad.numpy().ravel().astype(np.float64)
        if np.abs(r).max() < 1e-5 * gmax:
            continue                                   # analytically zero (a bias in front of a BatchNorm)
        g = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        assert abs(np.linalg.norm(g) - np.linalg.norm(r)) <= 2e-2 * np.linalg.norm(r), k
        num += ((g - r) ** 2).sum()
        den += (r ** 2).sum()
    assert np.sqrt(num / den) <= 2e-2


def test_config3_shape_gradients_at_recorded_decisions():
    """The gradient half of the same case in the mode bench.py times (f32x2), entry by entry: the HIP step records its
    own discrete decisions (arg-max of the four max-pools, on/off of every pre-activation within 1e-4 of zero), the
    oracle replays them in fp32 and -- replaying every decision of that run -- in fp64.  Bar per tensor: 1e-5 of its
    scale or 3x the oracle's own fp32 distance from its fp64 truth (the bar of
    test_detector_step_gradients_match_reference_with_pinned_decisions, here at N=16384 / M=512)."""
    from oracle import detector as od
    from usip_amd import functional as Fh
    from usip_amd import ops
    from usip_amd.step import DetectorStep, batch_to_device
    c = _config3_case()
    prev = ops.set_matmul_mode("f32x2")
    try:
        st = DetectorStep("ball", c["opt"], DEV)
        st.allow_pinned_decisions = True
        st.load_numpy_state(c["filled"])
        rec = Fh.pinned_decisions(record=True)
        with rec:
            st.step(batch_to_device(c["batch_np"], DEV))
            torch.cuda.synchronize()
    finally:
        ops.set_matmul_mode(prev)
    pools = [p.long().cpu() for p in rec.pools]
    relu_fix = [(i.long().cpu(), o.cpu()) for i, o in rec.relu]

    def oracle(dtype, tape):
        P = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in c["filled"].items()
             if not ("running_" in k or "num_batches" in k)}
        bufs = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in c["filled"].items() if "running_" in k}
        od.TAPE = tape
        try:
            res = od.detector_step(P, bufs, {k: torch.from_numpy(v).to(dtype) for k, v in c["batch_np"].items()}, "ball",
                                   c["opt"].node_knn_k_1, c["opt"].loss_sigma_lower_bound, c["opt"].keypoint_on_pc_alpha)
        finally:
            od.TAPE = None
        return P, res

    tape = od.DecisionTape(pools=pools, relu_fix=relu_fix)
    ref32, res32 = oracle(torch.float32, tape)
    assert tape.pools == [] and tape.relu_fix == []
    truth, _ = oracle(torch.float64, od.DecisionTape(replay=tape.rec))
    for k, v in st.detector.last_indices.items():
        assert np.array_equal(v.cpu().numpy(), res32[k].numpy()), k
    for k in ("keypoints", "sigmas", "loss"):
        assert_close(st.last[k].detach().cpu().numpy(), res32[k].detach().numpy(), name=k)
    gmax = max(float(p.grad.abs().max()) for p in truth.values())
    report, bad = {}, {}
    for k, p in st.detector.named_parameters():
        tru = truth[k].grad.numpy().ravel()
        scale = float(np.abs(tru).max())
        if scale < 1e-5 * gmax:
            continue
        hip = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        r32 = ref32[k].grad.numpy().ravel().astype(np.float64)
        noise = np.abs(r32 - tru).max() / scale
        e = dict(hip_vs_ref32=np.abs(hip - r32).max() / scale, hip_vs_truth=np.abs(hip - tru).max() / scale, ref32_vs_truth=noise)
        report[k] = e
        if max(e["hip_vs_ref32"], e["hip_vs_truth"]) > max(1e-5, 3 * noise):
            bad[k] = e
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "config3_recorded_decision_grad_errors_f32x2.json"), "w") as f:
            json.dump(dict(worst={q: max(v[q] for v in report.values()) for q in ("hip_vs_ref32", "hip_vs_truth", "ref32_vs_truth")},
                           relu_decisions_listed=int(sum(i.numel() for i, _ in relu_fix)), errors=report), f, indent=1)
    assert not bad, "gradients off at the recorded decisions: %s" % bad


def test_full_size_step_properties(matmul_mode_natural):
    """(both arithmetic modes, each with the dispatcher's own kernel choice: f32x3 here IS what bench.py times)
    BASELINE.json configs[2] at FULL size (8 pairs = 16 clouds, N=16384, M=512, K=64, Kn=16): the oracle
    needs minutes there, so the step is checked through size-independent properties:
    ball indices lie inside the radius-2 ball and are the first hits in index order; KNN rows are sorted by
    the exact distance and start with the query itself; the probabilistic-chamfer partner indices attain the
    row minima; the forward is reproducible bit for bit; loss and every gradient are finite and non-trivial."""
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
    torch.manual_seed(0)
    st = DetectorStep("ball", opt, DEV)
    batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), DEV)
    st.step(batch)
    loss1 = st.last["loss"].detach().clone()
    kp1 = st.last["keypoints"].detach().clone()
    x = torch.cat((batch["src_pc"], batch["dst_pc"]), 0)
    node = torch.cat((batch["src_node"], batch["dst_node"]), 0)
    ball = st.detector.last_indices["ball_idx"].long()
    dist = ops.pairwise_dist(node.contiguous(), x.contiguous())
    inside = dist <= 2.0
    n_in = inside.sum(-1)
    assert bool(torch.gather(inside, 2, ball)[n_in > 0].all())
    rank = torch.cumsum(inside.long(), -1) - 1
    j = torch.arange(64, device=DEV).view(1, 1, 64)
    genuine = j < torch.clamp(n_in, max=64).unsqueeze(-1)
    assert torch.equal(torch.gather(rank, 2, ball)[genuine], j.expand_as(ball)[genuine])
    knn = st.detector.last_indices["knn_I"].long()
    nd = ops.pairwise_dist(node.contiguous(), node.contiguous())
    kd = torch.gather(nd, 2, knn)
    assert bool((kd[..., 1:] >= kd[..., :-1]).all())
    assert torch.equal(knn[..., 0], torch.arange(512, device=DEV).expand(16, 512))
    assert torch.equal(kd[..., -1], torch.topk(nd, 16, dim=2, largest=False)[0][..., -1])
    J, I = st.chamfer_criteria.last_indices
    assert torch.isfinite(loss1) and float(loss1) > 0
    g = st.bucket.flat
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    st.detector.load_state_dict(st.detector.state_dict())
    torch.manual_seed(0)
    st2 = DetectorStep("ball", opt, DEV)
    st2.forward_losses(batch)
    assert torch.equal(st2.last["loss"].detach(), loss1) and torch.equal(st2.last["keypoints"].detach(), kp1)


def test_descriptor_step_matches_reference(matmul_mode):
    """SURVEY 8 f-1 (BASELINE configs[4] path): DescriptorLiteOld + DescPairScanLoss step against the fixture
    captured from the reference: ball indices bit-exact, floats 1e-5, gradients flip-tolerant as above."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DescriptorStep, batch_to_device
    g = load_golden("descriptor_micro.npz")
    opt = DetectorOptions(surface_normal_len=4)
    st = DescriptorStep(opt, DEV)
    st.load_numpy_state(synth.fill_parameters({k: tuple(v.shape) for k, v in st.descriptor.state_dict().items()}))
    st.descriptor.fixed_permutation = g["perm"]
    batch = batch_to_device({k: g[k] for k in ("anc_pc", "pos_pc", "anc_sn", "pos_sn", "anc_kp", "pos_kp",
                                               "anc_sigmas", "neg_idx")}, DEV)
    st.step(batch)
    torch.cuda.synchronize()
    for k in ("descriptors", "x_features", "triplet", "active", "loss"):
        assert_close(st.last[k].detach().cpu().numpy(), g[k], name=k)
    from oracle import detector as od
    ref_idx = od.ball_query_op(od.pairwise_norm(torch.from_numpy(np.concatenate([g["anc_kp"], g["pos_kp"]])),
                                                torch.from_numpy(np.concatenate([g["anc_pc"], g["pos_pc"]]))[:, :, g["perm"]]),
                               2.0, 64)
    assert np.array_equal(st.descriptor.last_indices["ball_idx"].cpu().numpy(), ref_idx.numpy())
    for k, v in st.descriptor.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.cpu().numpy(), g["buf/" + k], name=k)
    biggest = max(float(v) for k, v in g.items() if k.startswith("grad_norm/"))
    for k, p in st.descriptor.named_parameters():
        if float(g["grad_norm/" + k]) < 1e-5 * biggest:
            continue
        gr = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        assert abs(np.sqrt((gr ** 2).sum()) - float(g["grad_norm/" + k])) <= 2e-2 * float(g["grad_norm/" + k]), k


def test_full_size_descriptor_step_properties():
    """BASELINE.json configs[4] at its per-GPU size (4 pairs = 8 clouds, N=16384, 256 keypoints, K=64, r=2,
    descriptor_len 128; models/networks.py:333-385, losses.py:200-237).  The PyTorch-CPU oracle needs minutes
    there, so: ball indices bit-exact against the C oracle (oracle/usip_oracle.c on the torch.norm matrix), the
    grouped input exactly the gathered, decentred points, descriptors unit-length, the loss the mean of the
    triplet terms, finite non-trivial gradients, and a bit-for-bit reproducible step."""
    from oracle import native
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DescriptorStep, batch_to_device
    pairs, n, kp = 4, 16384, 256
    opt = DetectorOptions(surface_normal_len=4)
    b0 = synth.make_pair_batch(1234, pairs, n, kp, 4, "slab")
    rng = np.random.default_rng(99)
    batch_np = dict(anc_pc=b0["src_pc"], pos_pc=b0["dst_pc"], anc_sn=b0["src_sn"], pos_sn=b0["dst_sn"],
                    anc_kp=b0["src_node"], pos_kp=b0["dst_node"],
                    anc_sigmas=rng.uniform(0.1, 3.0, (pairs, kp)).astype(np.float32),
                    neg_idx=np.roll(np.arange(pairs), 1).astype(np.int64),
                    perm=rng.permutation(n).astype(np.int64))

    def run():
        torch.manual_seed(0)
        st = DescriptorStep(opt, DEV)
        st.step(batch_to_device(batch_np, DEV))
        torch.cuda.synchronize()
        return st

    st = run()
    perm = batch_np["perm"]
    x = np.concatenate([batch_np["anc_pc"], batch_np["pos_pc"]])[:, :, perm]
    sn = np.concatenate([batch_np["anc_sn"], batch_np["pos_sn"]])[:, :, perm]
    kps = np.concatenate([batch_np["anc_kp"], batch_np["pos_kp"]])
    want_idx = native.ball_query(native.pairwise_dist(np.ascontiguousarray(kps), np.ascontiguousarray(x)), 2.0, 64)
    got_idx = st.descriptor.last_indices["ball_idx"].cpu().numpy()
    assert np.array_equal(got_idx, want_idx)
    hits = (want_idx != want_idx[:, :, :1]).any(-1)
    assert hits.mean() > 0.5                                   # most balls hold more than one distinct point
    feat = st.last["x_features"].detach().cpu().numpy()
    aug = np.concatenate([x, sn], 1)
    want_feat = np.take_along_axis(aug[:, :, None, :], want_idx[:, None, :, :].astype(np.int64), axis=3)
    want_feat[:, :3] -= kps[:, :, :, None]
    assert np.array_equal(feat, want_feat)
    desc = st.last["descriptors"].detach()
    nrm = torch.norm(desc, dim=1)
    assert bool(torch.isfinite(desc).all()) and float((nrm - 1).abs().max()) < 1e-3
    trip = st.last["triplet"].detach()
    assert trip.shape == (pairs, kp) and bool((trip >= 0).all())
    assert_close(st.last["loss"].detach().cpu().numpy(), trip.mean().cpu().numpy(), rel=1e-6, name="loss")
    act = st.last["active"].detach()
    assert bool(((act >= 0) & (act <= 1)).all())
    g = st.bucket.flat
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    st2 = run()
    assert torch.equal(st2.last["loss"].detach(), st.last["loss"].detach())
    assert torch.equal(st2.last["descriptors"].detach(), desc)
    assert torch.equal(st2.bucket.flat, g)


def test_adam_update_matches_reference_optimizer():
    """a-11 ends with optimizer.step() (keypoint_detector.py:42-45, :207: Adam, lr, betas (0.9, 0.999)).
    The step's multi-tensor Adam on gradients that live in the flat all-reduce bucket must move the parameters
    exactly as torch.optim.Adam's plain single-tensor implementation does on the same gradients."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    torch.manual_seed(5)
    st = DetectorStep("ball", opt, DEV, with_optimizer=True)
    batch = batch_to_device(synth.make_pair_batch(7, 2, 1024, 32, 4, "sphere"), DEV)
    names = [n for n, p in st.detector.named_parameters() if p.requires_grad]
    ref = [p.detach().cpu().clone().requires_grad_(True) for p in st.bucket.params]
    ref_opt = torch.optim.Adam(ref, lr=opt.lr, betas=(0.9, 0.999), foreach=False, fused=False)
    for it in range(3):
        st.step(batch)
        for r, p in zip(ref, st.bucket.params):
            assert p.grad.data_ptr() >= st.bucket.flat.data_ptr()           # still a view into the bucket
            r.grad = p.grad.detach().cpu().clone()
        ref_opt.step()
        for n, r, p in zip(names, ref, st.bucket.params):
            assert_close(p.detach().cpu().numpy(), r.detach().numpy(), rel=2e-6, name="%s after step %d" % (n, it + 1))


def test_captured_adam_update_follows_a_changed_learning_rate():
    """every train_detector.py calls ModelDetector.update_learning_rate every lr_decay_step epochs
    (keypoint_detector.py:356-366: param_groups[..]['lr'] = ...).  The update replayed from a HIP graph must follow it:
    lr is halved between replays and the parameters are compared with torch.optim.Adam driven by the same gradients
    and the same schedule; the optimizer's checkpoint carries the decayed rate."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    torch.manual_seed(5)
    st = DetectorStep("ball", opt, DEV, with_optimizer=True, graph=True)
    batch = batch_to_device(synth.make_pair_batch(7, 2, 1024, 32, 4, "sphere"), DEV)
    ref = [p.detach().cpu().clone().requires_grad_(True) for p in st.bucket.params]
    ref_opt = torch.optim.Adam(ref, lr=opt.lr, betas=(0.9, 0.999), foreach=False, fused=False)
    for it in range(7):
        if it in (4, 6):                                   # steps 0, 1 run eagerly, 2 captures: 4 and 6 are replays
            for g in st.optimizer.param_groups:
                g["lr"] = g["lr"] * 0.5
            for g in ref_opt.param_groups:
                g["lr"] = g["lr"] * 0.5
        st.step(batch)
        for r, p in zip(ref, st.bucket.params):
            r.grad = p.grad.detach().cpu().clone()
        ref_opt.step()
    assert st.use_graph and len(st._graphs) == 1           # one capture served every learning rate
    for r, p in zip(ref, st.bucket.params):
        assert_close(p.detach().cpu().numpy(), r.detach().numpy(), rel=2e-6, name="parameters after the lr schedule")
    sd = st.optimizer.state_dict()
    assert sd["param_groups"][0]["lr"] == opt.lr * 0.25 and sd["state"][0]["step"].dim() == 0
    st2 = DetectorStep("ball", opt, DEV, with_optimizer=True)
    st2.optimizer.load_state_dict(sd)
    assert st2.optimizer.param_groups[0]["lr"] == opt.lr * 0.25 and float(st2.optimizer.state[st2.optimizer.param]["step"]) == 7.0


def test_graph_replay_equals_eager_steps(matmul_mode):
    """DetectorStep(graph=True): a step replayed from the captured HIP graphs is the eager step -- same kernels,
    same order.  Without an optimizer the parameters stay put, so every call can be compared tightly (loss,
    keypoints, every gradient, BatchNorm buffers); with Adam the loss trajectory, the gradients and the parameters
    after five updates are compared at 1e-5 and the replayed update's step counter is checked."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    batches = [batch_to_device(synth.make_pair_batch(50 + i, 2, 2048, 64, 4, "sphere"), DEV) for i in range(5)]
    torch.manual_seed(9)
    eager = DetectorStep("ball", opt, DEV)
    graph = DetectorStep("ball", opt, DEV, graph=True)
    graph.detector.load_state_dict(eager.detector.state_dict())
    for b in batches:                                  # calls 1-2 eager, call 3 captures + replays, 4-5 replay
        le, lg = eager.step(b).detach(), graph.step(b).detach()
        assert_close(lg.cpu().numpy(), le.cpu().numpy(), rel=1e-6, name="loss")
        assert_close(graph.last["keypoints"].detach().cpu().numpy(), eager.last["keypoints"].detach().cpu().numpy(),
                     rel=1e-6, name="keypoints")
        ge, gg = eager.bucket.flat, graph.bucket.flat
        assert float((ge - gg).norm() / ge.norm()) < 1e-5
    assert len(graph._graphs) == 1
    sg = graph.detector.state_dict()
    for k, v in eager.detector.state_dict().items():   # BatchNorm running statistics and counters
        if v.dtype.is_floating_point:
            assert_close(sg[k].cpu().numpy(), v.cpu().numpy(), rel=1e-6, name=k)
        else:
            assert torch.equal(sg[k], v), k
    # a batch of another shape captures its own graph (after running eagerly where needed)
    other = batch_to_device(synth.make_pair_batch(77, 1, 1024, 32, 4, "sphere"), DEV)
    assert_close(graph.step(other).detach().cpu().numpy(), eager.step(other).detach().cpu().numpy(), rel=1e-6, name="loss")
    assert len(graph._graphs) == 2
    # with Adam: every reduction of the step has a fixed order (the gather backward included), so a replayed
    # step hands Adam the eager step's gradients; the trajectories may differ only by what the capturable
    # (device-side step counter) form of the fused update rounds differently
    torch.manual_seed(9)
    eager = DetectorStep("ball", opt, DEV, with_optimizer=True)
    graph = DetectorStep("ball", opt, DEV, with_optimizer=True, graph=True)
    graph.detector.load_state_dict(eager.detector.state_dict())
    for b in batches:
        le, lg = float(eager.step(b).detach()), float(graph.step(b).detach())
        assert abs(le - lg) <= 1e-5 * max(abs(le), 0.1), (le, lg)
        assert float((eager.bucket.flat - graph.bucket.flat).norm() / eager.bucket.flat.norm()) < 1e-5
    for (k, a), (_, b) in zip(graph.detector.named_parameters(), eager.detector.named_parameters()):
        assert_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rel=1e-5, name=k)
    steps = {int(s["step"]) for s in graph.optimizer.state.values()}
    assert steps == {len(batches)}


def test_descriptor_graph_replay_equals_eager():
    """DescriptorStep(graph=True): the point permutation is a device input of the captured graph, so replayed
    steps with a caller-given permutation reproduce the eager steps."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DescriptorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4)
    torch.manual_seed(4)
    eager = DescriptorStep(opt, DEV)
    graph = DescriptorStep(opt, DEV, graph=True)
    graph.descriptor.load_state_dict(eager.descriptor.state_dict())
    rng = np.random.default_rng(0)
    for i in range(4):
        b0 = synth.make_pair_batch(30 + i, 2, 2048, 32, 4, "slab")
        batch = batch_to_device(dict(anc_pc=b0["src_pc"], pos_pc=b0["dst_pc"], anc_sn=b0["src_sn"], pos_sn=b0["dst_sn"],
                                     anc_kp=b0["src_node"], pos_kp=b0["dst_node"],
                                     anc_sigmas=rng.uniform(0.1, 3.0, (2, 32)).astype(np.float32),
                                     neg_idx=np.array([1, 0], dtype=np.int64),
                                     perm=rng.permutation(2048).astype(np.int64)), DEV)
        le, lg = eager.step(batch).detach(), graph.step(batch).detach()
        assert_close(lg.cpu().numpy(), le.cpu().numpy(), rel=1e-6, name="loss %d" % i)
        assert_close(graph.last["descriptors"].detach().cpu().numpy(), eager.last["descriptors"].detach().cpu().numpy(),
                     rel=1e-6, name="descriptors %d" % i)
        assert float((eager.bucket.flat - graph.bucket.flat).norm() / eager.bucket.flat.norm()) < 1e-5
    assert len(graph._graphs) == 1



@pytest.mark.parametrize("model", ["ball", "som"])
def test_deferred_weight_gradient_reductions_change_no_bit(model, matmul_mode, monkeypatch):
    """Round 5: inside a step the fixed-order sums of the weight gradients' partial tiles are recorded during backward and
    issued together behind it (usip_wgrad_defer / usip_wgrad_flush: two launches instead of one per layer).  Same summation
    order: losses, the gradient bucket and the parameters after three Adam steps are the SAME BITS as with every sum
    launched behind its producer (USIP_DEFER_WGRAD=0) -- eagerly and from a HIP graph."""
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    batch = batch_to_device(synth.make_pair_batch(22, 2, 2048, 64, 4, "sphere"), DEV)

    def run(defer, graph):
        monkeypatch.setenv("USIP_DEFER_WGRAD", "1" if defer else "0")
        torch.manual_seed(17)
        st = DetectorStep(model, opt, DEV, with_optimizer=True, graph=graph)
        losses = [st.step(batch).detach().clone() for _ in range(4)]
        torch.cuda.synchronize()
        return losses, st.bucket.flat.clone(), st.bucket.flat_param.detach().clone(), getattr(st, "deferred_reductions", 0)

    l0, g0, p0, n0 = run(False, False)
    assert n0 == 0
    for graph in (False, True):
        l1, g1, p1, n1 = run(True, graph)
        assert n1 >= 8                                      # every layer's weight gradient went through the one flush
        assert all(torch.equal(a, b) for a, b in zip(l0, l1))
        assert torch.equal(g0, g1) and torch.equal(p0, p1)
    assert ops._DEFER_KEEP is None                          # the mode never outlives a step


def test_deferred_reductions_with_a_frozen_batchnorm_weight(monkeypatch):
    """a layer whose sibling parameter is frozen has no gradient sink (functional._sink returns None), so
    its dW is a fresh tensor handed back to autograd -- whose AccumulateGrad reads it BEFORE the step's flush.  Under the
    deferred mode that tensor was still unwritten (silent garbage) and the flush then wrote into freed memory.  Now such a
    weight gradient's fixed-order sum is launched at once (ops._reduce_now / usip_wgrad_defer_hold): every gradient of a
    step with one frozen BatchNorm weight in a wide layer and one in a narrow layer is the SAME BITS with and without the
    deferred mode, the frozen layers' convolution weights included (round 6: also a pooled-concat layer of each kind and the
    gathered first layer of the KNN fusion module, whose dW are assembled from two launches)."""
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    batch = batch_to_device(synth.make_pair_batch(24, 2, 2048, 64, 4, "sphere"), DEV)

    def run(defer):
        monkeypatch.setenv("USIP_DEFER_WGRAD", "1" if defer else "0")
        torch.manual_seed(17)
        st = DetectorStep("ball", opt, DEV)
        frozen = [st.detector.knnlayer_1.layers_before[1].norm.weight, st.detector.conv2.norm.weight,
                  st.detector.conv4.norm.weight,                           # a pooled-concat layer (row-bias form)
                  st.detector.knnlayer_1.layers_after[0].norm.weight,      # the same in the KNN fusion module
                  st.detector.knnlayer_1.layers_before[0].norm.weight]     # the gathered first layer (csrc/knn_layer.hip)
        for p in frozen:                                    # after the bucket was built: these layers lose their sink
            p.requires_grad_(False)
            p.grad = None
        st.step(batch)
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in st.detector.named_parameters() if p.grad is not None}
        return grads, getattr(st, "deferred_reductions", 0)

    g0, n0 = run(False)
    g1, n1 = run(True)
    assert n0 == 0 and n1 >= 6
    assert set(g0) == set(g1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    for k in ("knnlayer_1.layers_before.1.conv.weight", "conv2.conv.weight"):
        assert k in g1 and bool(torch.isfinite(g1[k]).all()) and float(g1[k].abs().max()) > 0, k
    assert ops._DEFER_KEEP is None


def test_deferred_mode_is_left_when_backward_raises(monkeypatch):
    """An exception inside backward must not leave the library recording reductions for whoever calls it next."""
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    st = DetectorStep("ball", DetectorOptions(surface_normal_len=4, node_knn_k_1=8), DEV)
    batch = batch_to_device(synth.make_pair_batch(23, 1, 1024, 32, 4, "sphere"), DEV)
    real = ops.wgrad_flush

    def boom(device):
        raise RuntimeError("injected")
    monkeypatch.setattr(ops, "wgrad_flush", boom)
    with pytest.raises(RuntimeError, match="injected"):
        st.step(batch)
    monkeypatch.setattr(ops, "wgrad_flush", real)
    assert ops._DEFER_KEEP is None
    g = torch.Generator().manual_seed(0)
    G, X = torch.randn(2, 64, 512, generator=g).to(DEV), torch.randn(2, 32, 512, generator=g).to(DEV)
    dW = ops.mlp_wgrad(G, X)                                # launched AND reduced at once again
    want = torch.einsum("bmp,bnp->mn", G.double(), X.double())
    assert float((dW.double() - want).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize("model", ["ball", "som", "lite", "knn"])
def test_training_is_reproducible_bit_for_bit(model, matmul_mode):
    """Every reduction on the path has a fixed order (BatchNorm partials, split-K weight gradients, chamfer and
    nearest-neighbour partner sums, per-wave gather tables): two runs from the same seed must agree in every bit
    of the loss, the gradient bucket and the parameters after three Adam steps."""
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    batch = batch_to_device(synth.make_pair_batch(21, 2, 2048, 64, 4, "sphere"), DEV)

    def run():
        torch.manual_seed(17)
        st = DetectorStep(model, opt, DEV, with_optimizer=True)
        losses = [st.step(batch).detach().clone() for _ in range(3)]
        return losses, st.bucket.flat.clone(), [p.detach().clone() for p in st.bucket.params]

    l1, g1, p1 = run()
    l2, g2, p2 = run()
    assert all(torch.equal(a, b) for a, b in zip(l1, l2))
    assert torch.equal(g1, g2)
    assert all(torch.equal(a, b) for a, b in zip(p1, p2))


def test_descriptor_step_gradients_match_reference_with_pinned_decisions(matmul_mode):
    """f-1 at the same bar as the detector step (models/networks.py:333-385, losses.py:200-237): every parameter
    gradient of the DescriptorLiteOld + DescPairScanLoss step, entry by entry, with the forward's discrete
    decisions taken from the REFERENCE (fixture): the arg-max of both max-pools over K and the near-zero ReLU
    decisions of conv1..conv4.  The remaining decisions -- ball indices, the two descriptor-space arg-mins, which
    hinge terms are active -- must come out equal to the oracle's on their own (asserted)."""
    from oracle import detector as od
    from usip_amd import functional as Fh
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DescriptorStep, batch_to_device
    g = load_golden("descriptor_micro.npz")
    opt = DetectorOptions(surface_normal_len=4)
    keys = ("anc_pc", "pos_pc", "anc_sn", "pos_sn", "anc_kp", "pos_kp", "anc_sigmas", "neg_idx")
    n_pools = sum(k.startswith("idx/pool_arg_") for k in g)
    n_relu = sum(k.startswith("idx/relu_near_idx_") for k in g)
    assert (n_pools, n_relu) == (2, 4)
    pools = [torch.from_numpy(g["idx/pool_arg_%d" % i].astype(np.int64)) for i in range(n_pools)]
    relu_fix = [(torch.from_numpy(g["idx/relu_near_idx_%d" % i].astype(np.int64)), torch.from_numpy(g["idx/relu_near_on_%d" % i]))
                for i in range(n_relu)]
    st = DescriptorStep(opt, DEV)
    st.allow_pinned_decisions = True
    filled = synth.fill_parameters({k: tuple(v.shape) for k, v in st.descriptor.state_dict().items()})
    st.load_numpy_state(filled)
    st.descriptor.fixed_permutation = g["perm"]

    def oracle(dtype, tape):
        P = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in filled.items()
             if not ("running_" in k or "num_batches" in k)}
        bufs = {k: torch.from_numpy(v.copy()).to(dtype) for k, v in filled.items() if "running_" in k}
        batch = {k: torch.from_numpy(g[k]) for k in keys}
        batch = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in batch.items()}
        od.TAPE = tape
        try:
            res = od.descriptor_step(P, bufs, batch, torch.from_numpy(g["perm"]))
        finally:
            od.TAPE = None
        return P, res

    tape = od.DecisionTape(pools=pools, relu_fix=relu_fix)
    ref32, res32 = oracle(torch.float32, tape)
    assert tape.pools == [] and tape.relu_fix == []
    truth, _ = oracle(torch.float64, od.DecisionTape(replay=tape.rec))
    with Fh.pinned_decisions(pools=[p.int() for p in pools], relu_fix=relu_fix) as pins:
        st.step(batch_to_device({k: g[k] for k in keys}, DEV))
        torch.cuda.synchronize()
    assert pins.leftover == (0, 0) and sum(pins.flips) <= 64
    for k in ("descriptors", "triplet", "active", "loss"):
        assert_close(st.last[k].detach().cpu().numpy(), g[k], name=k)
    assert np.array_equal(st.descriptor.last_indices["ball_idx"].cpu().numpy(), res32["ball_idx"].numpy())
    j_pos, j_neg = st.triplet_criteria.last_indices
    assert np.array_equal(j_pos.cpu().numpy(), res32["nn_pos"].numpy())
    assert np.array_equal(j_neg.cpu().numpy(), res32["nn_neg"].numpy())
    biggest = max(float(v) for k, v in g.items() if k.startswith("grad_norm/"))
    report, bad = {}, {}
    for k, p in st.descriptor.named_parameters():
        gn = float(g["grad_norm/" + k])
        if gn < 1e-5 * biggest:
            continue
        hip = p.grad.detach().cpu().numpy().ravel().astype(np.float64)
        r32 = ref32[k].grad.numpy().ravel().astype(np.float64)
        tru = truth[k].grad.numpy().ravel()
        scale = max(np.abs(tru).max(), 1e-30)
        ref_noise = np.abs(r32 - tru).max() / scale
        bar = max(1e-5, 2 * ref_noise)
        e = dict(hip_vs_ref32=np.abs(hip - r32).max() / scale, hip_vs_truth=np.abs(hip - tru).max() / scale,
                 ref32_vs_truth=ref_noise,
                 digests=max(np.abs(hip[:48] - g["grad_head/" + k]).max() / scale, abs(np.sqrt((hip ** 2).sum()) - gn) / gn,
                             np.abs(synth.grad_projections(k, hip) - g["grad_proj/" + k]).max() / scale))
        report[k] = e
        if max(e["hip_vs_ref32"], e["digests"], e["hip_vs_truth"]) > bar:
            bad[k] = e
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "pinned_grad_errors_descriptor_micro_%s.json" % matmul_mode), "w") as f:
            json.dump(dict(matmul_mode=matmul_mode, errors=report, relu_decisions_nudged_per_layer=pins.flips), f, indent=1)
    assert not bad, "descriptor gradients off with the decisions pinned: %s" % bad


def test_full_size_f32_and_f32x3_steps_agree():
    """BASELINE.json configs[2] at FULL size (16 clouds, N=16384, M=512, K=64, Kn=16), the arithmetic bench.py
    times (f32x3, dispatcher's own kernel choice) against fp32 MFMA everywhere, same weights and batch:
    every index tensor equal, loss / keypoints / sigmas within 1e-5, BatchNorm buffers within 1e-5, and -- with the
    f32 run's discrete decisions (pool arg-max, near-zero ReLU on/off) handed to the f32x3 run -- every
    parameter gradient entry by entry.  Gradient bars: every tensor within 4e-5 of its scale (measured worst 1.3e-5,
    profiles/r03zh_full_size_mode_agreement.json: the per-kernel fp64-truth bound of either mode is ~1e-6 per product
    and a whole backward chains ~25 of them through BatchNorm's cancelling sums) and the whole gradient bucket
    within 1e-6 in relative norm (measured 1.3e-7)."""
    from usip_amd import functional as Fh
    from usip_amd import ops, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
    batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), DEV)

    def run(mode, pins):
        prev = ops.set_matmul_mode(mode)
        try:
            torch.manual_seed(0)
            st = DetectorStep("ball", opt, DEV)
            st.allow_pinned_decisions = True
            with pins:
                st.step(batch)
                torch.cuda.synchronize()
        finally:
            ops.set_matmul_mode(prev)
        return st

    rec = Fh.pinned_decisions(record=True)
    a = run("f32", rec)
    assert len(rec.pools) == 4 and len(rec.relu) == 12
    pins = Fh.pinned_decisions(pools=rec.pools, relu_fix=rec.relu)
    b = run("f32x3", pins)
    assert pins.leftover == (0, 0)
    for k, v in a.detector.last_indices.items():
        assert torch.equal(v, b.detector.last_indices[k]), k
    for k in ("loss", "keypoints", "sigmas", "loss_chamfer", "chamfer_pure"):
        assert_close(b.last[k].detach().cpu().numpy(), a.last[k].detach().cpu().numpy(), name=k)
    sa, sb = a.detector.state_dict(), b.detector.state_dict()
    for k, v in sa.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(sb[k].cpu().numpy(), v.cpu().numpy(), name=k)
    report = {}
    gmax = float(a.bucket.flat.abs().max())
    for (k, pa), (_, pb) in zip(a.detector.named_parameters(), b.detector.named_parameters()):
        ga, gb = pa.grad.double(), pb.grad.double()
        scale = float(ga.abs().max())
        if scale < 1e-6 * gmax:
            continue                                   # analytically zero (conv bias in front of a BatchNorm)
        report[k] = float((ga - gb).abs().max()) / scale
    rel_norm = float((a.bucket.flat.double() - b.bucket.flat.double()).norm() / a.bucket.flat.double().norm())
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "full_size_mode_agreement.json"), "w") as f:
            json.dump(dict(worst=max(report.values()), bucket_rel_norm=rel_norm, relu_nudged=pins.flips,
                           relu_listed=int(sum(i.numel() for i, _ in rec.relu)), per_parameter=report), f, indent=1)
    assert sum(pins.flips) <= 256
    assert max(report.values()) <= 4e-5, sorted(report.items(), key=lambda kv: -kv[1])[:5]
    assert rel_norm <= 1e-6, rel_norm


def _bench_line_and_record(res, out_dir, world):
    """(compact line bench.py printed LAST, full record it wrote next to it)."""
    import json
    import os
    text = res.stdout.decode()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and text.strip().splitlines()[-1] == lines[0], text[-2000:]
    assert len(lines[0]) <= (8000 if world == 1 else 12000), len(lines[0])
    return json.loads(lines[0]), json.load(open(os.path.join(str(out_dir), "bench_full_n%d.json" % world)))


def test_bench_gpus_2_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher in the command: bench.py starts the two ranks itself (what the
    driver's SCALE run invokes).  gloo + both ranks on cuda:0 because this box has one GPU and RCCL refuses two
    ranks per device; the line must report two ranks, their census, and the all-reduce it timed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, USIP_DIST_BACKEND="gloo", USIP_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               USIP_BENCH_OUT=str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--pairs", "2",
           "--points", "4096", "--nodes", "128", "--no-cpu-baseline", "--no-kernel-timing"]
    res = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line, out = _bench_line_and_record(res, tmp_path, 2)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and [r["rank"] for r in line["ranks"]] == [0, 1]
    assert line["ranks"][0]["checksum"] == line["ranks"][1]["checksum"] and line["distributed"]["replicas_identical"]
    assert line["distributed"]["allreduce_form"].startswith("two graphs")
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert [r["rank"] for r in out["ranks_seen"]] == [0, 1]
    assert len({r["pid"] for r in out["ranks_seen"]}) == 2
    d = out["distributed"]
    assert d["world_size"] == 2 and d["launcher"].startswith("self-spawned")
    assert d["bucket_bytes"] > 4_000_000 and d["allreduce_us"]["calls_timed_per_rank"] == 4
    assert d["allreduce_us"]["p50"] > 0 and len(d["step_ms_per_rank"]) == 2
    assert d["loss_per_rank"][0] != d["loss_per_rank"][1]       # every rank trains on its own pairs ...
    assert d["param_checksum_per_rank"][0] == d["param_checksum_per_rank"][1]   # ... and the replicas stay identical
    # and a launcher that starts another number of ranks than --gpus says is refused, not mis-reported
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert bad.returncode != 0 and b"WORLD_SIZE" in bad.stderr


def test_bench_two_ranks_graph_replay_on_one_gpu(tmp_path):
    """The N > 1 launch path of bench.py end to end: two ranks (gloo, both on cuda:0 -- RCCL refuses two ranks on
    one device, so this is the only way to exercise it on a 1-GPU box), HIP-graph capture next to a live process
    group, graph A / eager all-reduce of the flat gradient bucket / graph B every step, max-over-ranks timing,
    one JSON line from rank 0.  Both ranks must report the same loss trajectory (identical replicas + identical
    reduced gradients), and the line must say the step was replayed from graphs."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, USIP_DIST_BACKEND="gloo", USIP_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               USIP_BENCH_OUT=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "2", "--pairs", "2", "--points", "4096", "--nodes", "128", "--no-cpu-baseline",
           "--no-kernel-timing"]
    res = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line, out = _bench_line_and_record(res, tmp_path, 2)
    assert line["distributed"]["n1_probe"]["ratio"] > 0 and line["distributed"]["allreduce_in_graph"] is False
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["launch"].startswith("HIP graph replay"), out["config"]["launch"]
    assert "allreduce" in out["config"]["step"]
    assert out["value"] > 0 and np.isfinite(out["loss"])
    assert "capture failed" not in res.stderr.decode()
    d = out["distributed"]
    assert d["allreduce_in_graph"] is False and "2 graphs" in out["config"]["launch"]      # gloo cannot be captured
    assert d["n1_probe"]["n1_reference_ms"] > 0 and d["replicas_identical"]
    # (RCCL's collectives can be captured, and then the whole step is ONE graph: usip_amd/step.py.  That form needs one
    # device per rank and cannot run on this 1-GPU box; a gloo all-reduce inside a capture aborts the process, so the
    # attempt is made with backend "nccl" only, and a refused capture falls back to this two-graph form.)


@pytest.mark.parametrize("model,first", [("ball", "conv1.conv.weight"), ("som", "first_pointnet.layers.0.conv.weight")])
def test_first_layer_weight_gradient_from_the_next_layers_fused_backward(model, first, monkeypatch):
    """Round 6: the detector's first layer (7 -> 64, input without a gradient) no longer makes a pass over its own (dZ, Y):
    dW = dY . S^T with dY = a1 dYhat + q1 (y - mu) + (q1 mu + q0) is linear in three sums which the fused backward of the
    SECOND layer takes while it holds (dYhat, y) of the first in LDS (csrc/layer_bwd_x2.hip, WS; ops.wsum_finalize).  Same
    forward, so every other gradient is the same bits with and without it, and the first layer's weight gradient agrees to
    fp32 summation order."""
    from usip_amd import ops, prof, synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    prev_mode = ops.set_matmul_mode("f32x2")
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    batch = batch_to_device(synth.make_pair_batch(31, 2, 2048, 64, 4, "sphere"), DEV)

    def run(on):
        monkeypatch.setattr(ops, "WSUM", on)
        torch.manual_seed(11)
        st = DetectorStep(model, opt, DEV)
        prof.reset()
        prof.enable(True)
        try:
            st.step(batch)
            names = set(prof.summary())
        finally:
            prof.enable(False)
            prof.reset()
        return {k: p.grad.detach().clone() for k, p in st.detector.named_parameters()}, names

    try:
        g1, n1 = run(True)
        g0, n0 = run(False)
    finally:
        ops.set_matmul_mode(prev_mode)
    assert "wsum_finalize" in n1 and "wsum_finalize" not in n0
    assert "shared_mlp_wgrad 64x7" in n0 and "shared_mlp_wgrad 64x7" not in n1
    for k in g0:
        if k == first:
            assert_close(g1[k].cpu().numpy(), g0[k].cpu().numpy(), rel=2e-6, name=k)
        else:
            assert torch.equal(g1[k], g0[k]), k
