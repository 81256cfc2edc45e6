"""Differentiable building blocks of the detector path, on top of the HIP operators (ops.py).

Each function names the reference lines it replaces.  Index-producing steps are not
differentiable in the reference either (it detaches them and routes gradients through
torch.gather); the autograd.Functions here reproduce exactly those gradients.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops, prof


# Gradient sink (opt-in, used by usip_amd.step.DetectorStep): when enabled, the shared-MLP backward writes
# parameter gradients STRAIGHT into the parameters' .grad storage (views of the flat all-reduce bucket)
# instead of returning them to autograd, which would launch one accumulate kernel per parameter.  Valid
# when every parameter is used once per step and .grad was zeroed before the backward -- DetectorStep does both.
GRAD_SINK = False
# When set (DetectorStep), the per-layer `num_batches_tracked += 1` launches are skipped and the step bumps all
# counters with ONE multi-tensor add instead (12 tiny launches fewer per step; same values).
DEFER_BN_COUNTERS = False


# Narrow layers (64 inputs, 64 / 128 outputs) run their backward as ONE fused kernel (csrc/narrow_bwd.hip) instead of a
# data-gradient and a weight-gradient GEMM that each re-read (dZ, Y).  Same arithmetic class (exact fp32 MFMA).
FUSED_NARROW_BWD = True
# ... and take the BatchNorm-backward sums of the layer that produced their input on the way out (PRE_BN_SUMS below)
FUSED_NARROW_RED = True
import os as _os                                             # noqa: E402
if _os.environ.get("USIP_NARROW_BWD") is not None:           # A/B measurement switches (tools/)
    FUSED_NARROW_BWD = _os.environ["USIP_NARROW_BWD"] not in ("0", "off")
if _os.environ.get("USIP_NARROW_RED") is not None:
    FUSED_NARROW_RED = _os.environ["USIP_NARROW_RED"] not in ("0", "off")


# Scatter-add backwards (torch.gather's) as sums over destination-sorted segments (csrc/segment.hip) instead of LDS
# float atomics; USIP_SEGMENT_BWD=0 restores the round-1 kernels for A/B runs.
SEGMENT_BACKWARD = _os.environ.get("USIP_SEGMENT_BWD", "1") not in ("0", "off")


# Set by the training step for the duration of forward + backward: {weight.data_ptr(): K-major copy [Cin, Cout]}
# made for ALL layers by one launch (ops.multi_transpose) after the last parameter update.  None: every layer
# transposes its own weight (any caller outside the step).
WT_CACHE = None


# Test hook (tests/test_modules_gpu.py, whole-step gradient parity): when set to a list of int32 device tensors
# [B,C,M], the max-pools over K of the step take their ROUTING from it, in call order, instead of from their own
# arg-max: pooled = activation at the given position, gradient to the given position.  A rounding-level
# difference between two correct forwards can flip a near-tie of a max-pool and move a handful of gradient
# entries by O(1e-2); with the routing pinned to the reference's, gradients can be compared at 1e-5.
PIN_POOL_ARGS = None


class pinned_decisions:
    """THE one switch of the two test hooks (PIN_POOL_ARGS here, PIN_RELU_FIX below): a context manager that
    installs the given decisions and ALWAYS removes them again, so that a failing test cannot leave them behind.

        with Fh.pinned_decisions(pools=[...], relu_fix=[...]) as pins:
            step.step(batch)            # a step object built with allow_pinned_decisions=True
        pins.flips                      # nudged ReLU decisions per layer; pins.leftover: decisions nobody consumed

    Outside this context the hooks are inert; usip_amd.step refuses to run (and bench.py asserts it never does)
    with pins active unless the step object was created for such a test."""

    def __init__(self, pools=None, relu_fix=None, record=False):
        """record=True: take no decisions from outside; instead keep this forward's own (`.pools`: the arg-max of
        every pool, `.relu`: (flat index, on) of every pre-activation within PIN_RECORD_NEAR of zero per BatchNorm +
        ReLU layer), in the form another run accepts as pools= / relu_fix=."""
        self._pools = None if pools is None else list(pools)
        self._relu = None if relu_fix is None else list(relu_fix)
        self._record = bool(record)
        self.flips, self.leftover, self.pools, self.relu = [], None, [], []

    def __enter__(self):
        global PIN_POOL_ARGS, PIN_RELU_FIX, PIN_RECORD
        if pins_active():
            raise RuntimeError("usip_amd: pinned_decisions does not nest")
        if self._record:
            PIN_RECORD = self
        else:
            PIN_POOL_ARGS, PIN_RELU_FIX = self._pools, self._relu
        PIN_RELU_FLIPS.clear()
        return self

    def __exit__(self, *exc):
        global PIN_POOL_ARGS, PIN_RELU_FIX, PIN_RECORD
        self.leftover = (len(PIN_POOL_ARGS or []), len(PIN_RELU_FIX or []))
        self.flips = list(PIN_RELU_FLIPS)
        PIN_POOL_ARGS = PIN_RELU_FIX = PIN_RECORD = None
        PIN_RELU_FLIPS.clear()
        return False


PIN_RECORD = None            # the pinned_decisions(record=True) context that is open, if any
PIN_RECORD_NEAR = 1e-4       # as tests/golden/make_golden.py RELU_NEAR


def pins_active() -> bool:
    """True while a pinned_decisions context is open (test-only state; the product never sets it)."""
    return PIN_POOL_ARGS is not None or PIN_RELU_FIX is not None or PIN_RECORD is not None


def _pooled_act(y4, coef, relu, want_yarg=False):
    """(max over K of the lazily activated y4 [B,C,M,K], arg-max i32 [B,C,M][, y4 at the arg-max]); coef None: y4 is
    already activated."""
    yarg = None
    if coef is not None and want_yarg:
        pooled, arg, yarg = ops.group_max_act(y4, coef, relu, want_yarg=True)
    else:
        pooled, arg = ops.group_max_act(y4, coef, relu) if coef is not None else ops.group_max(y4)
    if PIN_RECORD is not None:
        PIN_RECORD.pools.append(arg.clone())
    if PIN_POOL_ARGS is not None:
        arg = PIN_POOL_ARGS.pop(0).to(device=y4.device, dtype=torch.int32).contiguous()
        if tuple(arg.shape) != tuple(y4.shape[:3]):
            raise RuntimeError("PIN_POOL_ARGS: routing %s does not fit a pool over %s" % (tuple(arg.shape), tuple(y4.shape)))
        B, C, M, K = y4.shape
        act = ops.bn_apply(y4.view(B, C, M * K), coef, relu).view(B, C, M, K) if coef is not None else y4
        pooled = act.gather(3, arg.long().unsqueeze(3)).squeeze(3).contiguous()
        if want_yarg:
            yarg = y4.gather(3, arg.long().unsqueeze(3)).squeeze(3).contiguous()
    return (pooled, arg, yarg) if want_yarg else (pooled, arg)


# Second test hook, same purpose: ReLU decisions.  A pre-activation within rounding distance of zero is "on" in one
# correct forward and "off" in another; with a few hundred positions per channel in the head (M nodes per cloud)
# one such flip moves that channel's gradient -- and everything upstream of it -- by O(1e-2).  When set to a list
# of (flat index int64 [n], on bool [n]) pairs, one per BatchNorm+ReLU layer in call order (the decisions of the
# REFERENCE for every pre-activation within 1e-4 of zero, from a fixture), every layer nudges the few pre-BN values
# whose decision differs from the given one across zero (by ~4e-7 of the activation's scale), so that ALL
# consumers of that y (the next GEMM's prologue, pooling, every backward kernel) take the given decision; every
# pre-activation NOT in the list must be farther than `PIN_RELU_MARGIN` from zero, i.e. unambiguous.
PIN_RELU_FIX = None
PIN_RELU_MARGIN = 5e-5
PIN_RELU_FLIPS = []          # number of nudged elements per layer, for the test's report


def _relu_hook(y, coef):
    """Called by every BatchNorm + ReLU layer with its pre-BN output: the identity unless a pinned_decisions context
    is open."""
    if PIN_RECORD is not None:
        z = ops.bn_apply(y, coef, False).reshape(-1)
        idx = torch.nonzero(z.abs() < PIN_RECORD_NEAR).reshape(-1)
        PIN_RECORD.relu.append((idx, z[idx] > 0))
        return y
    if PIN_RELU_FIX is not None:
        return _align_relu_decisions(y, coef)
    return y


def _align_relu_decisions(y, coef):
    """y [nb,C,P] pre-BN output, coef [>=2,C] -> y whose relu decisions at the listed elements are the given ones."""
    idx, on = PIN_RELU_FIX.pop(0)
    idx, on = idx.to(y.device), on.to(y.device)
    z = ops.bn_apply(y, coef, False)                     # the kernels' own arithmetic: fma(y, coef0, coef1)
    zf = z.reshape(-1)
    listed = torch.zeros(zf.numel(), dtype=torch.bool, device=y.device)
    listed[idx] = True
    if idx.numel() < zf.numel() and float(zf[~listed].abs().min()) < PIN_RELU_MARGIN:
        raise RuntimeError("PIN_RELU_FIX: an unlisted pre-activation is within %.0e of zero" % PIN_RELU_MARGIN)
    flip = (zf[idx] > 0) != on
    n = int(flip.sum())
    PIN_RELU_FLIPS.append(n)
    if n:
        C, P = y.shape[1], y.shape[2]
        fi, fon = idx[flip], on[flip]
        ch = (fi // P) % C
        sc, sh = coef[0][ch], coef[1][ch]
        t = torch.where(fon, 1.0, -1.0) * 4e-7 * torch.clamp(sh.abs(), min=1.0)
        y = y.clone()
        y.view(-1)[fi] = (t - sh) / sc
        if not bool(((ops.bn_apply(y, coef, False).reshape(-1)[idx] > 0) == on).all()):
            raise RuntimeError("PIN_RELU_FIX: could not align the decisions of a layer")
    return y


def _kmajor(w2):
    """[Cin, Cout] copy of the weight matrix w2 [Cout, Cin]."""
    cached = WT_CACHE.get(w2.data_ptr()) if WT_CACHE is not None else None
    return cached if cached is not None else w2.detach().t().contiguous()


def _sink(*params):
    """The .grad storages the backward kernels of one layer will OVERWRITE (not accumulate into), or None when
    the sink is off.  Overwriting is only right when a parameter is used once per step: a second use in the same
    step (weight sharing, gradient accumulation over micro-batches) would silently drop the first contribution,
    so it raises instead.  FlatGradBucket.zero() starts a new step."""
    if not GRAD_SINK:
        return None
    out = []
    for p in params:
        g = getattr(p, "grad", None) if p is not None else None
        if p is not None and (g is None or not g.is_contiguous()):
            return None
        out.append(g)
    for p in params:
        if p is not None:
            if getattr(p, "_usip_sink_used", False):
                raise RuntimeError("usip_amd: a parameter is used twice in one step while the gradient sink is on "
                                   "(its backward overwrites .grad); run this step without DetectorStep's sink")
            p._usip_sink_used = True
    return tuple(out)


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError("usip_amd: %s needs device tensors; the HIP path has no CPU fallback" % what)


# --------------------------------------------------------------------------- distances
class _NearestDistance(torch.autograd.Function):
    """d[b,i] = min_j |a_i - b_j|, J[b,i] = first arg-min; gradients as autograd gives them through
    torch.norm + torch.min in the reference (models/losses.py:62-66, :135-143): the unit vector
    to the selected partner for a, its negative scattered onto b, zero at zero distance."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        d, arg32 = ops.nearest(a, b) if a.shape[1] == 3 else ops.nearest_nd(a, b)
        ctx.save_for_backward(a, b, d, arg32)
        ctx.mark_non_differentiable(arg32)
        ctx.set_materialize_grads(False)
        return d, arg32

    @staticmethod
    def backward(ctx, gd, _garg):
        if gd is None:
            return None, None
        a, b, d, arg32 = ctx.saved_tensors
        ga, gb = ops.nearest_backward(a, b, d, arg32, gd.contiguous(), ctx.needs_input_grad[1])
        return (ga if ctx.needs_input_grad[0] else None), gb


def nearest_distance_i32(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """As nearest_distance, arg-min as the kernels produce it (int32)."""
    require_device(a, "nearest_distance")
    return _NearestDistance.apply(a, b)


class _ChamferProb(torch.autograd.Function):
    """The sigma arithmetic of the probabilistic chamfer loss (models/losses.py:82-99) on the minima of the two
    nearest-neighbour reductions: one forward and one backward launch instead of ~110 element-wise ones."""

    @staticmethod
    def forward(ctx, a, J, c, I, sigma_src, sigma_dst):
        sigma_src, sigma_dst = sigma_src.contiguous(), sigma_dst.contiguous()
        out = ops.chamfer_prob(a, J, c, I, sigma_src, sigma_dst)
        ctx.save_for_backward(a, J, c, I, sigma_src, sigma_dst)
        ctx.set_materialize_grads(False)
        loss, pure, weighted = out[0], out[1], out[2]
        ctx.mark_non_differentiable(pure, weighted)
        return loss, pure, weighted

    @staticmethod
    def backward(ctx, gloss, _gp, _gw):
        if gloss is None:
            return (None,) * 6
        a, J, c, I, ss, sd = ctx.saved_tensors
        da, dc, dss, dsd = ops.chamfer_prob_backward(gloss.contiguous(), a, J, c, I, ss, sd)
        return da, None, dc, None, dss, dsd


def chamfer_prob(a, J32, c, I32, sigma_src, sigma_dst):
    """-> (loss, chamfer_pure, chamfer_weighted), 0-dim tensors; the last two carry no gradient."""
    return _ChamferProb.apply(a, J32, c, I32, sigma_src, sigma_dst)


class _DetectorHead(torch.autograd.Function):
    """keypoints = ks[:, :3] + centre, sigmas = softplus(ks[:, 3]) + lower bound (models/networks.py:150-154) as one
    launch each way instead of split / add / softplus / add and their five backward launches."""

    @staticmethod
    def forward(ctx, ks, centre, lower):
        ks = ks.contiguous()
        kp, sg = ops.detector_head(ks, centre.contiguous(), lower)
        ctx.save_for_backward(ks)
        ctx.set_materialize_grads(False)
        return kp, sg

    @staticmethod
    def backward(ctx, g_kp, g_sg):
        (ks,) = ctx.saved_tensors
        if g_kp is None and g_sg is None:
            return None, None, None
        return ops.detector_head_backward(None if g_kp is None else g_kp.contiguous(),
                                          None if g_sg is None else g_sg.contiguous(), ks), None, None


def detector_head(ks, centre, sigma_lower_bound: float):
    """ks [B,4,M] -> (keypoints [B,3,M], sigmas [B,M]); centre carries no gradient (it does not in the reference:
    nodes / cluster means are inputs)."""
    require_device(ks, "detector_head")
    return _DetectorHead.apply(ks, centre.detach(), float(sigma_lower_bound))


class _RigidTransform(torch.autograd.Function):
    """(R * scale) . x + shift per cloud (models/keypoint_detector.py:182-184); R, scale, shift are inputs."""

    @staticmethod
    def forward(ctx, x, R, scale, shift):
        R, scale = R.contiguous(), scale.contiguous()
        ctx.save_for_backward(R, scale)
        return ops.rigid_transform(x.contiguous(), R, scale, shift.contiguous())

    @staticmethod
    def backward(ctx, g):
        R, scale = ctx.saved_tensors
        return ops.rigid_transform(g.contiguous(), R, scale, None, transpose=True), None, None, None


def rigid_transform(x, R, scale, shift):
    require_device(x, "rigid_transform")
    return _RigidTransform.apply(x, R, scale.reshape(-1), shift.reshape(shift.shape[0], 3))


class _DetectorLossCombine(torch.autograd.Function):
    """loss = loss_chamfer + alpha * (mean(d_src) + mean(d_dst)) (models/keypoint_detector.py:196-204), d [2B,M] the
    keypoint-to-cloud distances with the src rows first.  -> (loss, alpha*mean(d_src), alpha*mean(d_dst)); the last
    two are for logging."""

    @staticmethod
    def forward(ctx, d, chamfer, alpha):
        d = d.contiguous()
        out = ops.detector_loss_combine(d, chamfer.reshape(1), alpha)
        ctx.shape, ctx.alpha = tuple(d.shape), float(alpha)
        loss, on_src, on_dst = out[0], out[1], out[2]
        ctx.mark_non_differentiable(on_src, on_dst)
        ctx.set_materialize_grads(False)                      # no zero-filled gradients for the two logging outputs
        return loss, on_src, on_dst

    @staticmethod
    def backward(ctx, gloss, _g1, _g2):
        if gloss is None:
            return None, None, None
        gloss = gloss.contiguous().reshape(1)
        n = 1
        for v in ctx.shape:
            n *= v
        return ops.fill_scaled(gloss, ctx.alpha / (n // 2), ctx.shape), gloss.reshape(()), None


def detector_loss_combine(d, chamfer, alpha: float):
    return _DetectorLossCombine.apply(d, chamfer, float(alpha))


def nearest_distance(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """a [B,C,Ma], b [B,C,Nb] -> (min distance [B,Ma], arg-min int64 [B,Ma] as torch.min returns it).  C == 3:
    coordinates (exact oracle arithmetic); any other C: descriptors (Nb <= 1024)."""
    d, arg32 = nearest_distance_i32(a, b)
    return d, arg32.long()


def knn_indices(query: torch.Tensor, database: torch.Tensor, K: int) -> torch.Tensor:
    """The K nearest database points of every query point, nearest first, int32 [B,M,K]
    (torch.norm + topk(sorted=True), models/layers.py:417-421).  Distances use the oracle platform's
    arithmetic order, so the ordering is the oracle's."""
    require_device(query, "knn_indices")
    q, d = query.detach().contiguous(), database.detach().contiguous()
    if d.shape[2] <= 1024:
        return ops.knn(q, d, K)
    # node counts above 1024 are outside the detector's configurations: distance matrix + ATen top-k
    return torch.topk(ops.pairwise_dist(q, d), k=K, dim=2, largest=False, sorted=True)[1].int()


# --------------------------------------------------------------------------- gathers
def gather_neighbours(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[b,c,m,k] = x[b,c,idx[b,m,k]]  (models/operations.py:271-287, layers.py:422-426,
    networks.py:699-700); backward = scatter-add, as torch.gather's."""
    B, C, _ = x.shape
    _, M, K = idx.shape
    flat = idx.reshape(B, 1, M * K).expand(B, C, M * K)
    return torch.gather(x, 2, flat).view(B, C, M, K)


# --------------------------------------------------------------------------- shared MLP layer
class LazyAct:
    """Output of a shared-MLP layer kept as (pre-BatchNorm GEMM output y, BN coefficients coef[2,C]):
    the activation relu(y*coef[0]+coef[1]) is formed by whoever consumes it -- the next layer's GEMM and
    weight-gradient prologues, or the fused BN+ReLU+max pooling -- so the activated tensor is never written
    to or read back from HBM.  `y` is the autograd-tracked tensor and STANDS FOR the activated output:
    gradients flowing into it are gradients w.r.t. the activated output."""

    def __init__(self, y: torch.Tensor, coef: torch.Tensor, relu: bool, shape, wsrc=None):
        self.y, self.coef, self.relu, self.shape = y, coef, relu, tuple(shape)
        # the gradient-free INPUT [nb, <= 8, P] of the layer that produced y (the detector's first layer), when the
        # consumer's fused backward may take that layer's weight-gradient sums on its way (ops.wsum_supported)
        self.wsrc = wsrc

    def materialize(self) -> torch.Tensor:
        return _Materialize.apply(self.y, self.coef, self.relu).view(self.shape)


class _Materialize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, coef, relu):
        return ops.bn_apply(y, coef, relu)

    @staticmethod
    def backward(ctx, dz):
        return dz, None, None            # y stands for the activated output already


def as_tensor(x):
    return x.materialize() if isinstance(x, LazyAct) else x


# BatchNorm-backward partial sums that the kernel PRODUCING a gradient tensor already took (the fused narrow backward
# has the tile of dX in registers and the tile of the producing layer's pre-BN output in LDS):
#   {dz.data_ptr(): (shape of dz, [partials [2, rows, C], ...], dz._version when the sums were taken)}
# The layer that receives that tensor as its incoming gradient then skips its own pass over (dZ, Y).
# Entries are only registered inside the training step (GRAD_SINK: one consumer per tensor, no hooks, no retained
# graph), are ALWAYS removed by the receiver whether it can use them or not, carry the tensor's version counter so
# that a gradient autograd accumulated into in place (a second consumer) is recognised as changed, and the step
# clears the table when it starts and when it ends -- a stale entry can never meet a recycled address.
PRE_BN_SUMS = {}
# the fused backward of a max-pooled layer also takes the sums of the (pooled-concat) layer in front of it
POOLED_LAYER_RED = _os.environ.get("USIP_POOLED_LAYER_RED", "1") != "0"


def _register_pre_bn_sums(dx, partials):
    if GRAD_SINK:
        PRE_BN_SUMS[dx.data_ptr()] = (tuple(dx.shape), partials, dx._version)


def _take_pre_bn_sums(dz, shape=None):
    """The partial sums registered for exactly this tensor (address, shape, unmodified since), else None; the entry
    is removed either way."""
    pre = PRE_BN_SUMS.pop(dz.data_ptr(), None)
    if pre is None or pre[0] != tuple(dz.shape if shape is None else shape) or pre[2] != dz._version:
        return None
    return pre


def _own_bn_backward(dz, y, coef, mean, invstd, gamma, relu, sink, group=0, took=None):
    """(dgamma, dbeta, coef4[, gsum]) of this layer: from partial sums its gradient's producer already took
    (PRE_BN_SUMS), else by the stand-alone reduction pass over (dZ, y).
    (Folding these sums into the consuming layer's data-gradient GEMM epilogue for the MFMA-bound layers was
    implemented and measured slower -- their epilogue is not overlapped -- and removed)"""
    go, bo = (sink[2], sink[3]) if sink else (None, None)
    pre = _take_pre_bn_sums(dz)                  # removed even when this layer cannot use them
    if pre is not None and group == 0 and relu:
        if took is not None:
            took.extend(pre[1])                  # (the caller looks for RedSums.wsum)
        return ops.bn_backward_from_partials(pre[1], dz.shape[0] * dz.shape[2], coef, mean, invstd, go, bo)
    if pre is not None and group and relu and len(pre[1]) == 1:
        # a pooled-concat layer: the producer (the fused backward of a max-pooled layer) also took the
        # per-neighbourhood sums
        gsum = getattr(pre[1][0], "gsum", None)
        if gsum is not None and gsum.shape[3] * group == dz.shape[2]:
            return ops.bn_backward_from_partials(pre[1], dz.shape[0] * dz.shape[2], coef, mean, invstd, go, bo) + (gsum,)
    dgamma, dbeta, coef4, gsum = ops.bn_backward_reduce(dz, y, coef, mean, invstd, gamma, relu, group=group,
                                                        dgamma_out=go, dbeta_out=bo)
    return (dgamma, dbeta, coef4, gsum) if group else (dgamma, dbeta, coef4)


# The direct f32x2 data-gradient GEMM (csrc/gemm_x2d.hip) can leave the BatchNorm-backward sums of the layer that produced
# its output's activation (round 4, VERDICT r3 item 3: it holds the dX tile in a layout it can read that layer's pre-BN
# tile in; tests/test_f32x2_mode_gpu.py::test_direct_gemm_backward_leaves_the_producing_layers_bn_sums).  MEASURED SLOWER
# and therefore OFF by default: 4.88 -> 5.30 ms per step, same box, twice (profiles/r04ae_gemm_red_ab.txt).  The
# stand-alone pass it replaces reads (dX, Y) at 11.7 TB/s -- dX was just written and sits in the 256 MB Infinity Cache
# -- i.e. 46 us per layer, while the epilogue's reads of the Y tile are latency-bound (one 8-row slab in flight per
# lane; more would spill: the kernel runs at the 128-VGPR limit).  USIP_GEMM_RED=1 enables it.
GEMM_RED = _os.environ.get("USIP_GEMM_RED", "0") not in ("0", "off")


def _dgrad(x, w2c, dz, pro, X2=None, coef=None, pool=None, M=None, a_offset=0, xcoef=None, red_group=0):
    """Data-gradient GEMM dX = W^T . dY with dY rebuilt from (dZ, Y) in the prologue.  xcoef: the input x is the pre-BN
    output of a training-mode BatchNorm + ReLU layer with these coefficients -- when the launch can, its epilogue takes
    that layer's BatchNorm-backward sums and they are registered for it (PRE_BN_SUMS)."""
    if (GEMM_RED and FUSED_NARROW_RED and GRAD_SINK and pro >= 2 and xcoef is not None and xcoef.shape[0] >= 4
            and (M is None or M == x.shape[1]) and a_offset >= 0):
        dx, _, r = ops.mlp_gemm(w2c, dz, pro=pro, X2=X2, coef=coef, tag="dgrad", pool=pool, M=M, a_offset=a_offset,
                                red=(x, xcoef), red_group=red_group)
        if r is not None:
            _register_pre_bn_sums(dx, [r])
        return dx
    return ops.mlp_gemm(w2c, dz, pro=pro, X2=X2, coef=coef, tag="dgrad", pool=pool, M=M, a_offset=a_offset)[0]


class _SharedMLPLayer(torch.autograd.Function):
    """1x1 convolution (+bias) -> BatchNorm (batch statistics) -> ReLU as ONE autograd node on the
    HIP kernels of csrc/shared_mlp.hip (models/layers.py:208-216, :293-303 + autograd's backward).

      forward : GEMM (+bias, + per-channel sum/sum^2 partials in its epilogue; the input may be a
                LazyAct, activated in the GEMM's prologue) -> statistics finalise (+ running stats)
                -> BN+ReLU apply, or nothing when the output is handed on as a LazyAct
      backward: one reduction pass (dgamma, dbeta) -> data-gradient GEMM and weight-gradient GEMM
                that both rebuild dY = BN'(ReLU'(dZ)) from (dZ, Y) in their prologue; dY is never
                written to memory.
    The conv bias in front of a training-mode BatchNorm has an analytically zero gradient (the
    reference's autograd returns rounding noise there); zeros are returned.
    """

    @staticmethod
    def forward(ctx, x, xcoef, w2, bias, gamma, beta, running_mean, running_var, training, momentum, eps,
                relu, defer, sink, nograd_prefix=0, wsrc=None):
        ctx.sink = sink
        ctx.nograd_prefix = int(nograd_prefix)
        ctx.wsrc = wsrc                      # (an input tensor of the step: alive for its whole duration)
        ctx.set_materialize_grads(False)     # no zero-filled gradient for the (non-differentiable) coef output
        x = x.contiguous()
        # K-major copy of the weight [Cin][Cout]: the GEMM can also read W transposed in place (negative lda),
        # but the strided operand loads cost more (+0.3 ms/step measured) than these tiny copies
        wt = _kmajor(w2)
        nb, _, P = x.shape
        ctx.has_bn = gamma is not None
        ctx.relu = bool(relu)
        ctx.train_stats = bool(training)
        pro = 0 if xcoef is None else 1
        if not ctx.has_bn:
            if relu:
                raise NotImplementedError("usip_amd: ReLU without BatchNorm is outside the detector path")
            y, _ = ops.mlp_gemm(wt, x, bias, pro=pro, coef=xcoef)
            ctx.save_for_backward(x, xcoef, w2)
            return y, None
        if training:
            y, stats = ops.mlp_gemm(wt, x, bias, want_stats=True, pro=pro, coef=xcoef)
            mean, invstd, coef = ops.bn_finalize(stats, nb * P, gamma, beta, eps, momentum,
                                                 running_mean, running_var)
            if relu:
                y = _relu_hook(y, coef)
        else:
            y, _ = ops.mlp_gemm(wt, x, bias, pro=pro, coef=xcoef)
            invstd = torch.rsqrt(running_var + eps)
            mean = running_mean
            scale = gamma * invstd
            coef = torch.stack((scale, beta - running_mean * scale)).contiguous()
        ctx.save_for_backward(x, xcoef, w2, y, coef, mean, invstd, gamma)
        if defer:
            ctx.mark_non_differentiable(coef)
            return y, coef
        return ops.bn_apply(y, coef, relu), None

    @staticmethod
    def backward(ctx, dz, _dcoef):
        if dz is None:
            return (None,) * 16
        dz = dz.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
        tail = (None,) * 10
        sink = ctx.sink                      # (w.grad, b.grad[, gamma.grad, beta.grad]) or None
        if not ctx.has_bn:
            x, xcoef, w2 = ctx.saved_tensors
            dx = None
            if need_x:
                dx = _dgrad(x, w2.contiguous(), dz, pro=0)
            dw = db = None
            if need_w:
                dw = ops.mlp_wgrad(dz, x, xcoef=xcoef, out=sink[0].view(w2.shape) if sink else None)
            if ctx.needs_input_grad[3]:
                db = ops.bn_backward_reduce(dz, None, None, None, None, None, False,
                                            dbeta_out=sink[1] if sink else None)[1]
            if sink:
                dw = db = None               # already in .grad
            return (dx, None, dw, db, None, None) + tail
        if not ctx.train_stats:
            raise NotImplementedError("usip_amd: backward through eval-mode BatchNorm is outside the path")
        x, xcoef, w2, y, coef, mean, invstd, gamma = ctx.saved_tensors
        took = []
        dgamma, dbeta, coef4 = _own_bn_backward(dz, y, coef, mean, invstd, gamma, ctx.relu, sink, took=took)
        if not ctx.relu:
            # BatchNorm with no ReLU behind it: only the generic form of a layer built with a non-default activation
            # (layers._generic_forward).  The GEMM prologues rebuild dY WITH the ReLU decision (mlp_common.h::pro_apply),
            # so dY = a1*dZ + q1*y + q0 is written out here and the plain gradient GEMMs run on it.
            C = dz.shape[1]
            dy = torch.addcmul(torch.addcmul(coef4[3].view(1, C, 1), y, coef4[2].view(1, C, 1)),
                               dz, coef4[0].view(1, C, 1))
            dx = _dgrad(x, w2.contiguous(), dy, pro=0) if need_x else None
            dw = None
            if need_w:
                dw = ops.mlp_wgrad(dy, x, xcoef=xcoef, out=sink[0].view(w2.shape) if sink else None)
            db = torch.zeros_like(gamma) if (ctx.needs_input_grad[3] and not sink) else None
            if sink:
                dw = db = dgamma = dbeta = None
            return (dx, None, dw, db, dgamma, dbeta) + tail
        x2 = (FUSED_NARROW_BWD and need_x and need_w and ctx.relu and ctx.nograd_prefix == 0
              and ops.layer_backward_x2_supported(x.shape[1], w2.shape[0], x.shape[2], (dz, y, x), coef4, xcoef))
        if x2 or (FUSED_NARROW_BWD and need_x and need_w and ctx.relu and ctx.nograd_prefix == 0
                  and ops.narrow_backward_supported(x.shape[1], w2.shape[0], x.shape[2], (dz, y, x))):
            red = FUSED_NARROW_RED and xcoef is not None and xcoef.shape[0] >= 4   # input = lazy activation of a train-mode BN layer
            if x2:                                           # f32x2 with the operand bounds at hand: csrc/layer_bwd_x2.hip
                res = ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2.contiguous(), Cin=x.shape[1],
                                                dw_out=sink[0].view(w2.shape) if sink else None, want_red=red,
                                                wsrc=ctx.wsrc if red else None)
            else:
                res = ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2.contiguous(),
                                              dw_out=sink[0].view(w2.shape) if sink else None, want_red=red)
            dx, dw = res[0], res[1]
            if red:
                _register_pre_bn_sums(dx, [res[2]])
            db = torch.zeros_like(gamma) if (ctx.needs_input_grad[3] and not sink) else None
            if sink:
                dw = db = dgamma = dbeta = None
            return (dx, None, dw, db, dgamma, dbeta) + tail
        dx = None
        if need_x:
            pre = ctx.nograd_prefix
            if pre > 0:
                # the first `pre` input channels carry no gradient (neighbour coordinates, layers.py:422-430):
                # only the remaining rows of dX are computed, straight into their slice of the full tensor --
                # for 3 + 128 channels that is one 128-row tile instead of two
                dx = torch.empty_like(x)
                dx[:, :pre].zero_()
                ops.mlp_gemm(w2.contiguous(), dz, pro=2, X2=y, coef=coef4, tag="dgrad", M=x.shape[1] - pre,
                             a_offset=pre, out=dx, out_row_offset=pre)
            else:
                dx = _dgrad(x, w2.contiguous(), dz, pro=2, X2=y, coef=coef4, xcoef=xcoef)
        dw = None
        if need_w:
            # the detector's first layer (gradient-free input of <= 8 rows): the fused backward of the NEXT layer took the
            # sums dW follows from while it had (dYhat, y) of this one in LDS (csrc/layer_bwd_x2.hip, WS) -- no pass over
            # this layer's (dZ, Y) at all
            pack = took[0].wsum if (len(took) == 1 and getattr(took[0], "wsum", None) is not None) else None
            if (pack is not None and not need_x and xcoef is None and pack[2].data_ptr() == x.data_ptr()
                    and tuple(pack[2].shape) == tuple(x.shape) and w2.shape[1] == x.shape[1] and w2.is_contiguous()):
                dw = ops.wsum_finalize(pack, coef4, mean, sink[0].view(w2.shape) if sink else torch.empty_like(w2))
            else:
                dw = ops.mlp_wgrad(dz, x, pro=2, G2=y, coef4=coef4, xcoef=xcoef,
                                   out=sink[0].view(w2.shape) if sink else None)
        db = torch.zeros_like(gamma) if (ctx.needs_input_grad[3] and not sink) else None
        if sink:
            dw = db = dgamma = dbeta = None  # written in place; the bias gradient is the zero already there
        return (dx, None, dw, db, dgamma, dbeta) + tail


class _SharedMLPLayerMax(torch.autograd.Function):
    """conv1x1 -> BatchNorm -> ReLU -> max over the K neighbours as ONE autograd node, for layers whose
    output feeds only the pooling (conv5 of the ball detector, the last layer of the KNN fusion):
      forward : GEMM (+stats) -> finalise -> fused BN+ReLU+max straight from the GEMM output
      backward: the incoming gradient is dpooled [B,C,M]; dZ = (k == arg) ? dpooled : 0 is never
                materialised -- the BN reduction visits only the B*C*M arg-max elements and the data- and
                weight-gradient GEMMs synthesise dZ in their prologue (PRO_BN_BWD_POOL)."""

    @staticmethod
    def forward(ctx, x, xcoef, dims, w2, bias, gamma, beta, running_mean, running_var, momentum, eps, sink):
        B, Cin, M, K = dims
        x3 = x.contiguous().view(B, Cin, M * K)
        Cout = w2.shape[0]
        y, stats = ops.mlp_gemm(_kmajor(w2), x3, bias, want_stats=True,
                                pro=0 if xcoef is None else 1, coef=xcoef)
        mean, invstd, coef = ops.bn_finalize(stats, B * M * K, gamma, beta, eps, momentum, running_mean, running_var)
        y = _relu_hook(y, coef)
        pooled, arg, yarg = _pooled_act(y.view(B, Cout, M, K), coef, True, want_yarg=True)
        ctx.save_for_backward(x3, xcoef, w2, y, coef, mean, invstd, gamma, arg, yarg)
        ctx.dims, ctx.sink, ctx.x_shape = (B, Cin, Cout, M, K), sink, tuple(x.shape)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x3, xcoef, w2, y, coef, mean, invstd, gamma, arg, yarg = ctx.saved_tensors
        B, Cin, Cout, M, K = ctx.dims
        sink = ctx.sink
        dpooled = dpooled.contiguous()
        dgamma, dbeta, coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(B, Cout, M, K), coef, mean, invstd,
                                                           gamma, True, dgamma_out=sink[2] if sink else None,
                                                           dbeta_out=sink[3] if sink else None, yarg=yarg)
        pool = (dpooled, arg, K)
        dx = dw = None
        if (FUSED_NARROW_BWD and ctx.needs_input_grad[0] and ctx.needs_input_grad[3]
                and ops.layer_backward_x2_supported(Cin, Cout, M * K, (y, x3), coef4, xcoef, pooled=True)):
            # f32x2: data and weight gradient from ONE pass over (Y, X) -- csrc/layer_bwd_x2.hip
            # ... which also leaves the producing layer's BatchNorm-backward sums (per channel and per neighbourhood)
            red = (POOLED_LAYER_RED and FUSED_NARROW_RED and xcoef is not None and xcoef.shape[0] >= 4 and K % 32 == 0
                   and bool(GRAD_SINK))
            res = ops.mlp_layer_backward_x2(None, y, coef4, x3, xcoef, w2.contiguous(), Cin=Cin, pool=pool,
                                            dw_out=sink[0].view(w2.shape) if sink else None, want_red=red, want_gsum=red)
            dx, dw = res[0], res[1]
            if red:
                _register_pre_bn_sums(dx, [res[2]])
            db = torch.zeros_like(gamma) if (ctx.needs_input_grad[4] and not sink) else None
            if sink:
                dw = db = dgamma = dbeta = None
            return (dx.view(ctx.x_shape), None, None, dw, db, dgamma, dbeta) + (None,) * 5
        if ctx.needs_input_grad[0]:
            dx = _dgrad(x3, w2.contiguous(), None, pro=3, X2=y, coef=coef4, pool=pool, xcoef=xcoef,
                        red_group=K if (POOLED_LAYER_RED and K in (16, 32)) else 0)
            dx = dx.view(ctx.x_shape)
        if ctx.needs_input_grad[3]:
            dw = ops.mlp_wgrad(None, x3, pro=3, G2=y, coef4=coef4, xcoef=xcoef, pool=pool,
                               out=sink[0].view(w2.shape) if sink else None)
        db = torch.zeros_like(gamma) if (ctx.needs_input_grad[4] and not sink) else None
        if sink:
            dw = db = dgamma = dbeta = None
        return (dx, None, None, dw, db, dgamma, dbeta) + (None,) * 5


def conv1x1_bn_relu_max(x, weight: torch.Tensor, bias: Optional[torch.Tensor], bn) -> torch.Tensor:
    """max_k relu(bn(conv1x1(x))) for x [B,Cin,M,K] (tensor or LazyAct) -> [B,Cout,M]; the layer's
    activated output exists nowhere else, so neither it nor its gradient is ever materialised."""
    shape = x.shape
    K = shape[3]
    fused = (bn is not None and bn.training and bias is not None and _group_sums_supported(K)
             and torch.is_grad_enabled())
    if not fused:
        return group_max(conv1x1_bn_act(x, weight, bias, bn, True, defer=True))
    xcoef = None
    if isinstance(x, LazyAct):
        if not x.relu:
            x = x.materialize()
        else:
            x, xcoef = x.y, x.coef
    require_device(x, "the shared MLP")
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    if bn.num_batches_tracked is not None and not DEFER_BN_COUNTERS:
        bn.num_batches_tracked.add_(1)
    return _SharedMLPLayerMax.apply(x, xcoef, tuple(shape), w2, bias, bn.weight, bn.bias, bn.running_mean,
                                    bn.running_var, bn.momentum, bn.eps, _sink(weight, bias, bn.weight, bn.bias))


def _group_sums_supported(K: int) -> bool:
    lpg = K // 4
    return K >= 4 and K % 4 == 0 and (lpg & (lpg - 1)) == 0 and lpg <= 64


class _SharedMLPLayerPooled(torch.autograd.Function):
    """The shared-MLP layer whose input is cat(h, expand(pooled)) over channels -- the reference
    expands the max-pooled neighbourhood feature over K and concatenates it (networks.py:706-709,
    layers.py:433-435) -- WITHOUT building that tensor:

        W . [h ; pooled] = W_h . h  +  (W_p . pooled)[m]          (constant inside a neighbourhood)

    The second term is a GEMM over M positions instead of M*K and enters the big GEMM's epilogue as
    a per-neighbourhood row bias.  Backward: sum_k dY follows from per-neighbourhood sums the BN
    reduction pass already produces, so d(pooled) and dW_p are M-sized GEMMs as well.  Same math as
    the reference up to fp32 summation order; half the multiply-adds of that layer."""

    @staticmethod
    def forward(ctx, h, hcoef, dims, pooled, w2, bias, gamma, beta, running_mean, running_var, training, momentum,
                eps, relu, pooled_first, defer, sink):
        ctx.sink = sink
        B, Ch, M, K = dims
        Cp = pooled.shape[1]
        Cout = w2.shape[0]
        ctx.set_materialize_grads(False)
        poff, hoff = (0, Cp) if pooled_first else (Ch, 0)
        h3 = h.contiguous().view(B, Ch, M * K)
        pooled = pooled.contiguous()
        wt = _kmajor(w2)                                                    # [Ctot][Cout]
        r, _ = ops.mlp_gemm(wt[poff:poff + Cp], pooled, tag="fwd_pooled")   # [B,Cout,M]
        y, stats = ops.mlp_gemm(wt[hoff:hoff + Ch], h3, bias, want_stats=True, rowbias=r, rb_group=K,
                                pro=0 if hcoef is None else 1, coef=hcoef)
        mean, invstd, coef = ops.bn_finalize(stats, B * M * K, gamma, beta, eps, momentum,
                                             running_mean, running_var)
        if relu:
            y = _relu_hook(y, coef)
        ctx.save_for_backward(h3, hcoef, pooled, w2, y, coef, mean, invstd, gamma)
        ctx.dims = (B, Ch, Cp, Cout, M, K, poff, hoff)
        ctx.h_shape = tuple(h.shape)
        ctx.relu = bool(relu)
        if defer:
            ctx.mark_non_differentiable(coef)
            return y, coef
        return ops.bn_apply(y, coef, relu), None

    @staticmethod
    def backward(ctx, dz, _dcoef):
        if dz is None:
            return (None,) * 17
        h3, hcoef, pooled, w2, y, coef, mean, invstd, gamma = ctx.saved_tensors
        B, Ch, Cp, Cout, M, K, poff, hoff = ctx.dims
        dz = dz.contiguous().view(B, Cout, M * K)
        sink = ctx.sink
        # this layer needs the per-neighbourhood sums as well, which only the stand-alone pass produces
        dgamma, dbeta, coef4, gsum = _own_bn_backward(dz, y, coef, mean, invstd, gamma, ctx.relu, sink, group=K)
        # sum over the K neighbours of dY = a1*dYhat + q1*y + q0
        sdy = ops.bn_group_dy_sum(gsum, coef4, K)
        w2c = w2.contiguous()
        dpooled = dh = dw = None
        if ctx.needs_input_grad[3]:
            dpooled = ops.mlp_gemm(w2c, sdy, tag="dgrad_pooled", M=Cp, a_offset=poff)[0]
        want = FUSED_NARROW_BWD and ctx.needs_input_grad[0] and ctx.needs_input_grad[4] and ctx.relu
        x2 = want and ops.layer_backward_x2_supported(Ch, Cout, M * K, (dz, y, h3), coef4, hcoef)
        fused = x2 or (want and ops.narrow_backward_supported(Ch, Cout, M * K, (dz, y, h3)))
        # (a dW that is not a view of the step's gradient bucket goes back to autograd: its sums are launched at once)
        if fused:                                           # data and weight gradient of the feature half in one pass
            dw = sink[0].view(w2c.shape) if sink else torch.empty_like(w2c)
            red = FUSED_NARROW_RED and hcoef is not None and hcoef.shape[0] >= 4
            back = ops.mlp_layer_backward_x2 if x2 else ops.mlp_narrow_backward      # f32x2 (csrc/layer_bwd_x2.hip) / fp32 MFMA
            with ops.reduce_now(not sink):
                res = back(dz, y, coef4, h3, hcoef, w2c, wcol=hoff, dw_out=dw, Cin=Ch, want_red=red)
                ops.mlp_wgrad(sdy, pooled, out=dw, coloff=poff)
            if red:
                _register_pre_bn_sums(res[0], [res[2]])
            dh = res[0].view(ctx.h_shape)
        if ctx.needs_input_grad[0] and not fused:
            dh = _dgrad(h3, w2c, dz, pro=2, X2=y, coef=coef4, M=Ch, a_offset=hoff, xcoef=hcoef).view(ctx.h_shape)
        if ctx.needs_input_grad[4] and not fused:
            dw = sink[0].view(w2c.shape) if sink else torch.empty_like(w2c)
            with ops.reduce_now(not sink):
                ops.mlp_wgrad(dz, h3, pro=2, G2=y, coef4=coef4, out=dw, coloff=hoff, xcoef=hcoef)
                ops.mlp_wgrad(sdy, pooled, out=dw, coloff=poff)
        db = torch.zeros_like(gamma) if (ctx.needs_input_grad[5] and not sink) else None
        if sink:
            dw = db = dgamma = dbeta = None
        return (dh, None, None, dpooled, dw, db, dgamma, dbeta) + (None,) * 9


def conv1x1_bn_act_pooled(h, pooled: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], bn,
                          relu: bool, pooled_first: bool, defer: bool = False):
    """conv1x1_bn_act(cat((expand(pooled), h) if pooled_first else (h, expand(pooled)), dim=1)) without the
    concatenated tensor.  h [B,Ch,M,K] (tensor or LazyAct), pooled [B,Cp,M].  defer=True returns a LazyAct."""
    hshape = h.shape
    K = hshape[3]
    fused = (bn is not None and bn.training and bias is not None and _group_sums_supported(K)
             and torch.is_grad_enabled())
    if not fused:
        ht = as_tensor(h)
        e = pooled.unsqueeze(3).expand(-1, -1, -1, K)
        return conv1x1_bn_act(torch.cat((e, ht) if pooled_first else (ht, e), dim=1), weight, bias, bn, relu,
                              defer=defer)
    hcoef = None
    if isinstance(h, LazyAct):
        if not h.relu:
            h = h.materialize()
        else:
            h, hcoef = h.y, h.coef
    require_device(h, "the shared MLP")
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    if bn.num_batches_tracked is not None and not DEFER_BN_COUNTERS:
        bn.num_batches_tracked.add_(1)
    out, coef = _SharedMLPLayerPooled.apply(h, hcoef, tuple(hshape), pooled, w2, bias, bn.weight, bn.bias,
                                            bn.running_mean, bn.running_var, True, bn.momentum, bn.eps, relu,
                                            pooled_first, defer, _sink(weight, bias, bn.weight, bn.bias))
    oshape = (hshape[0], w2.shape[0], hshape[2], K)
    return LazyAct(out, coef, relu, oshape) if defer else out.view(oshape)


def conv1x1_bn_act(x, weight: torch.Tensor, bias: Optional[torch.Tensor],
                   bn: Optional[torch.nn.modules.batchnorm._BatchNorm], relu: bool, defer: bool = False,
                   nograd_prefix: int = 0):
    """One shared-MLP layer: 1x1 convolution (+bias) -> BatchNorm (batch statistics when the
    module is in training mode) -> ReLU  (models/layers.py:208-216, :293-303).
    x [B,Cin,*positions] (tensor or LazyAct), weight [Cout,Cin,1(,1)] -> [B,Cout,*positions];
    defer=True (BatchNorm layers only) returns a LazyAct instead of the activated tensor.
    nograd_prefix: the first so many input channels need no gradient (their rows of dX are zeros, not computed)."""
    shape = x.shape
    xcoef = wsrc_in = None
    if isinstance(x, LazyAct):
        if not x.relu:
            x = x.materialize()
        else:
            x, xcoef, wsrc_in = x.y, x.coef, x.wsrc
    require_device(x, "the shared MLP")
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    x3 = x.reshape(shape[0], shape[1], -1)
    oshape = (shape[0], w2.shape[0]) + tuple(shape[2:])
    if bn is None:
        y, _ = _SharedMLPLayer.apply(x3, xcoef, w2, bias, None, None, None, None, False, 0.0, 0.0, relu, False,
                                     _sink(weight, bias))
        return y.view(oshape)
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None and not DEFER_BN_COUNTERS:
        bn.num_batches_tracked.add_(1)
    y, coef = _SharedMLPLayer.apply(x3, xcoef, w2, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    training, bn.momentum, bn.eps, relu, defer,
                                    _sink(weight, bias, bn.weight, bn.bias) if training else None, nograd_prefix, wsrc_in)
    if not defer:
        return y.view(oshape)
    # a first layer (plain input of <= 8 rows that needs no gradient): its consumer's fused backward may take this layer's
    # weight-gradient sums on the way (LazyAct.wsrc)
    first = (training and relu and xcoef is None and not x3.requires_grad and x3.shape[1] <= 8 and x3.is_contiguous()
             and torch.is_grad_enabled())
    return LazyAct(y, coef, relu, oshape, wsrc=x3 if first else None)


# --------------------------------------------------------------------------- grouping / pooling
class _GroupMax(torch.autograd.Function):
    """max over the K neighbours (torch.max(dim=3), networks.py:706,710, layers.py:433,438); the
    gradient goes to the first arg-max."""

    @staticmethod
    def forward(ctx, z):
        pooled, arg = _pooled_act(z.contiguous(), None, False)
        ctx.save_for_backward(arg)
        ctx.K = z.shape[3]
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        (arg,) = ctx.saved_tensors
        return ops.group_max_backward(dpooled.contiguous(), arg, ctx.K)


class _GroupMaxAct(torch.autograd.Function):
    """BN-apply + ReLU + max over K in one pass over the producing layer's pre-BN output."""

    @staticmethod
    def forward(ctx, y4, coef, relu):
        pooled, arg = _pooled_act(y4.contiguous(), coef, relu)
        ctx.save_for_backward(arg)
        ctx.K = y4.shape[3]
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        (arg,) = ctx.saved_tensors
        return ops.group_max_backward(dpooled.contiguous(), arg, ctx.K), None, None


class _GroupMaxActFork(torch.autograd.Function):
    """max over K of a lazily activated layer output that ALSO feeds another layer directly
    (networks.py:706-709, layers.py:433-436: the pooled feature is concatenated back onto its own source).
    Returns (pooled, alias of y); in backward the sparse pooling gradient is added into the dense gradient
    that arrives for the alias, in place, instead of being written as a second dense tensor and summed."""

    @staticmethod
    def forward(ctx, y4, coef, relu):
        pooled, arg, yarg = _pooled_act(y4.contiguous(), coef, relu, want_yarg=True)
        ctx.save_for_backward(arg, y4, coef, yarg)
        ctx.K = y4.shape[3]
        ctx.relu = bool(relu)
        ctx.set_materialize_grads(False)
        return pooled, y4.view_as(y4)

    @staticmethod
    def backward(ctx, dpooled, dy):
        arg, y4, coef, yarg = ctx.saved_tensors
        if dpooled is None:
            return dy, None, None
        if dy is None:
            return ops.group_max_backward(dpooled.contiguous(), arg, ctx.K), None, None
        B, C, M, K = y4.shape
        pre = _take_pre_bn_sums(dy, (B, C, M * K))            # sums the producer of dy took for the dense part
        # Inside the training step (GRAD_SINK: no hooks, no retained graph, one consumer) dy is the data gradient the
        # consuming layer just produced for this node alone, so it is updated in place; anywhere else autograd may
        # share that buffer (retain_grad, hooks, a second consumer), so the sum goes into a private copy
        dy = dy.contiguous() if GRAD_SINK else dy.clone(memory_format=torch.contiguous_format)
        dpooled = dpooled.contiguous()
        out = ops.group_max_backward_add_(dy, dpooled, arg)
        if pre is not None and coef.shape[0] >= 4 and GRAD_SINK:
            # the sums are linear in the gradient: add those of the sparse pooling part (B*C*M elements)
            sparse = ops.bn_pool_backward_partials(dpooled, arg, y4, coef, coef[2], coef[3], ctx.relu, yarg=yarg)
            out3 = out.view(B, C, M * K)
            PRE_BN_SUMS[out3.data_ptr()] = ((B, C, M * K), pre[1] + [sparse], out3._version)
        return out, None, None


def group_max_fork(z):
    """(max over K, z) for a tensor that is pooled AND passed on: for a LazyAct the returned z is an alias whose
    gradient is combined with the pooling gradient in one sparse update (see _GroupMaxActFork)."""
    if isinstance(z, LazyAct) and _group_sums_supported(z.shape[3]):
        pooled, y = _GroupMaxActFork.apply(z.y.view(z.shape), z.coef, z.relu)
        return pooled, LazyAct(y.view(z.y.shape), z.coef, z.relu, z.shape)
    return group_max(z), z


def group_max(z) -> torch.Tensor:
    """z [B,C,M,K] (tensor or LazyAct) -> [B,C,M]."""
    if isinstance(z, LazyAct):
        K = z.shape[3]
        if _group_sums_supported(K):
            return _GroupMaxAct.apply(z.y.view(z.shape), z.coef, z.relu)
        z = z.materialize()
    require_device(z, "group_max")
    return _GroupMax.apply(z)


class _KnnGroup(torch.autograd.Function):
    """cat(gather(database, I) - query, gather(feat, I)) as one tensor [B, 3+C, M, K]
    (models/layers.py:422-430); only the features carry a gradient (coordinates are detached there)."""

    @staticmethod
    def forward(ctx, feat, database, query, idx32):
        B, C, N = feat.shape
        _, M, K = idx32.shape
        out = torch.empty((B, 3 + C, M, K), dtype=torch.float32, device=feat.device)
        ops.group_gather(database.contiguous(), idx32, sub=query.contiguous(), out=out, coff=0)
        ops.group_gather(feat.contiguous(), idx32, out=out, coff=3)
        ctx.dims = (C, N)
        ctx.csr = SEGMENT_BACKWARD and feat.requires_grad and ops.segment_sum_supported(N, M * K)
        if ctx.csr:                            # the neighbour lists sorted by the node they point at, for the backward
            ctx.save_for_backward(*ops.csr_by_index(idx32.view(B, M * K), N))
        else:
            ctx.save_for_backward(idx32)
        return out

    @staticmethod
    def backward(ctx, dout):
        C, N = ctx.dims
        if ctx.csr:
            start, perm = ctx.saved_tensors
            return ops.segment_sum(dout.contiguous(), start, perm, C, coff=3), None, None, None
        (idx32,) = ctx.saved_tensors
        return ops.group_gather_backward(dout.contiguous(), idx32, C, N, coff=3), None, None, None


# Round 6: the first layer of GeneralKNNFusionModule without the gathered tensor (csrc/knn_layer.hip); USIP_KNN_LAYER=0
# keeps the gather + generic layer (A/B runs, and the form the new one is tested against).
KNN_FIRST_LAYER = _os.environ.get("USIP_KNN_LAYER", "1") not in ("0", "off")


class _KnnFirstLayer(torch.autograd.Function):
    """layers_before[0](cat(gather(database, I) - query, gather(feat, I))) of GeneralKNNFusionModule
    (models/layers.py:422-431 + :208-216: conv1x1 + BatchNorm (batch statistics) + ReLU) as ONE node that never builds the
    B x (3+C) x M x K tensor:  W . [d ; feat[:, n]] = W_c . d + (W_f . feat)[:, n]  -- the feature half of the product is
    taken over the N database points (an M-sized GEMM), gathered by the neighbour index and completed with the three
    coordinate terms in one pass that writes the layer's pre-BN output and its statistics.  Backward: one pass over
    (dZ, Y) forms dY, its segment sums over the neighbour lists (fixed order) and the coordinate columns of dW; d feat
    and the feature columns of dW are M-sized products again.  Same math as the reference up to fp32 summation order.
    Returns (pre-BN output [B,Cout,M*K], coef) = the parts of a LazyAct."""

    @staticmethod
    def forward(ctx, feat, database, query, idx32, w2, bias, gamma, beta, running_mean, running_var, momentum, eps, sink):
        B, C, N = feat.shape
        _, M, K = idx32.shape
        ctx.set_materialize_grads(False)
        feat = feat.contiguous()
        database, query = database.contiguous(), query.contiguous()
        w2c = w2.contiguous()
        U, _ = ops.mlp_gemm(_kmajor(w2)[3:], feat, bias, tag="fwd_knn_nodes")          # [B,Cout,N] = W_f . feat + bias
        y, stats = ops.knn_layer_forward(U, w2c, database, query, idx32)
        mean, invstd, coef = ops.bn_finalize(stats.view(2, w2.shape[0], B), B * M * K, gamma, beta, eps, momentum,
                                             running_mean, running_var)
        y = _relu_hook(y, coef)
        start, perm = ops.csr_by_index(idx32.view(B, M * K), N)
        dcoord = ops.group_gather(database, idx32, sub=query)                      # [B,3,M,K], layers.py:428-430
        ctx.save_for_backward(feat, dcoord, w2, y, coef, mean, invstd, gamma, start, perm)
        ctx.sink = sink
        ctx.mark_non_differentiable(coef)
        return y, coef

    @staticmethod
    def backward(ctx, dz, _dcoef):
        if dz is None:
            return (None,) * 13
        feat, dcoord, w2, y, coef, mean, invstd, gamma, start, perm = ctx.saved_tensors
        sink = ctx.sink
        C = feat.shape[1]
        dz = dz.contiguous().view(y.shape)
        dgamma, dbeta, coef4 = _own_bn_backward(dz, y, coef, mean, invstd, gamma, True, sink)
        dU, dwc = ops.knn_layer_backward(dz, y, coef4, True, dcoord.view(dcoord.shape[0], 3, -1), start, perm,
                                         dcoord.shape[2], dcoord.shape[3])
        w2c = w2.contiguous()
        dfeat = dw = None
        if ctx.needs_input_grad[0]:
            dfeat = ops.mlp_gemm(w2c, dU, tag="dgrad_knn_nodes", M=C, a_offset=3)[0]     # W_f^T . dU
        if ctx.needs_input_grad[4]:
            dw = sink[0].view(w2c.shape) if sink else torch.empty_like(w2c)
            with ops.reduce_now(not sink):                                                # (private dW: summed at once)
                ops.mlp_wgrad(dU, feat, out=dw, coloff=3)                                 # dU . feat^T -> columns 3..
            dw[:, :3].copy_(dwc)
        db = torch.zeros_like(gamma) if (ctx.needs_input_grad[5] and not sink) else None  # in front of BatchNorm: zero
        if sink:
            dw = db = dgamma = dbeta = None
        return (dfeat, None, None, None, dw, db, dgamma, dbeta) + (None,) * 5


def knn_first_layer_supported(feat, idx32, bias, bn, relu: bool) -> bool:
    return (KNN_FIRST_LAYER and relu and bias is not None and bn is not None and bn.training and torch.is_grad_enabled()
            and isinstance(feat, torch.Tensor) and feat.is_cuda
            and ops.knn_layer_supported(feat.shape[2], idx32.shape[1], idx32.shape[2]))


def knn_first_layer(feat, database, query, idx32, weight, bias, bn) -> "LazyAct":
    """relu(bn(conv1x1(cat(gather(database, I) - query, gather(feat, I))))) as a LazyAct [B,Cout,M,K]
    (knn_first_layer_supported; coordinates carry no gradient, models/layers.py:428-430 operates on detached inputs
    there as here)."""
    require_device(feat, "knn_first_layer")
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    if bn.num_batches_tracked is not None and not DEFER_BN_COUNTERS:
        bn.num_batches_tracked.add_(1)
    y, coef = _KnnFirstLayer.apply(feat, database.detach(), query.detach(), idx32, w2, bias, bn.weight, bn.bias,
                                   bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                   _sink(weight, bias, bn.weight, bn.bias))
    return LazyAct(y, coef, True, (feat.shape[0], w2.shape[0], idx32.shape[1], idx32.shape[2]))


class _ClusterBroadcast(torch.autograd.Function):
    """out[b,c,n] = x[b,c,idx[b,n]]: every point receives its SOM node's feature (models/networks.py:119-125,
    torch.gather on an index expanded over the channels).  Backward = the sum over each node's member points,
    through the reproducible LDS scatter of csrc/group.hip instead of ATen's float atomics."""

    @staticmethod
    def forward(ctx, x, idx32, start, perm):
        B, C, M = x.shape
        N = idx32.shape[1]
        ctx.csr = start is not None
        ctx.save_for_backward(*((start, perm) if ctx.csr else (idx32,)))
        ctx.dims = (C, M)
        return ops.group_gather(x.contiguous(), idx32.view(B, N, 1)).view(B, C, N)

    @staticmethod
    def backward(ctx, dout):
        C, M = ctx.dims
        B, _, N = dout.shape
        if ctx.csr:
            start, perm = ctx.saved_tensors
            return ops.segment_sum(dout.contiguous(), start, perm, C), None, None, None
        (idx32,) = ctx.saved_tensors
        return ops.group_gather_backward(dout.contiguous().view(B, C, N, 1), idx32.view(B, N, 1), C, M), None, None, None


def cluster_broadcast(x, idx32, csr=None):
    """x [B,C,M] node features, idx32 i32 [B,N] node of every point -> [B,C,N].  csr: (start, perm) of idx32
    (ops.csr_by_index) when the caller has it -- the backward then sums sorted segments."""
    require_device(x, "cluster_broadcast")
    if csr is None and SEGMENT_BACKWARD and x.requires_grad and ops.segment_sum_supported(x.shape[2], idx32.shape[1]):
        csr = ops.csr_by_index(idx32, x.shape[2])
    start, perm = csr if csr is not None else (None, None)
    return _ClusterBroadcast.apply(x, idx32, start, perm)


class _SomPoolLayer(torch.autograd.Function):
    """The plain (no BatchNorm, no ReLU) last layer of a SOM PointNet together with what the reference does to its
    output (models/networks.py:114-133):

        y = conv1x1(x);  idx = index_max(y, min_idx);  y_max = gather(y, idx) * mask_row_max
        concat:  out = cat(y, gather(y_max, min_idx))          [B, 2C, N]     (first PointNet, :117-125)
        else  :  out = y_max                                   [B, C, M]      (second PointNet, :130-133)

    as one autograd node: the GEMM writes y into the top rows of the concatenated tensor, index_max emits the masked
    values with the indices (the winning key already holds the value), the broadcast fills the bottom rows -- no
    torch.gather, mask multiply or torch.cat launches.  Backward: the broadcast's gradient is a sum over sorted
    segments (csrc/segment.hip), the gather's gradient is B*C*M elements added INTO the dense gradient of y (for the
    second PointNet: into zeros), and the layer's own data/weight/bias gradients follow from that tensor."""

    @staticmethod
    def forward(ctx, x, xcoef, w2, bias, min_idx32, count, start, perm, M, concat, sink):
        x = x.contiguous()
        nb, _, N = x.shape
        C = w2.shape[0]
        wt = _kmajor(w2)
        pro = 0 if xcoef is None else 1
        ctx.set_materialize_grads(False)
        if concat:
            out = torch.empty((nb, 2 * C, N), dtype=torch.float32, device=x.device)
            ops.mlp_gemm(wt, x, bias, pro=pro, coef=xcoef, out=out, out_row_offset=0)
            idx, val = ops.index_max_values(out, min_idx32, count, M, C=C)
            ops.group_gather(val, min_idx32.view(nb, N, 1), out=out.view(nb, 2 * C, N, 1), coff=C)
        else:
            y, _ = ops.mlp_gemm(wt, x, bias, pro=pro, coef=xcoef)
            idx, out = ops.index_max_values(y, min_idx32, count, M)
        ctx.save_for_backward(x, xcoef, w2, idx, count, start, perm, min_idx32)
        ctx.dims, ctx.concat, ctx.sink = (nb, C, N, int(M)), bool(concat), sink
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, dout, _didx):
        if dout is None:
            return (None,) * 11
        x, xcoef, w2, idx, count, start, perm, min_idx32 = ctx.saved_tensors
        nb, C, N, M = ctx.dims
        sink = ctx.sink
        dout = dout.contiguous()
        if ctx.concat:
            g = ops.segment_sum(dout, start, perm, C, coff=C)            # d(y_max): empty nodes have empty segments
            # dy = the top rows of dout + the gather's scatter term, as the contiguous tensor the GEMMs below read
            dz = ops.index_max_values_backward(g, idx, count, min_idx32, N, src=dout, soff=0)
        else:
            dz = ops.index_max_values_backward(dout, idx, count, min_idx32, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(x, w2.contiguous(), dz, pro=0)
        if ctx.needs_input_grad[2]:
            dw = ops.mlp_wgrad(dz, x, xcoef=xcoef, out=sink[0].view(w2.shape) if sink else None)
        if ctx.needs_input_grad[3]:
            db = ops.bn_backward_reduce(dz, None, None, None, None, None, False, dbeta_out=sink[1] if sink else None)[1]
        if sink:
            dw = db = None
        return (dx, None, dw, db) + (None,) * 7


def som_pool_layer(x, weight, bias, min_idx32, count, csr, M: int, concat: bool):
    """x [B,Cin,N] (tensor or LazyAct) -> (cat(y, broadcast(y_max)) [B,2C,N] or y_max [B,C,M], index_max i32 [B,C,M])
    for y = conv1x1(x); see _SomPoolLayer.  csr = ops.csr_by_index(min_idx32, M)."""
    xcoef = None
    if isinstance(x, LazyAct):
        if not x.relu:
            x = x.materialize()
        else:
            x, xcoef = x.y, x.coef
    require_device(x, "the SOM pooling layer")
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    return _SomPoolLayer.apply(x, xcoef, w2, bias, min_idx32, count, csr[0], csr[1], int(M), bool(concat),
                               _sink(weight, bias))


def knn_group(feat, database, query, idx32):
    require_device(feat, "knn_group")
    return _KnnGroup.apply(feat, database, query, idx32)
