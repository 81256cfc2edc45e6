"""nn.Module surface of the shared-MLP path: same class names, constructor signatures and
state_dict keys as the reference's models/layers.py, for the classes its detectors instantiate
(SURVEY 8 a-5, a-6, a-7, a-13), so released checkpoints load and models/networks.py builds on
them unchanged.  The arithmetic lives in usip_amd.functional / the HIP library.

1x1 kernels only.  The options every train_detector.py of the reference defaults to ('batch' or no normalisation,
'relu' or no activation) run the fused kernels (lazy BatchNorm + ReLU, layer + max nodes, row-bias GEMMs).  The
other values models/layers.py accepts -- activation 'elu' | 'swish' | 'leakyrelu' | 'selu', normalization
'instance' -- run a GENERIC form (round 6): the convolution on the HIP GEMM, BatchNorm through the HIP
statistics kernels, instance normalisation and the activation as plain device-tensor operations, the
expand + cat + max sequences literally as the reference writes them -- same results as the reference (fixtures
detector_ball_elu_instance / detector_som_swish), none of the fusions.  Unknown strings raise NotImplementedError.
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from . import functional as Fh


class _EpochDecayBatchNorm(_BatchNorm):
    """BatchNorm whose momentum decays with the epoch (models/layers.py:49-71, :100-121):
    momentum = max(0.01, momentum0 * decay ** (epoch // step)) once epoch >= 1 and step > 0."""

    _dims = ()

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 momentum_decay_step=None, momentum_decay=1):
        super().__init__(num_features, eps, momentum, affine)
        self.momentum_decay_step = momentum_decay_step
        self.momentum_decay = momentum_decay
        self.momentum_original = self.momentum
        self.epoch_driven = False

    def _check_input_dim(self, input):
        if input.dim() not in self._dims:
            raise ValueError("expected %s input (got %dD input)" %
                             (" or ".join("%dD" % d for d in self._dims), input.dim()))

    def momentum_for(self, epoch):
        """The momentum a forward at `epoch` switches to, or None when the rule does not apply (the module then
        keeps whatever momentum it has -- the reference never resets it, layers.py:61-66)."""
        if epoch is not None and epoch >= 1 and self.momentum_decay_step is not None \
                and self.momentum_decay_step > 0:
            return max(0.01, self.momentum_original * self.momentum_decay ** (epoch // self.momentum_decay_step))
        return None

    def decay_momentum(self, epoch):
        if epoch is not None:
            self.epoch_driven = True          # this module is handed the epoch by its caller (not all are)
        m = self.momentum_for(epoch)
        if m is not None:
            self.momentum = m

    def forward(self, input, epoch=None):
        self._check_input_dim(input)
        self.decay_momentum(epoch)
        return torch.nn.functional.batch_norm(input, self.running_mean, self.running_var, self.weight,
                                              self.bias, self.training, self.momentum, self.eps)


class MyBatchNorm1d(_EpochDecayBatchNorm):
    _dims = (2, 3)


class MyBatchNorm2d(_EpochDecayBatchNorm):
    _dims = (4,)


class Swish(nn.Module):
    """models/layers.py:15-20."""

    def forward(self, x):
        return 1.78718727865 * (x * torch.sigmoid(x) - 0.20662096414)


_ACTIVATIONS = {"relu": nn.ReLU, "elu": lambda: nn.ELU(alpha=1.0), "swish": Swish,
                "leakyrelu": lambda: nn.LeakyReLU(0.01), "selu": nn.SELU}          # layers.py:181-191, :262-272


def _check_supported(activation, normalization):
    """-> True when the layer takes the GENERIC (unfused) form: any activation but ReLU, or instance normalisation."""
    if activation is not None and activation not in _ACTIVATIONS:
        raise NotImplementedError("usip_amd: unknown activation %r" % (activation,))
    if normalization not in (None, "batch", "instance"):
        raise NotImplementedError("usip_amd: unknown normalization %r" % (normalization,))
    return activation not in (None, "relu") or normalization == "instance"


def _generic_forward(layer, x, epoch):
    """conv -> norm -> act of a layer built with non-default options (models/layers.py:208-216, :293-303): the 1x1
    convolution (and a 'batch' normalisation's statistics) on the HIP kernels, the rest as device-tensor operations."""
    x = Fh.as_tensor(x)
    bn = layer.norm if layer.normalization == "batch" else None
    if bn is not None:
        bn.decay_momentum(epoch)
    y = Fh.conv1x1_bn_act(x, layer.conv.weight, layer.conv.bias, bn, False)
    if layer.normalization == "instance":
        y = layer.norm(y)
    if layer.activation is not None:
        y = layer.act(y)
    return y


def is_generic(layer) -> bool:
    return bool(getattr(layer, "_generic", False))


def pooled_concat_layer(layer, h, pooled, pooled_first: bool, epoch=None, defer: bool = True):
    """layer(cat(expand(pooled), h) or cat(h, expand(pooled))) -- models/layers.py:433-435, networks.py:706-709: the fused
    row-bias form for a BatchNorm + ReLU (or plain) layer, the literal expand + cat for a generic one."""
    if is_generic(layer):
        ht = Fh.as_tensor(h)
        e = pooled.unsqueeze(3).expand(-1, -1, -1, ht.shape[3])
        return layer(torch.cat((e, ht) if pooled_first else (ht, e), dim=1), epoch)
    bn = getattr(layer, "norm", None)
    if bn is not None:
        bn.decay_momentum(epoch)
    return Fh.conv1x1_bn_act_pooled(h, pooled, layer.conv.weight, layer.conv.bias, bn, layer.activation == "relu",
                                    pooled_first=pooled_first, defer=defer)


def _init_conv(conv, fan_in):
    conv.weight.data.normal_(0, math.sqrt(2.0 / fan_in))        # layers.py:196-205, :278-287
    if conv.bias is not None:
        conv.bias.data.zero_()


class MyConv2d(nn.Module):
    """1x1 Conv2d + BatchNorm2d + ReLU over grouped neighbourhoods B x C x M x K
    (models/layers.py:172-216)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True,
                 activation=None, normalization=None, momentum=0.1, bn_momentum_decay_step=None,
                 bn_momentum_decay=1):
        super().__init__()
        self._generic = _check_supported(activation, normalization)
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        if tuple(ks) != (1, 1) or stride not in (1, (1, 1)) or padding not in (0, (0, 0)):
            raise NotImplementedError("usip_amd: the shared MLP is a 1x1 convolution (stride 1, no padding)")
        self.activation = activation
        self.normalization = normalization
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias)
        if normalization == "batch":
            self.norm = MyBatchNorm2d(out_channels, momentum=momentum, affine=True,
                                      momentum_decay_step=bn_momentum_decay_step,
                                      momentum_decay=bn_momentum_decay)
        elif normalization == "instance":
            self.norm = nn.InstanceNorm2d(out_channels, momentum=momentum, affine=True)       # layers.py:189-190
        if activation is not None:
            self.act = _ACTIVATIONS[activation]()
        _init_conv(self.conv, in_channels)

    def forward(self, x, epoch=None, defer=False, nograd_prefix=0):
        """defer=True (internal use by the fused networks): return a functional.LazyAct.
        nograd_prefix: leading input channels that need no gradient (see functional.conv1x1_bn_act)."""
        if self._generic:
            return _generic_forward(self, x, epoch)
        bn = getattr(self, "norm", None)
        if bn is not None:
            bn.decay_momentum(epoch)
        return Fh.conv1x1_bn_act(x, self.conv.weight, self.conv.bias, bn, self.activation == "relu",
                                 defer=defer and bn is not None and self.activation == "relu",
                                 nograd_prefix=nograd_prefix)


class EquivariantLayer(nn.Module):
    """Conv1d(k=1) + BatchNorm1d + ReLU over points B x C x N (models/layers.py:248-303)."""

    def __init__(self, num_in_channels, num_out_channels, activation="relu", normalization=None,
                 momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        self._generic = _check_supported(activation, normalization)
        self.num_in_channels = num_in_channels
        self.num_out_channels = num_out_channels
        self.activation = activation
        self.normalization = normalization
        self.conv = nn.Conv1d(num_in_channels, num_out_channels, kernel_size=1, stride=1, padding=0)
        if normalization == "batch":
            self.norm = MyBatchNorm1d(num_out_channels, momentum=momentum, affine=True,
                                      momentum_decay_step=bn_momentum_decay_step,
                                      momentum_decay=bn_momentum_decay)
        elif normalization == "instance":
            self.norm = nn.InstanceNorm1d(num_out_channels, momentum=momentum, affine=True)   # layers.py:265-266
        if activation is not None:
            self.act = _ACTIVATIONS[activation]()
        _init_conv(self.conv, num_in_channels)

    def forward(self, x, epoch=None, defer=False):
        if self._generic:
            return _generic_forward(self, x, epoch)
        bn = getattr(self, "norm", None)
        if bn is not None:
            bn.decay_momentum(epoch)
        return Fh.conv1x1_bn_act(x, self.conv.weight, self.conv.bias, bn, self.activation == "relu",
                                 defer=defer and bn is not None and self.activation == "relu")


class PointNet(nn.Module):
    """Stack of EquivariantLayers; the last one has neither normalisation nor activation
    (models/layers.py:524-544)."""

    def __init__(self, in_channels, out_channels_list, activation, normalization, momentum=0.1,
                 bn_momentum_decay_step=None, bn_momentum_decay=1, output_init_radius=None):
        super().__init__()
        self.layers = nn.ModuleList()
        prev = in_channels
        last = len(out_channels_list) - 1
        for i, c_out in enumerate(out_channels_list):
            if i != last:
                self.layers.append(EquivariantLayer(prev, c_out, activation, normalization, momentum,
                                                    bn_momentum_decay_step, bn_momentum_decay))
            else:
                self.layers.append(EquivariantLayer(prev, c_out, None, None))
            prev = c_out
        if output_init_radius is not None:
            self.layers[last].conv.bias.data.uniform_(-output_init_radius, output_init_radius)

    def forward(self, x, epoch=None):
        # activations between layers stay lazy (BN+ReLU applied in the next GEMM's prologue)
        for layer in self.layers:
            x = layer(x, epoch, defer=True)
        return Fh.as_tensor(x)

    def forward_som_pooled(self, x, epoch, min_idx32, count, csr, M, concat):
        """The SOM detector's use of a PointNet (models/networks.py:114-133): its output goes through index_max,
        gather and mask, and (first PointNet) is concatenated with the broadcast maxima.  The plain last layer and
        those steps run as one node (functional._SomPoolLayer).
        -> (cat(out, broadcast(out_max)) [B,2C,N] if concat else out_max [B,C,M], index_max i32 [B,C,M])."""
        for layer in list(self.layers)[:-1]:
            x = layer(x, epoch, defer=True)
        last = self.layers[-1]
        if getattr(last, "norm", None) is not None or last.activation is not None:
            raise NotImplementedError("usip_amd: forward_som_pooled expects a plain last layer (layers.py:524-544)")
        return Fh.som_pool_layer(x, last.conv.weight, last.conv.bias, min_idx32, count, csr, M, concat)


class GeneralKNNFusionModule(nn.Module):
    """query -> database KNN, gather, shared MLP, max over K, concat, shared MLP, max over K
    (models/layers.py:375-440)."""

    def __init__(self, in_channels, out_channels_list_before, out_channels_list_after, activation,
                 normalization, momentum=0.1, bn_momentum_decay_step=None, bn_momentum_decay=1):
        super().__init__()
        kw = dict(kernel_size=1, stride=1, padding=0, bias=True, activation=activation,
                  normalization=normalization, momentum=momentum,
                  bn_momentum_decay_step=bn_momentum_decay_step, bn_momentum_decay=bn_momentum_decay)
        self.layers_before = nn.ModuleList()
        prev = in_channels
        for c_out in out_channels_list_before:
            self.layers_before.append(MyConv2d(prev, c_out, **kw))
            prev = c_out
        self.layers_after = nn.ModuleList()
        prev = 2 * prev
        for c_out in out_channels_list_after:
            self.layers_after.append(MyConv2d(prev, c_out, **kw))
            prev = c_out

    def forward(self, query, database, x, K, epoch=None):
        """query Bx3xM, database Bx3xN, x BxCxN -> BxC'xM.  Coordinates carry no gradient."""
        knn_I = Fh.knn_indices(query, database, K)                       # layers.py:417-421
        self.last_knn_I = knn_I
        first0 = self.layers_before[0]
        bn0 = getattr(first0, "norm", None)
        if (not is_generic(first0) and isinstance(bn0, _BatchNorm)
                and Fh.knn_first_layer_supported(x, knn_I, first0.conv.bias, bn0, first0.activation == "relu")):
            # :422-431 without the gathered tensor (csrc/knn_layer.hip): the feature half of the first layer's product is
            # taken over the database points and gathered
            bn0.decay_momentum(epoch)
            h = Fh.knn_first_layer(x, database, query, knn_I, first0.conv.weight, first0.conv.bias, bn0)
            rest_before = list(self.layers_before)[1:]
        else:
            h = Fh.knn_group(x, database.detach(), query.detach(), knn_I)    # :422-430
            rest_before = list(self.layers_before)
        for layer in rest_before:
            # the three decentered coordinates in front of the features are inputs without a gradient
            h = layer(h, epoch, defer=True, nograd_prefix=3 if layer is first0 else 0)
        pooled, h = Fh.group_max_fork(h)                                 # :433 (BN+ReLU+max in one pass)
        first, rest = self.layers_after[0], list(self.layers_after)[1:]
        y = pooled_concat_layer(first, h, pooled, True, epoch)           # :435
        for layer in rest[:-1]:
            y = layer(y, epoch, defer=True)
        if rest and isinstance(getattr(rest[-1], "norm", None), _BatchNorm) and rest[-1].activation == "relu" \
                and not is_generic(rest[-1]):
            last = rest[-1]                                              # last layer + max over K fused
            last.norm.decay_momentum(epoch)
            return Fh.conv1x1_bn_relu_max(y, last.conv.weight, last.conv.bias, last.norm)   # :436-438
        for layer in rest[-1:]:
            y = layer(y, epoch, defer=True)
        return Fh.group_max(y)                                           # :438
