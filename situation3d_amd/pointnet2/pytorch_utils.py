"""Shared per-point MLP blocks -- host-side mirror of lib/pointnet2/pytorch_utils.py.

Only what the hot path uses is kept: `SharedMLP` (pytorch_utils.py:11-36), its `Conv2d` /
`Conv1d` building blocks (:67-188), the `BatchNorm*` wrappers (:39-64), `FC` (:225-262) and
`BNMomentumScheduler` (:264-298, used by lib/solver.py:248-257).  Module/child names are the
reference's, so `state_dict()` keys match checkpoint for checkpoint
(`layer{i}.conv.weight`, `layer{i}.bn.bn.{weight,bias,running_mean,running_var,...}`).
"""
import torch
import torch.nn as nn


def _named(block, name, module):
    block.add_module(name, module)


class PointwiseConv2d(nn.Conv2d):
    """nn.Conv2d whose 1x1 / stride-1 case runs as ONE batched GEMM W(Cout,Cin) @ X(B,Cin,P*S)
    (library GEMM, f32) instead of MIOpen's NCHW->NHWC transposes + implicit-GEMM convolution.
    Same parameters, same state_dict keys, same math as the reference's Conv2d 1x1
    (pytorch_utils.py:147-185)."""

    def forward(self, x):
        if self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) \
                and self.groups == 1 and x.dim() == 4:
            b, cin, p, s = x.shape
            # bmm with a stride-0 batch of W: torch.matmul(2-D, 3-D) would fold the batch into
            # rows and materialise TRANSPOSED copies of the whole activation (fwd and bwd)
            w = self.weight.view(1, self.out_channels, cin).expand(b, -1, -1)
            y = torch.bmm(w, x.reshape(b, cin, p * s))
            if self.bias is not None:
                y = y + self.bias.view(1, -1, 1)
            return y.view(b, self.out_channels, p, s)
        return super().forward(x)


class _BNWrap(nn.Sequential):
    """One child called `<name>bn`; weight=1 / bias=0 init (pytorch_utils.py:39-47)."""

    def __init__(self, in_size, bn_cls, name=""):
        super().__init__()
        _named(self, name + "bn", bn_cls(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNWrap):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, nn.BatchNorm1d, name)


class BatchNorm2d(_BNWrap):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, nn.BatchNorm2d, name)


class _ConvBlock(nn.Sequential):
    """conv (+bn) (+activation), or the pre-activation ordering (pytorch_utils.py:67-120):
    conv has a bias only when there is no BN; conv weight init = `init`, bias init = 0."""

    def __init__(self, conv_cls, bn_cls, in_size, out_size, kernel_size, stride, padding,
                 activation, bn, init, bias, preact, name):
        super().__init__()
        bias = bias and (not bn)
        conv_unit = conv_cls(in_size, out_size, kernel_size=kernel_size, stride=stride,
                             padding=padding, bias=bias)
        init(conv_unit.weight)
        if bias:
            nn.init.constant_(conv_unit.bias, 0)
        bn_unit = bn_cls(in_size if preact else out_size) if bn else None
        if preact:
            if bn:
                _named(self, name + "bn", bn_unit)
            if activation is not None:
                _named(self, name + "activation", activation)
        _named(self, name + "conv", conv_unit)
        if not preact:
            if bn:
                _named(self, name + "bn", bn_unit)
            if activation is not None:
                _named(self, name + "activation", activation)


class Conv1d(_ConvBlock):
    def __init__(self, in_size, out_size, *, kernel_size=1, stride=1, padding=0,
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_,
                 bias=True, preact=False, name=""):
        super().__init__(nn.Conv1d, BatchNorm1d, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class Conv2d(_ConvBlock):
    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_,
                 bias=True, preact=False, name=""):
        super().__init__(PointwiseConv2d, BatchNorm2d, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d(+BN)+ReLU applied to (B, C, npoint, nsample) (pytorch_utils.py:11-36)."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False,
                 first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = (not first) or (not preact) or (i != 0)
            _named(self, name + "layer{}".format(i),
                   Conv2d(args[i], args[i + 1], bn=plain and bn,
                          activation=activation if plain else None, preact=preact))


class FC(nn.Sequential):
    """pytorch_utils.py:225-262"""

    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False, init=None,
                 preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        if preact:
            if bn:
                _named(self, name + "bn", BatchNorm1d(in_size))
            if activation is not None:
                _named(self, name + "activation", activation)
        _named(self, name + "fc", fc)
        if not preact:
            if bn:
                _named(self, name + "bn", BatchNorm1d(out_size))
            if activation is not None:
                _named(self, name + "activation", activation)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """pytorch_utils.py:272-298: applies `setter(bn_lambda(epoch))` to every BN layer."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))
