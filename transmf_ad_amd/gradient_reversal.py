"""Gradient reversal (reference: models/gradient_reversal/functional.py:4-18,
module.py:5-11): identity in the forward pass, ``-alpha * grad`` in the backward pass.
``alpha`` may be a Python number or a tensor (the reference passes ``torch.Tensor([2])``)."""
import torch
from torch import nn


class _ReverseGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * (-ctx.alpha), None


def revgrad(x, alpha):
    return _ReverseGrad.apply(x, alpha)


class GradientReversal(nn.Module):
    def __init__(self, alpha=1.0):
        super().__init__()
        self.alpha = alpha

    def forward(self, x):
        return revgrad(x, self.alpha)
