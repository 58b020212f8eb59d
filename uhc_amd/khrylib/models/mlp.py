"""MLP trunk (mirror of uhc/khrylib/models/mlp.py:5-27): Linear + activation per hidden layer, `affine_layers`
parameter names kept so reference checkpoints load."""
import torch
import torch.nn as nn

_ACT = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b whose weight gradient dW = dy^T x -- a GEMM with a small output and K = batch (tens of thousands of
    samples in a full-batch PPO epoch) -- runs as a batched split-K product: rocBLAS gives such shapes a few dozen
    workgroups on 256 CUs (10-40 TFLOP/s at float64, 0.1 for the 1-wide value head); S partial products fill the chip
    (57-75 TFLOP/s, tools/ubench/gemm_splitk.py).  Same mathematics, another summation order."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        n, out, inn = x.shape[0], w.shape[0], w.shape[1]
        gx = gy.mm(w) if ctx.needs_input_grad[0] else None
        s = 32 if out * inn < (1 << 17) else 8
        if n % s == 0 and x.is_contiguous():
            gw = torch.bmm(gy.view(s, n // s, out).transpose(1, 2), x.view(s, n // s, inn)).sum(0)
        else:
            gw = gy.t().mm(x)
        return gx, gw, gy.sum(0)


def linear(layer: nn.Linear, x):
    """nn.Linear forward; large 2-D batches on the GPU take the split-K weight-gradient path."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= 4096 and torch.is_grad_enabled() and layer.weight.requires_grad:
        return _LinearSplitK.apply(x, layer.weight, layer.bias)
    return layer(x)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dims=(128, 128), activation="tanh"):
        super().__init__()
        self.activation = nn.GELU() if activation == "gelu" else _ACT[activation]
        self.out_dim = hidden_dims[-1]
        self.affine_layers = nn.ModuleList()
        last = input_dim
        for nh in hidden_dims:
            self.affine_layers.append(nn.Linear(last, nh))
            last = nh

    def forward(self, x):
        for affine in self.affine_layers:
            x = self.activation(linear(affine, x))
        return x
