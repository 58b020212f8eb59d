"""Multiplicative-compositional policy of the `uhc_implicit` release config (mirror of uhc/models/policy_mcp.py:9-37).

`num_primitive` MLP actors and a softmax composer over them; the action mean is the composer-weighted sum of the
primitives' means.  Parameter names and shapes equal the reference's (`nets.<i>.0.affine_layers.<l>.*`,
`nets.<i>.1.*`, `composer.0.affine_layers.<l>.*`, `action_log_std`) so released checkpoints load unchanged.

The forward pass is laid out for large on-device batches: the first layer of all primitives is one GEMM against the
row-concatenated weights, deeper layers are one batched GEMM over the primitive axis (instead of 8 x 4 separate
launches), and the composer weights are applied in the same pass."""
import torch
import torch.nn as nn

from ..khrylib.models.mlp import MLP
from ..khrylib.rl.core import DiagGaussian, Policy


class PolicyMCP(Policy):
    def __init__(self, cfg, action_dim, state_dim, net_out_dim=None):
        super().__init__()
        self.type = "gaussian"
        hsize, htype = list(cfg.policy_hsize), cfg.policy_htype
        self.nets = nn.ModuleList()
        for _ in range(cfg.num_primitive):
            head = nn.Linear(hsize[-1], action_dim)
            head.weight.data.mul_(0.1)
            head.bias.data.mul_(0.0)
            self.nets.append(nn.Sequential(MLP(state_dim, hsize, htype), head))
        # the composer MLP applies the activation after its last (num_primitive-wide) layer too, then the softmax
        self.composer = nn.Sequential(MLP(state_dim, list(cfg.get("composer_dim", [300, 200])) + [cfg.num_primitive], htype), nn.Softmax(dim=1))
        self.action_log_std = nn.Parameter(torch.ones(1, action_dim) * cfg.log_std, requires_grad=not cfg.fix_std)

    def primitive_means(self, x):
        """(N, state_dim) -> (N, num_primitive, action_dim), all primitives per layer in one (batched) GEMM."""
        P = len(self.nets)
        act = self.nets[0][0].activation
        n_hidden = len(self.nets[0][0].affine_layers)
        W0 = torch.cat([net[0].affine_layers[0].weight for net in self.nets], 0)       # (P*h0, state_dim)
        b0 = torch.cat([net[0].affine_layers[0].bias for net in self.nets], 0)
        h = act(torch.addmm(b0, x, W0.t())).view(x.shape[0], P, -1).transpose(0, 1)     # (P, N, h0)
        for l in range(1, n_hidden):
            W = torch.stack([net[0].affine_layers[l].weight for net in self.nets])      # (P, h_l, h_{l-1})
            b = torch.stack([net[0].affine_layers[l].bias for net in self.nets])
            h = act(torch.baddbmm(b[:, None, :], h, W.transpose(1, 2)))
        W = torch.stack([net[1].weight for net in self.nets])
        b = torch.stack([net[1].bias for net in self.nets])
        return torch.baddbmm(b[:, None, :], h, W.transpose(1, 2)).transpose(0, 1)       # (N, P, action_dim)

    def forward(self, x):
        x_all = self.primitive_means(x)
        weight = self.composer(x)
        action_mean = torch.sum(weight[:, :, None] * x_all, dim=1)
        action_std = torch.exp(self.action_log_std.expand_as(action_mean))
        return DiagGaussian(action_mean, action_std)

    def get_fim(self, x):
        dist = self.forward(x)
        cov_inv = self.action_log_std.exp().pow(-2).squeeze(0).repeat(x.size(0))
        param_count, std_index, std_id = 0, 0, 0
        for i, (name, param) in enumerate(self.named_parameters()):
            if name == "action_log_std":
                std_id, std_index = i, param_count
            param_count += param.view(-1).shape[0]
        return cov_inv.detach(), dist.loc, {"std_id": std_id, "std_index": std_index}
