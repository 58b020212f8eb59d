"""MLP trunk (mirror of uhc/khrylib/models/mlp.py:5-27): Linear + activation per hidden layer, `affine_layers`
parameter names kept so reference checkpoints load."""
import torch
import torch.nn as nn

_ACT = {"tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}


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
            x = self.activation(affine(x))
        return x
