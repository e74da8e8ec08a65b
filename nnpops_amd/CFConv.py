"""CFConv -- SchNet continuous-filter convolution layer (reference src/pytorch/CFConv.py:29-83).

    neighbors = CFConvNeighbors(cutoff)
    conv = CFConv(gaussianWidth, 'ssp', weights1[G, W], biases1[W], weights2[W, W], biases2[W])
    neighbors.build(positions)
    output = conv(neighbors, positions, input)        # differentiable in positions and input
"""
import torch
from torch import Tensor

from . import torch_binding
from .CFConvNeighbors import CFConvNeighbors

torch_binding.load()


class CFConv(torch.nn.Module):

    def __init__(self, gaussianWidth: float, activation: str, weights1: Tensor, biases1: Tensor, weights2: Tensor,
                 biases2: Tensor) -> None:
        super().__init__()
        self.holder = torch.classes.NNPOpsCFConv.Holder(gaussianWidth, activation, weights1, biases1, weights2, biases2)

    def forward(self, neighbors: CFConvNeighbors, positions: Tensor, input: Tensor) -> Tensor:
        return torch.ops.NNPOpsCFConv.operation(self.holder, neighbors.holder, positions, input)
