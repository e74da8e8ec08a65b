"""CFConvNeighbors -- neighbour list of the continuous-filter convolution
(reference src/pytorch/CFConvNeighbors.py:27-45)."""
import torch
from torch import Tensor

from . import torch_binding

torch_binding.load()


class CFConvNeighbors(torch.nn.Module):

    def __init__(self, cutoff: float) -> None:
        super().__init__()
        self.holder = torch.classes.NNPOpsCFConvNeighbors.Holder(cutoff)

    @torch.jit.export
    def build(self, positions: Tensor) -> None:
        self.holder.build(positions)
