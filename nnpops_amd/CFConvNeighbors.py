"""CFConvNeighbors -- neighbour list of the continuous-filter convolution
(reference src/pytorch/CFConvNeighbors.py:27-45)."""
from typing import Optional

import torch
from torch import Tensor

from . import torch_binding

torch_binding.load()


class CFConvNeighbors(torch.nn.Module):

    def __init__(self, cutoff: float) -> None:
        super().__init__()
        self.holder = torch.classes.NNPOpsCFConvNeighbors.Holder(cutoff)

    @torch.jit.export
    def build(self, positions: Tensor, box: Optional[Tensor] = None) -> None:
        """``box`` (3, 3, rows = lattice vectors in reduced form) is an extension: the reference's Python surface
        is non-periodic although its core supports a box (src/schnet/CFConv.h:57).  A holder is periodic or not
        for life, decided by its first build."""
        if box is None:
            self.holder.build(positions)
        else:
            self.holder.build_periodic(positions, box)
