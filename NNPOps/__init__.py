"""Import-compatible facade: ``import NNPOps`` / ``from NNPOps.SymmetryFunctions import ...`` resolve to the
MI355X implementation in ``nnpops_amd`` (reference package layout: src/pytorch/__init__.py)."""
import importlib
import sys

from nnpops_amd import torch_binding as _binding

_binding.load()      # torch.ops.load_library(libNNPOpsPyTorch.so), as the reference does at import

for _name in ("SymmetryFunctions", "SpeciesConverter", "EnergyShifter", "BatchedNN", "OptimizedTorchANI", "CFConv",
              "CFConvNeighbors", "neighbors", "neighbors.getNeighborPairs", "pme", "pme.pme"):
    sys.modules[f"NNPOps.{_name}"] = importlib.import_module(f"nnpops_amd.{_name}")

from nnpops_amd.OptimizedTorchANI import OptimizedTorchANI  # noqa: E402,F401
