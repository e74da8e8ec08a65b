"""Particle Mesh Ewald, direct-space part (reference src/pytorch/pme/__init__.py)."""
from .pme import PME  # noqa: F401
