"""Neighbour search (reference src/pytorch/neighbors/__init__.py)."""
from .getNeighborPairs import getNeighborPairs  # noqa: F401
