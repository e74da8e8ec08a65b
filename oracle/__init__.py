"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms for the hot path (ANI symmetry functions,
SchNet CFConv + half neighbour list, getNeighborPairs), plus -- where ``oracle/_ref`` has been
built -- a doorway onto the reference's own CPU sources compiled in place.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / the timed baseline: never as the thing measured or
shipped.  ``nnpops_amd`` must not import it (tests/test_layout.py enforces that).

    from oracle import AniOracle, CFConvOracle, CFConvNeighborsOracle, neighbor_pairs_oracle
    from oracle import have_ref, RefAni, RefCFConv, RefCFConvNeighbors
"""
from .bindings import (AniOracle, AniOracle64, CFConvNeighborsOracle, CFConvOracle, RefAni, RefCFConv,  # noqa: F401
                       RefCFConvNeighbors, build_oracle, have_ref, oracle_lib_path, ref_lib_path)
from .pme_oracle import pme_direct_oracle  # noqa: F401
from .neighbors_oracle import neighbor_pairs_oracle, neighbor_pairs_backward_oracle  # noqa: F401
