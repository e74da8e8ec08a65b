"""ctypes view of ``libnnpops_hip.so`` -- the C ABI declared in ``include/nnpops_hip.h``.

This is the thinnest possible host layer: it loads the in-tree library, declares every exported
symbol's signature, and offers small classes that hand PyTorch device tensors (used purely as
device-memory owners) to the C ABI by raw pointer.  There is no CPU fallback: if the library is
missing or there is no HIP device the calls raise.

Reference interfaces mirrored (file:line relative to /root/reference/src):
    AniSymmetryFunctions   -> ani/ANISymmetryFunctions.h:41-154   (computeSymmetryFunctions / backprop)
    CFConvNeighbors, CFConv -> schnet/CFConv.h:37-217             (build / compute / backprop)
    neighbor_pairs          -> pytorch/neighbors/getNeighborPairsCUDA.cu:103-196
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnnpops_hip.so")

OK = 0
ERR_CAPACITY = -4

# name -> (restype, argtypes); the single source the symbol-export test checks against the header
SIGNATURES = {
    "nnpops_last_error": (C.c_char_p, []),
    "nnpops_version": (C.c_char_p, []),
    "nnpops_device_count": (C.c_int, []),
    "nnpops_ani_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
    "nnpops_ani_destroy": (C.c_int, [C.c_void_p]),
    "nnpops_ani_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nnpops_ani_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_ani_backprop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_ani_compute_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "nnpops_ani_backprop_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "nnpops_ani_check": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nnpops_ani_check_begin": (C.c_int, [C.c_void_p]),
    "nnpops_ani_check_end": (C.c_int, [C.c_void_p]),
    "nnpops_ani_set_neighbor_algorithm": (C.c_int, [C.c_void_p, C.c_int]),
    "nnpops_ani_set_molecules": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "nnpops_ani_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "nnpops_ani_set_timing_stride": (C.c_int, [C.c_void_p, C.c_int]),
    "nnpops_ani_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "nnpops_ani_timing_overhead": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "nnpops_ani_overflow_word": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "nnpops_ani_read_overflow": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "nnpops_ani_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "nnpops_ani_check_begin_with": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "nnpops_ani_set_timing_merge": (C.c_int, [C.c_void_p, C.c_int]),
    "nnpops_cfconv_neighbors_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_float, C.c_int, C.c_int]),
    "nnpops_cfconv_neighbors_destroy": (C.c_int, [C.c_void_p]),
    "nnpops_cfconv_neighbors_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nnpops_cfconv_neighbors_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_cfconv_neighbors_check": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "nnpops_cfconv_neighbors_export": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nnpops_cfconv_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "nnpops_cfconv_destroy": (C.c_int, [C.c_void_p]),
    "nnpops_cfconv_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nnpops_cfconv_compute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_cfconv_backprop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "nnpops_split_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_long]),
    "nnpops_rows_dot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "nnpops_gemm_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p,
                                    C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long,
                                    C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "nnpops_mlp_packed_halves": (C.c_int64, [C.c_int, C.c_int]),
    "nnpops_mlp_d1_halves": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "nnpops_mlp_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "nnpops_mlp_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "nnpops_mlp_input_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nnpops_mlp_energy_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "nnpops_mlp_energy_mean_shifted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "nnpops_scale_by_scalar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "nnpops_neighbor_pairs_workspace_bytes": (C.c_int64, [C.c_int]),
    "nnpops_neighbor_pairs_forward": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int64,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_pme_direct_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "nnpops_pme_direct": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_pme_direct_indexed_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "nnpops_pme_direct_indexed": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_neighbor_pairs_backward": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_neighbor_pairs_backward_workspace_bytes": (C.c_int64, [C.c_int]),
    "nnpops_neighbor_pairs_index_ints": (C.c_int64, [C.c_int, C.c_int64]),
    "nnpops_neighbor_pairs_index_workspace_bytes": (C.c_int64, [C.c_int, C.c_int64]),
    "nnpops_neighbor_pairs_build_index": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_neighbor_pairs_backward_indexed_workspace_bytes": (C.c_int64, [C.c_int, C.c_int64]),
    "nnpops_neighbor_pairs_backward_indexed": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nnpops_neighbor_pairs_backward_ws": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class NNPOpsHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[nnpops_hip {code}] {message}")
        self.code = code


def lib():
    """Load libnnpops_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m nnpops_amd.build` "
                          "(hipcc --offload-arch=gfx950). nnpops_amd has no CPU fallback.")
    handle = C.CDLL(LIB_PATH)
    missing = [name for name in SIGNATURES if not hasattr(handle, name)]
    if missing:                             # header/library mismatch: fail loudly
        raise ImportError(f"{LIB_PATH} does not export {missing}; rebuild with `python -m nnpops_amd.build --force`")
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = restype
        fn.argtypes = argtypes
    # the binary must have been built from the sources next to it (build.py embeds their hash in nnpops_version())
    from . import build as _build
    have, want = handle.nnpops_version().decode(), _build.source_hash()
    if not have.endswith("src:" + want):
        raise ImportError(f"{LIB_PATH} is stale: it reports '{have}' but the sources hash to {want}; "
                          "rebuild with `python -m nnpops_amd.build`")
    _lib = handle
    return _lib


def _check(code):
    if code != OK:
        raise NNPOpsHipError(code, lib().nnpops_last_error().decode())


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev_f32(t, name, shape=None):
    if not t.is_cuda:
        raise ValueError(f'"{name}" must live on the HIP device (nnpops_amd has no CPU path)')
    if t.dtype != torch.float32:
        raise ValueError(f'"{name}" must be float32')
    if not t.is_contiguous():
        raise ValueError(f'"{name}" must be contiguous')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'"{name}" has shape {tuple(t.shape)}, expected {tuple(shape)}')
    return t


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class AniSymmetryFunctions:
    """One molecule / frame's ANI symmetry-function evaluator on one GPU.

    Mirrors the reference object: construct once with the frozen parameters, ``compute`` then
    ``backprop`` (which relies on state left by the last ``compute``)."""

    def __init__(self, num_species, radial_cutoff, angular_cutoff, atom_species, radial_functions, angular_functions,
                 periodic=False, torchani=True, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        species = np.ascontiguousarray(atom_species, dtype=np.int32)
        rf = np.ascontiguousarray(radial_functions, dtype=np.float32).reshape(-1, 2)
        af = np.ascontiguousarray(angular_functions, dtype=np.float32).reshape(-1, 4)
        self.num_atoms, self.num_species = int(species.shape[0]), int(num_species)
        self.num_radial, self.num_angular = int(rf.shape[0]), int(af.shape[0])
        self.periodic = bool(periodic)
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        _check(self._lib.nnpops_ani_create(C.byref(self._h), self.num_atoms, self.num_species, radial_cutoff,
                                           angular_cutoff, int(self.periodic), species.ctypes.data_as(C.c_void_p),
                                           self.num_radial, rf.ctypes.data_as(C.c_void_p), self.num_angular,
                                           af.ctypes.data_as(C.c_void_p), int(bool(torchani)), self.device.index))
        self.radial_width = self.num_species * self.num_radial
        self.angular_width = self.num_species * (self.num_species + 1) // 2 * self.num_angular

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.nnpops_ani_destroy(h)
            self._h = None

    def set_molecules(self, molecule_offsets):
        """Treat the handle's atoms as a batch of independent non-periodic molecules:
        atoms [offsets[m], offsets[m+1]) belong to molecule m (None restores one system)."""
        if molecule_offsets is None:
            _check(self._lib.nnpops_ani_set_molecules(self._h, 0, None))
            return
        off = np.ascontiguousarray(molecule_offsets, dtype=np.int32)
        _check(self._lib.nnpops_ani_set_molecules(self._h, int(off.shape[0]) - 1, off.ctypes.data_as(C.c_void_p)))

    def set_neighbor_algorithm(self, algorithm):
        _check(self._lib.nnpops_ani_set_neighbor_algorithm(self._h, int(algorithm)))

    def compute(self, positions, box=None, radial=None, angular=None, check=True):
        """positions [N,3] (device fp32), box [3,3] or None -> (radial [N,S*nR], angular [N,NB*nA]).

        ``check=True`` blocks once on the stream to see whether a neighbour row overflowed and, if
        so, repeats the computation with the grown buffers (the C ABI reports, never truncates)."""
        _dev_f32(positions, "positions", (self.num_atoms, 3))
        if self.periodic:
            if box is None:
                raise ValueError("periodic evaluator needs box vectors")
            _dev_f32(box, "box", (3, 3))
        if radial is None:
            radial = torch.empty((self.num_atoms, self.radial_width), dtype=torch.float32, device=positions.device)
        if angular is None:
            angular = torch.empty((self.num_atoms, self.angular_width), dtype=torch.float32, device=positions.device)
        _check(self._lib.nnpops_ani_set_stream(self._h, _stream_ptr(positions.device)))
        for _ in range(8):
            _check(self._lib.nnpops_ani_compute(self._h, _ptr(positions), _ptr(box if self.periodic else None),
                                                _ptr(radial), _ptr(angular)))
            if not check:
                break
            code = self._lib.nnpops_ani_check(self._h, None, None)
            if code == OK:
                break
            if code != ERR_CAPACITY:
                _check(code)
        else:
            raise NNPOpsHipError(ERR_CAPACITY, "neighbour buffers kept overflowing")
        return radial, angular

    def backprop(self, radial_grad, angular_grad, position_grad=None):
        _dev_f32(radial_grad, "radial_grad", (self.num_atoms, self.radial_width))
        _dev_f32(angular_grad, "angular_grad", (self.num_atoms, self.angular_width))
        if position_grad is None:
            position_grad = torch.empty((self.num_atoms, 3), dtype=torch.float32, device=radial_grad.device)
        _check(self._lib.nnpops_ani_set_stream(self._h, _stream_ptr(radial_grad.device)))
        _check(self._lib.nnpops_ani_backprop(self._h, _ptr(radial_grad), _ptr(angular_grad), _ptr(position_grad)))
        return position_grad

    KERNELS = ("neighbors", "radial_forward", "angular_forward", "radial_backward", "angular_backward", "cell_grid")

    def enable_timing(self, enable=True, only=None, every=1):
        """HIP-event timing of the kernels: all of them, or just the names in ``only``, on every ``every``-th launch
        (each event costs a few microseconds of stream time, so benchmarks time one kernel on a sample of the steps
        inside their timed region)."""
        mask = int(bool(enable))
        if enable and only:
            mask = sum(1 << (self.KERNELS.index(k) + 1) for k in only)
        _check(self._lib.nnpops_ani_set_timing_stride(self._h, int(every)))
        _check(self._lib.nnpops_ani_enable_timing(self._h, mask))

    def describe(self):
        """-> {key: value} of nnpops_ani_describe: which kernels this handle runs (forward, backward, uniform, grid, literal, ...)."""
        buf = C.create_string_buffer(512)
        _check(self._lib.nnpops_ani_describe(self._h, buf, 512))
        return dict(word.split("=", 1) for word in buf.value.decode().split())

    def overflow_word(self):
        """The builders' sticky overflow word (bit 0 rows, 1 box too small for the grid, 2 cell bins, 3 an atom outgrew its backward
        class), read from the device (blocks on the handle's stream's device)."""
        word = C.c_int32(0)
        torch.cuda.synchronize()
        _check(self._lib.nnpops_ani_read_overflow(self._h, C.byref(word)))     # (copied by the library's own runtime: ADVICE r05)
        return word.value

    def set_timing_merge(self, merge):
        """True: one bracket around build + angular forward (reported as "neighbors") and one around the two backward kernels (reported as
        "angular_backward") instead of the four single ones: single + single - merged = what a bracket costs, measured in place."""
        _check(self._lib.nnpops_ani_set_timing_merge(self._h, int(bool(merge))))

    def get_timing(self):
        """-> {kernel: (total_ms, launches)} since the last call; blocks on the stream."""
        ms = (C.c_double * len(self.KERNELS))()
        cnt = (C.c_int * len(self.KERNELS))()
        _check(self._lib.nnpops_ani_get_timing(self._h, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.KERNELS)}

    def timing_overhead(self):
        """Seconds an event pair reports for an empty bracket on this handle's stream (blocks)."""
        ms = C.c_double(0)
        _check(self._lib.nnpops_ani_timing_overhead(self._h, C.byref(ms)))
        return 1e-3 * ms.value

    def check_begin(self):
        """First half of the deferred capacity check (right after compute(check=False)): True when the check is in flight and
        check_end() will finish it, False when it cannot be deferred (call neighbor_stats() / compute(check=True) instead)."""
        return self._lib.nnpops_ani_check_begin(self._h) == 1

    def check_end(self):
        """Second half: OK, or ERR_CAPACITY after the buffers have grown (compute() and its consumers must be issued again)."""
        code = self._lib.nnpops_ani_check_end(self._h)
        if code not in (OK, ERR_CAPACITY):
            _check(code)
        return code

    def neighbor_stats(self):
        """(max neighbours within Rcr, max neighbours within Rca) of the last compute; blocks."""
        a, b = C.c_int(0), C.c_int(0)
        code = self._lib.nnpops_ani_check(self._h, C.byref(a), C.byref(b))
        if code not in (OK, ERR_CAPACITY):
            _check(code)
        return a.value, b.value


# ---------------------------------------------------------------------------------------------
# getNeighborPairs (reference src/pytorch/neighbors/getNeighborPairsCUDA.cu)
# ---------------------------------------------------------------------------------------------
_DTYPE_CODE = {torch.float32: 0, torch.float64: 1}


def neighbor_pairs_forward(positions, cutoff, max_num_pairs=-1, box=None):
    """-> (neighbors int32[2,P], deltas[P,3], distances[P], num_pairs int32[1]) on positions' device.

    P = N(N-1)/2 when max_num_pairs == -1, else max_num_pairs.  No host synchronisation."""
    if not positions.is_cuda:
        raise ValueError('"positions" must live on the HIP device (nnpops_amd has no CPU path)')
    if positions.dtype not in _DTYPE_CODE:
        raise ValueError('"positions" must be float32 or float64')
    if positions.dim() != 2 or positions.size(1) != 3 or not positions.is_contiguous():
        raise ValueError('Expected "positions" to be a contiguous (num_atoms, 3) tensor')
    n = positions.size(0)
    max_num_pairs = int(max_num_pairs)
    if not (max_num_pairs > 0 or max_num_pairs == -1):
        raise ValueError('Expected "max_num_pairs" to be positive or equal to -1')
    slots = n * (n - 1) // 2 if max_num_pairs == -1 else max_num_pairs
    dev, dt = positions.device, positions.dtype
    if box is not None and box.numel():
        if tuple(box.shape) != (3, 3):
            raise ValueError('Expected "box_vectors" to have shape (3, 3)')
        box = box.to(device=dev, dtype=dt).contiguous()
    else:
        box = None
    neighbors = torch.empty((2, slots), dtype=torch.int32, device=dev)
    deltas = torch.empty((slots, 3), dtype=dt, device=dev)
    distances = torch.empty((slots,), dtype=dt, device=dev)
    num_pairs = torch.empty((1,), dtype=torch.int32, device=dev)
    L = lib()
    ws = torch.empty((int(L.nnpops_neighbor_pairs_workspace_bytes(n)),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(L.nnpops_neighbor_pairs_forward(_DTYPE_CODE[dt], n, _ptr(positions), _ptr(box), float(cutoff), max_num_pairs,
                                               _ptr(neighbors), _ptr(deltas), _ptr(distances), _ptr(num_pairs), _ptr(ws),
                                               _stream_ptr(dev)))
    return neighbors, deltas, distances, num_pairs


def neighbor_pairs_backward(num_atoms, neighbors, deltas, distances, grad_deltas, grad_distances):
    dev, dt = deltas.device, deltas.dtype
    grad_positions = torch.empty((num_atoms, 3), dtype=dt, device=dev)
    with torch.cuda.device(dev):
        ws = torch.empty((int(lib().nnpops_neighbor_pairs_backward_workspace_bytes(num_atoms)) // 8,), dtype=torch.int64, device=dev)
        _check(lib().nnpops_neighbor_pairs_backward_ws(_DTYPE_CODE[dt], num_atoms, distances.numel(), _ptr(neighbors),
                                                       _ptr(deltas.contiguous()), _ptr(distances.contiguous()),
                                                       _ptr(grad_deltas.contiguous()), _ptr(grad_distances.contiguous()),
                                                       _ptr(grad_positions), _ptr(ws), _stream_ptr(dev)))
    return grad_positions


def neighbor_pairs_build_index(num_atoms, neighbors):
    """Transposed index of a list the forward op emitted (grouped by neighbors[0]): int32 tensor for neighbor_pairs_backward_indexed."""
    dev = neighbors.device
    L = lib()
    slots = neighbors.size(1)
    index = torch.empty((int(L.nnpops_neighbor_pairs_index_ints(num_atoms, slots)),), dtype=torch.int32, device=dev)
    ws = torch.empty((int(L.nnpops_neighbor_pairs_index_workspace_bytes(num_atoms, slots)) // 8 + 1,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _check(L.nnpops_neighbor_pairs_build_index(num_atoms, slots, _ptr(neighbors), _ptr(index), _ptr(ws), _stream_ptr(dev)))
    return index


def neighbor_pairs_backward_indexed(num_atoms, neighbors, deltas, distances, grad_deltas, grad_distances, index):
    """The backward pass without atomics (owner-computes gather over the transposed index)."""
    dev, dt = deltas.device, deltas.dtype
    grad_positions = torch.empty((num_atoms, 3), dtype=dt, device=dev)
    L = lib()
    with torch.cuda.device(dev):
        ws = torch.empty((int(L.nnpops_neighbor_pairs_backward_indexed_workspace_bytes(_DTYPE_CODE[dt], distances.numel())) // 8 + 1,),
                         dtype=torch.int64, device=dev)
        _check(L.nnpops_neighbor_pairs_backward_indexed(_DTYPE_CODE[dt], num_atoms, distances.numel(), _ptr(neighbors),
                                                        _ptr(deltas.contiguous()), _ptr(distances.contiguous()),
                                                        _ptr(grad_deltas.contiguous()), _ptr(grad_distances.contiguous()),
                                                        _ptr(index), _ptr(grad_positions), _ptr(ws), _stream_ptr(dev)))
    return grad_positions


def pme_direct(positions, charges, neighbors, deltas, distances, exclusions, alpha, coulomb, index=None):
    """Direct-space PME on a pair list (reference src/pytorch/pme/pmeCUDA.cu:30-100) through the C ABI.
    -> (energy float32[1], dE/dpositions [N, 3], dE/dcharges [N]); `exclusions` int32 [N, max], rows sorted descending.
    index: the list's transposed index (neighbor_pairs_build_index) -- the list must then be one the forward op emitted: no atomics."""
    _dev_f32(positions, "positions")
    _dev_f32(charges, "charges")
    n, pairs = positions.size(0), neighbors.size(1)
    dev = positions.device
    exclusions = exclusions.to(device=dev, dtype=torch.int32).contiguous()
    max_excl = exclusions.size(1)
    energy = torch.empty((1,), dtype=torch.float32, device=dev)
    pos_deriv = torch.empty((n, 3), dtype=torch.float32, device=dev)
    charge_deriv = torch.empty((n,), dtype=torch.float32, device=dev)
    L = lib()
    if index is not None:
        ws = torch.empty((int(L.nnpops_pme_direct_indexed_workspace_bytes(pairs, n)),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _check(L.nnpops_pme_direct_indexed(n, pairs, max_excl, _ptr(positions), _ptr(charges), _ptr(neighbors.contiguous()),
                                               _ptr(deltas.contiguous()), _ptr(distances.contiguous()), _ptr(exclusions) if max_excl else None,
                                               _ptr(index), float(alpha), float(coulomb), _ptr(energy), _ptr(pos_deriv), _ptr(charge_deriv),
                                               _ptr(ws), _stream_ptr(dev)))
        return energy, pos_deriv, charge_deriv
    ws = torch.empty((int(L.nnpops_pme_direct_workspace_bytes(pairs, n, max_excl)),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(L.nnpops_pme_direct(n, pairs, max_excl, _ptr(positions), _ptr(charges), _ptr(neighbors.contiguous()),
                                   _ptr(deltas.contiguous()), _ptr(distances.contiguous()), _ptr(exclusions) if max_excl else None,
                                   float(alpha), float(coulomb), _ptr(energy), _ptr(pos_deriv), _ptr(charge_deriv), _ptr(ws),
                                   _stream_ptr(dev)))
    return energy, pos_deriv, charge_deriv


# ---------------------------------------------------------------------------------------------
# SchNet CFConv (reference src/schnet/CFConv.h)
# ---------------------------------------------------------------------------------------------
class CFConvNeighbors:
    """Neighbour list of the continuous-filter convolution (reference CFConvNeighbors, CFConv.h:37-85)."""

    def __init__(self, num_atoms, cutoff, periodic=False, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        self.num_atoms, self.cutoff, self.periodic = int(num_atoms), float(cutoff), bool(periodic)
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        _check(self._lib.nnpops_cfconv_neighbors_create(C.byref(self._h), self.num_atoms, self.cutoff, int(self.periodic),
                                                        self.device.index))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.nnpops_cfconv_neighbors_destroy(h)
            self._h = None

    def build(self, positions, box=None, check=True):
        _dev_f32(positions, "positions", (self.num_atoms, 3))
        if self.periodic:
            if box is None:
                raise ValueError("periodic neighbour list needs box vectors")
            _dev_f32(box, "box", (3, 3))
        _check(self._lib.nnpops_cfconv_neighbors_set_stream(self._h, _stream_ptr(positions.device)))
        for _ in range(8):
            _check(self._lib.nnpops_cfconv_neighbors_build(self._h, _ptr(positions), _ptr(box if self.periodic else None)))
            if not check:
                return
            code = self._lib.nnpops_cfconv_neighbors_check(self._h, None)
            if code == OK:
                return
            if code != ERR_CAPACITY:
                _check(code)
        raise NNPOpsHipError(ERR_CAPACITY, "neighbour buffers kept overflowing")

    def num_pairs(self):
        n = C.c_int(0)
        _check(self._lib.nnpops_cfconv_neighbors_check(self._h, C.byref(n)))
        return n.value

    def export(self):
        """-> (pair_atoms int32[2,P], distances f32[P]) of the half list {(i, j>i)}, i ascending, j ascending (host arrays)."""
        P = self.num_pairs()
        cap = max(P, 1)
        atoms = np.empty((2, cap), np.int32)
        dist = np.empty((cap,), np.float32)
        _check(self._lib.nnpops_cfconv_neighbors_export(self._h, cap, atoms.ctypes.data_as(C.c_void_p),
                                                        dist.ctypes.data_as(C.c_void_p)))
        return atoms[:, :P], dist[:P]


class CFConv:
    """Continuous-filter convolution (reference CFConv, CFConv.h:109-217).  ``w1`` is the core-level
    [width][num_gaussians] array, ``w2`` is [out][in]."""

    def __init__(self, num_atoms, width, num_gaussians, cutoff, gaussian_width, activation, w1, b1, w2, b2,
                 periodic=False, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        act = {"ssp": 0, "tanh": 1, 0: 0, 1: 1}.get(activation)
        if act is None:
            raise ValueError('Invalid value of "activation"')
        self.num_atoms, self.width, self.num_gaussians = int(num_atoms), int(width), int(num_gaussians)
        w1 = np.ascontiguousarray(w1, np.float32).reshape(-1)
        w2 = np.ascontiguousarray(w2, np.float32).reshape(-1)
        b1 = np.ascontiguousarray(b1, np.float32).reshape(-1)
        b2 = np.ascontiguousarray(b2, np.float32).reshape(-1)
        assert w1.size == width * num_gaussians and w2.size == width * width and b1.size == width and b2.size == width
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        _check(self._lib.nnpops_cfconv_create(C.byref(self._h), self.num_atoms, self.width, self.num_gaussians, cutoff,
                                              int(bool(periodic)), gaussian_width, act, w1.ctypes.data_as(C.c_void_p),
                                              b1.ctypes.data_as(C.c_void_p), w2.ctypes.data_as(C.c_void_p),
                                              b2.ctypes.data_as(C.c_void_p), self.device.index))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.nnpops_cfconv_destroy(h)
            self._h = None

    def compute(self, neighbors, positions, x, box=None, out=None):
        _dev_f32(positions, "positions", (self.num_atoms, 3))
        _dev_f32(x, "input", (self.num_atoms, self.width))
        if out is None:
            out = torch.empty_like(x)
        _check(self._lib.nnpops_cfconv_set_stream(self._h, _stream_ptr(x.device)))
        _check(self._lib.nnpops_cfconv_compute(self._h, neighbors._h, _ptr(positions), _ptr(box), _ptr(x), _ptr(out)))
        return out

    def backprop(self, neighbors, positions, x, out_grad, box=None):
        _dev_f32(positions, "positions", (self.num_atoms, 3))
        _dev_f32(x, "input", (self.num_atoms, self.width))
        _dev_f32(out_grad, "output_grad", (self.num_atoms, self.width))
        x_grad = torch.empty_like(x)
        pos_grad = torch.empty((self.num_atoms, 3), dtype=torch.float32, device=x.device)
        _check(self._lib.nnpops_cfconv_set_stream(self._h, _stream_ptr(x.device)))
        _check(self._lib.nnpops_cfconv_backprop(self._h, neighbors._h, _ptr(positions), _ptr(box), _ptr(x), _ptr(out_grad),
                                                _ptr(x_grad), _ptr(pos_grad)))
        return x_grad, pos_grad


# ---- dense layers (batched_nn.hip) ----
def split_planes(w, transpose=False):
    """fp32 matrix [rows][cols] on the device -> (hi, lo) fp16 planes [R][ldp] of it (or of its transpose), ldp = the row
    length rounded up to 32, zero padded; what nnpops_gemm_split takes as its B operand."""
    w = _dev_f32(w, "w")
    src_rows, src_cols = w.shape
    rows, cols = (src_cols, src_rows) if transpose else (src_rows, src_cols)
    ldp = (cols + 31) // 32 * 32
    hi = torch.empty((rows, ldp), dtype=torch.float16, device=w.device)
    lo = torch.empty_like(hi)
    _check(lib().nnpops_split_planes(_stream_ptr(w.device), rows, cols, _ptr(w), src_cols, int(transpose), _ptr(hi), _ptr(lo), ldp))
    return hi, lo


def gemm_split(a, planes, bias=None, celu_of=None, alpha=0.1, a_scale=1.0, out=None, a_rows=None, c_rows=None, rows=None):
    """out[M][N] = a[M][K] @ B, B = the planes of an [N][K] matrix (see split_planes).  ``bias``: add it and apply CELU;
    ``celu_of``: multiply by CELU'(.) of that saved activation instead.  Single problem (the batched / strided form is
    what the torch op uses)."""
    a = _dev_f32(a, "a")
    hi, lo = planes
    m, k = a.shape
    if rows is not None:
        m = int(rows)                                            # (with a_rows: the number of mapped rows)
    n = hi.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    epi = 1 if bias is not None else 2 if celu_of is not None else 0
    _check(lib().nnpops_gemm_split(_stream_ptr(a.device), m, n, k, 1, _ptr(a), k, 0, _ptr(hi), _ptr(lo), hi.shape[1], 0, _ptr(out), n, 0,
                                   epi, _ptr(bias) if bias is not None else None, 0, _ptr(celu_of) if celu_of is not None else None, n, 0,
                                   0, None, 0, 0, None, 0, float(alpha), float(a_scale),
                                   _ptr(a_rows) if a_rows is not None else None, _ptr(c_rows) if c_rows is not None else None))
    return out


# ---- the atomic networks of a frame in two launches (mlp_fused.hip) ----
MLP_MAX_KINDS = 8


class _MlpKind(C.Structure):
    _fields_ = [("num_atoms", C.c_int), ("h1", C.c_int), ("h2", C.c_int), ("h3", C.c_int),
                ("w0", C.c_void_p), ("w2", C.c_void_p), ("w4", C.c_void_p), ("w4t", C.c_void_p), ("w2t", C.c_void_p), ("w0t", C.c_void_p),
                ("b0", C.c_void_p), ("b2", C.c_void_p), ("b4", C.c_void_p), ("w6", C.c_void_p), ("b6", C.c_void_p), ("d1", C.c_void_p),
                ("w0tm", C.c_void_p)]


class _MlpFrame(C.Structure):
    _fields_ = [("num_kinds", C.c_int), ("num_features", C.c_int), ("num_members", C.c_int), ("x", C.c_void_p), ("ldx", C.c_int),
                ("rows", C.c_void_p), ("energies", C.c_void_p), ("alpha", C.c_float), ("dx", C.c_void_p), ("lddx", C.c_int),
                ("upstream", C.c_void_p), ("dx_scale", C.c_float), ("kinds", _MlpKind * MLP_MAX_KINDS),
                ("x_groups", C.c_void_p), ("dead_groups", C.c_void_p), ("num_dead_groups", C.c_int), ("dx_partial", C.c_void_p),
                ("mean_scale", C.c_float), ("mean_out", C.c_void_p), ("mean_shift", C.c_void_p), ("mean_out_shifted", C.c_void_p),
                ("publish_word", C.c_void_p), ("publish_to", C.c_void_p), ("publish_stamp", C.c_int32), ("act_scale_log2", C.c_int)]


def mlp_pack(w, rows, cols, transpose=False, permute=False):
    """fp32 device matrix -> the fragment planes nnpops_mlp_forward / _input_grad take as the LEFT operand of a product with
    `rows` outputs and `cols` inputs (`w` is [rows][cols], or [cols][rows] with transpose)."""
    w = _dev_f32(w, "w")
    out = torch.empty((lib().nnpops_mlp_packed_halves(rows, cols),), dtype=torch.float16, device=w.device)
    _check(lib().nnpops_mlp_pack(_stream_ptr(w.device), rows, cols, _ptr(w), w.shape[1], int(transpose), int(permute), _ptr(out)))
    return out


def _up32(v):
    return (v + 31) // 32 * 32


class FusedMLP:
    """The atomic networks of one frame (reference BatchedNN.py:37-122) on nnpops_mlp_forward / nnpops_mlp_input_grad.
    ``kinds``: one dict per species present, in the order the atoms are grouped: w0 [M,H1,F], b0 [M,H1], w2 [M,H2,H1], b2,
    w4 [M,H3,H2], b4, w6 [M,H3], b6 [M] (float32 device tensors, torch Linear layout) and ``atoms`` (int32 device tensor:
    the rows of x that hold the atoms of the kind).  Widths are zero padded to multiples of 32 here.

    ``live_groups`` (optional): the 16-column blocks of x that can be non-zero (the AEV blocks of the species the molecule
    contains).  The first layer is then packed over those columns only (nnpops_hip.h: x_groups), the others are neither read nor
    multiplied, and their gradient is written as zero; with at most 256 live columns the forward launch also forms the input
    gradient member by member (dx_partial) and input_grad() only adds the members up."""

    def __init__(self, kinds, num_features, alpha=0.1, live_groups=None, act_scale_log2=4):
        if not 1 <= len(kinds) <= MLP_MAX_KINDS:
            raise ValueError(f"1..{MLP_MAX_KINDS} kinds")
        self.x_width = int(num_features)
        self.live = None
        if live_groups is not None:
            assert self.x_width % 16 == 0, "column blocks of 16: the width of x must be a multiple of 16"
            self.live = sorted(int(g) for g in live_groups)
            cols = torch.tensor([16 * g + c for g in self.live for c in range(16)], dtype=torch.long)
            kinds = [dict(kd, w0=kd["w0"][:, :, cols.to(kd["w0"].device)].contiguous()) for kd in kinds]
            num_features = 16 * len(self.live)
        self.F, self.alpha = int(num_features), float(alpha)
        self.M = int(kinds[0]["w0"].shape[0])
        self._keep = []
        self.frame = _MlpFrame()
        self.frame.num_kinds, self.frame.num_features, self.frame.num_members, self.frame.alpha = len(kinds), self.F, self.M, self.alpha
        self.frame.act_scale_log2 = int(act_scale_log2)       # activations are scaled by 2^-k before the fp16 split (nnpops_hip.h)
        dev = kinds[0]["w0"].device
        self.device = dev
        rows = []
        for k, kd in enumerate(kinds):
            M, H1, F = kd["w0"].shape
            H2, H3 = kd["w2"].shape[1], kd["w4"].shape[1]
            assert F == self.F and M == self.M
            h1, h2, h3 = _up32(H1), _up32(H2), _up32(H3)

            def pad(t, *shape):
                out = torch.zeros(shape, dtype=torch.float32, device=dev)
                out[tuple(slice(0, n) for n in t.shape)] = t
                return out.contiguous()
            w0, w2, w4 = pad(kd["w0"], M, h1, F), pad(kd["w2"], M, h2, h1), pad(kd["w4"], M, h3, h2)
            b0, b2, b4, w6 = pad(kd["b0"], M, h1), pad(kd["b2"], M, h2), pad(kd["b4"], M, h3), pad(kd["w6"], M, h3)
            b6 = kd["b6"].to(torch.float32).contiguous()
            planes = {
                "w0": torch.cat([mlp_pack(w0[m], h1, F) for m in range(M)]),
                "w2": torch.cat([mlp_pack(w2[m], h2, h1, permute=True) for m in range(M)]),
                "w4": torch.cat([mlp_pack(w4[m], h3, h2, permute=True) for m in range(M)]),
                "w4t": torch.cat([mlp_pack(w4[m], h2, h3, transpose=True, permute=True) for m in range(M)]),
                "w2t": torch.cat([mlp_pack(w2[m], h1, h2, transpose=True, permute=True) for m in range(M)]),
                "w0t": mlp_pack(w0.reshape(M * h1, F), F, M * h1, transpose=True, permute=True),
            }
            if self.live is not None and F <= 256:
                planes["w0tm"] = torch.cat([mlp_pack(w0[m], F, h1, transpose=True, permute=True) for m in range(M)])
            n = int(kd["atoms"].numel())
            d1 = torch.empty((max(int(lib().nnpops_mlp_d1_halves(n, M, h1)), 1),), dtype=torch.float16, device=dev)
            self._keep += [planes, b0, b2, b4, w6, b6, d1]
            fk = self.frame.kinds[k]
            fk.num_atoms, fk.h1, fk.h2, fk.h3 = n, h1, h2, h3
            for name, t in planes.items():
                setattr(fk, name, t.data_ptr())
            fk.b0, fk.b2, fk.b4, fk.w6, fk.b6, fk.d1 = b0.data_ptr(), b2.data_ptr(), b4.data_ptr(), w6.data_ptr(), b6.data_ptr(), d1.data_ptr()
            rows.append(kd["atoms"].to(torch.int32))
        self.rows = torch.cat(rows).contiguous()
        self.frame.rows = self.rows.data_ptr()
        self.energies = torch.empty((self.rows.numel(), self.M), dtype=torch.float32, device=dev)
        self.frame.energies = self.energies.data_ptr()
        if self.live is not None:
            dead = [g for g in range(self.x_width // 16) if g not in set(self.live)]
            self.x_groups = torch.tensor(self.live, dtype=torch.int32, device=dev)
            self.dead_groups = torch.tensor(dead if dead else [0], dtype=torch.int32, device=dev)
            self.frame.x_groups, self.frame.dead_groups, self.frame.num_dead_groups = self.x_groups.data_ptr(), self.dead_groups.data_ptr(), len(dead)
            if self.F <= 256:
                self.dx_partial = torch.empty((self.M, self.rows.numel(), self.F), dtype=torch.float32, device=dev)
                self.frame.dx_partial = self.dx_partial.data_ptr()

    def forward(self, x, with_gradient=True):
        """x [atoms][>= F] float32 -> energies [grouped atoms][M] (every member's network output per atom)."""
        x = _dev_f32(x, "x")
        self.frame.x, self.frame.ldx = x.data_ptr(), x.shape[1]
        _check(lib().nnpops_mlp_forward(_stream_ptr(x.device), C.byref(self.frame), int(with_gradient)))
        return self.energies

    def energy_mean(self, scale=1.0):
        """scale * sum of the energies of the last forward(): one launch, double accumulation, fixed order."""
        out = torch.empty((1,), dtype=torch.float32, device=self.energies.device)
        _check(lib().nnpops_mlp_energy_mean(_stream_ptr(out.device), _ptr(self.energies), self.energies.numel(), float(scale), _ptr(out)))
        return out

    def energy_mean_shifted(self, shift, scale=1.0):
        """(double)(float)(scale * sum) + shift[0]: the mean of energy_mean() promoted and shifted like the reference's EnergyShifter
        (``energies + self_energies``); ``shift`` a float64 device tensor, the result float64 [1]."""
        assert shift.dtype == torch.float64 and shift.is_cuda and shift.numel() >= 1
        out = torch.empty((1,), dtype=torch.float64, device=self.energies.device)
        _check(lib().nnpops_mlp_energy_mean_shifted(_stream_ptr(out.device), _ptr(self.energies), self.energies.numel(), float(scale),
                                                    _ptr(shift), _ptr(out)))
        return out

    def input_grad(self, like, upstream=None, out=None, scale=1.0):
        """dE/dx of the summed energies of the last forward(with_gradient=True): [atoms][F] float32 (rows of atoms that
        belong to no kind are left as they are)."""
        if out is None:
            out = torch.zeros((like.shape[0], self.x_width), dtype=torch.float32, device=like.device)
        self.frame.dx, self.frame.lddx = out.data_ptr(), out.shape[1]
        self.frame.upstream = upstream.data_ptr() if upstream is not None else None
        self.frame.dx_scale = float(scale)
        _check(lib().nnpops_mlp_input_grad(_stream_ptr(out.device), C.byref(self.frame)))
        return out


def scale_by_scalar(values, factor):
    """values (float32 device tensor) * float(factor[0]), ``factor`` a float32 or float64 DEVICE scalar: one launch
    (nnpops_scale_by_scalar -- the backward of the one-node OptimizedTorchANI step)."""
    values = _dev_f32(values, "values")
    assert factor.is_cuda and factor.dtype in (torch.float32, torch.float64) and factor.numel() >= 1
    out = torch.empty_like(values)
    _check(lib().nnpops_scale_by_scalar(_stream_ptr(values.device), _ptr(values), values.numel(), _ptr(factor),
                                        int(factor.dtype == torch.float64), _ptr(out)))
    return out
