"""ctypes bindings for the oracle libraries -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

``liboracle.so``      : this repository's C restatement (oracle/ani_oracle.c, oracle/cfconv_oracle.c).
``libnnpops_ref.so``  : the reference's own CPU sources compiled in place (oracle/ref_shim.cpp);
                        present only where ``make -C oracle ref`` could run (i.e. where
                        /root/reference exists) -- the prebuilt file travels to the GPU box.

Both expose the same call shapes, so every class below takes ``lib`` + a symbol prefix.
All arrays are numpy float32 / int32, C-contiguous, on the host.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def oracle_lib_path():
    return os.path.join(_HERE, "_build", "liboracle.so")


def ref_lib_path():
    return os.path.join(_HERE, "_ref", "libnnpops_ref.so")


def build_oracle(with_ref=True, quiet=True):
    """Compile the checker (and, where /root/reference exists, oracle/_ref).  Building the
    checker is not using it: __graft_entry__.build() calls this."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=out)
    if with_ref and os.path.isdir("/root/reference/src/ani"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=out)
        # the reference's own C++ test suites against the HIP library (needs nnpops_amd/libnnpops_hip.so)
        if os.path.exists(os.path.join(os.path.dirname(_HERE.rstrip("/")), "nnpops_amd", "libnnpops_hip.so")):
            subprocess.check_call(["make", "-C", _HERE, "ref_tests"], stdout=out)


def have_ref():
    return os.path.exists(ref_lib_path())


_libs = {}


def _load(which):
    if which in _libs:
        return _libs[which]
    if which == "oracle":
        path = oracle_lib_path()
        if not os.path.exists(path):
            build_oracle(with_ref=False)
        prefix_ani, prefix_nb, prefix_cf = "ani_oracle_", "cfconv_oracle_neighbors_", "cfconv_oracle_"
    else:
        path = ref_lib_path()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built (needs /root/reference; run `make -C oracle ref`)")
        prefix_ani, prefix_nb, prefix_cf = "ref_ani_", "ref_cfconv_neighbors_", "ref_cfconv_"
    lib = C.CDLL(path)

    def sig(name, restype, argtypes):
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
        return fn

    api = {}
    api["ani_create"] = sig(prefix_ani + "create", C.c_void_p,
                            [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _i32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int])
    api["ani_destroy"] = sig(prefix_ani + "destroy", None, [C.c_void_p])
    api["ani_forward"] = sig(prefix_ani + "forward", None, [C.c_void_p, _f32p, C.c_void_p, _f32p, _f32p])
    api["ani_backward"] = sig(prefix_ani + "backward", None, [C.c_void_p, _f32p, _f32p, _f32p])
    api["nb_create"] = sig(prefix_nb + "create", C.c_void_p, [C.c_int, C.c_float, C.c_int])
    api["nb_destroy"] = sig(prefix_nb + "destroy", None, [C.c_void_p])
    api["nb_build"] = sig(prefix_nb + "build", None, [C.c_void_p, _f32p, C.c_void_p])
    api["nb_num_pairs"] = sig(prefix_nb + "num_pairs", C.c_int, [C.c_void_p])
    if which == "oracle":
        api["nb_start"] = sig(prefix_nb + "start", C.POINTER(C.c_int), [C.c_void_p])
        api["nb_other"] = sig(prefix_nb + "other", C.POINTER(C.c_int), [C.c_void_p])
        api["nb_dist"] = sig(prefix_nb + "dist", C.POINTER(C.c_float), [C.c_void_p])
    else:
        api["nb_export"] = sig(prefix_nb + "export", None, [C.c_void_p, _i32p, _i32p, _f32p])
    api["cf_create"] = sig(prefix_cf + "create", C.c_void_p,
                           [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, _f32p, _f32p, _f32p, _f32p])
    api["cf_destroy"] = sig(prefix_cf + "destroy", None, [C.c_void_p])
    api["cf_forward"] = sig(prefix_cf + "forward", None, [C.c_void_p, C.c_void_p, _f32p, C.c_void_p, _f32p, _f32p])
    api["cf_backward"] = sig(prefix_cf + "backward", None,
                             [C.c_void_p, C.c_void_p, _f32p, C.c_void_p, _f32p, _f32p, _f32p, _f32p])
    _libs[which] = api
    return api


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _boxptr(box):
    if box is None:
        return None, None
    b = _f32(box).reshape(3, 3)
    return b, b.ctypes.data_as(C.c_void_p)


class _AniBase:
    """Stateful like the reference: backward() uses what the last forward() left behind
    (reference src/ani/ANISymmetryFunctions.h:83-84)."""
    _which = "oracle"

    def __init__(self, n_species, rc_radial, rc_angular, species, radial_functions, angular_functions,
                 periodic=False, torchani=True):
        self.api = _load(self._which)
        self.species = np.ascontiguousarray(species, dtype=np.int32)
        self.n_atoms = int(self.species.shape[0])
        self.n_species = int(n_species)
        self.rf = _f32(radial_functions).reshape(-1, 2)
        self.af = _f32(angular_functions).reshape(-1, 4)
        self.n_radial, self.n_angular = self.rf.shape[0], self.af.shape[0]
        self.periodic = bool(periodic)
        self.handle = C.c_void_p(self.api["ani_create"](self.n_atoms, self.n_species, rc_radial, rc_angular,
                                                        int(self.periodic), self.species, self.n_radial, self.rf,
                                                        self.n_angular, self.af, int(bool(torchani))))

    def __del__(self):
        if getattr(self, "handle", None):
            self.api["ani_destroy"](self.handle)
            self.handle = None

    def forward(self, positions, box=None):
        pos = _f32(positions).reshape(self.n_atoms, 3)
        nb = self.n_species * (self.n_species + 1) // 2
        radial = np.empty((self.n_atoms, self.n_species * self.n_radial), np.float32)
        angular = np.empty((self.n_atoms, nb * self.n_angular), np.float32)
        keep, bp = _boxptr(box if self.periodic else None)
        self.api["ani_forward"](self.handle, pos, bp, radial, angular)
        return radial, angular

    def backward(self, radial_grad, angular_grad):
        out = np.empty((self.n_atoms, 3), np.float32)
        self.api["ani_backward"](self.handle, _f32(radial_grad), _f32(angular_grad), out)
        return out


class AniOracle(_AniBase):
    _which = "oracle"


class AniOracle64:
    """oracle/ani_oracle.c compiled with -DORACLE_DOUBLE (liboracle64.so): the reference algorithm evaluated in double
    precision on the same float32 inputs and parameters.  Not a parity target -- the yardstick for how far the fp32
    reference itself is from the exact answer (used where the problem is ill conditioned, e.g. paper-mode angle gradients
    of nearly collinear triples).  Same stateful interface as AniOracle; float64 arrays out."""

    def __init__(self, n_species, rc_radial, rc_angular, species, radial_functions, angular_functions,
                 periodic=False, torchani=True):
        path = os.path.join(_HERE, "_build", "liboracle64.so")
        if not os.path.exists(path):
            build_oracle(with_ref=False)
        lib = C.CDLL(path)
        f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        lib.ani_oracle64_create.restype = C.c_void_p
        lib.ani_oracle64_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, _i32p, C.c_int, f64p, C.c_int, f64p, C.c_int]
        lib.ani_oracle64_destroy.argtypes = [C.c_void_p]
        lib.ani_oracle64_forward.argtypes = [C.c_void_p, f64p, C.c_void_p, f64p, f64p]
        lib.ani_oracle64_backward.argtypes = [C.c_void_p, f64p, f64p, f64p]
        self.lib = lib
        self.species = np.ascontiguousarray(species, dtype=np.int32)
        self.n_atoms, self.n_species = int(self.species.shape[0]), int(n_species)
        # float32 parameters, widened exactly
        self.rf = np.ascontiguousarray(_f32(radial_functions).reshape(-1, 2), dtype=np.float64)
        self.af = np.ascontiguousarray(_f32(angular_functions).reshape(-1, 4), dtype=np.float64)
        self.periodic = bool(periodic)
        self.handle = C.c_void_p(lib.ani_oracle64_create(self.n_atoms, self.n_species, float(np.float32(rc_radial)),
                                                         float(np.float32(rc_angular)), int(self.periodic), self.species,
                                                         self.rf.shape[0], self.rf, self.af.shape[0], self.af, int(bool(torchani))))

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.ani_oracle64_destroy(self.handle)
            self.handle = None

    def forward(self, positions, box=None):
        pos = np.ascontiguousarray(_f32(positions).reshape(self.n_atoms, 3), dtype=np.float64)
        nb = self.n_species * (self.n_species + 1) // 2
        radial = np.empty((self.n_atoms, self.n_species * self.rf.shape[0]), np.float64)
        angular = np.empty((self.n_atoms, nb * self.af.shape[0]), np.float64)
        b = np.ascontiguousarray(_f32(box).reshape(3, 3), dtype=np.float64) if (self.periodic and box is not None) else None
        self.lib.ani_oracle64_forward(self.handle, pos, b.ctypes.data_as(C.c_void_p) if b is not None else None, radial, angular)
        return radial, angular

    def backward(self, radial_grad, angular_grad):
        out = np.empty((self.n_atoms, 3), np.float64)
        self.lib.ani_oracle64_backward(self.handle, np.ascontiguousarray(radial_grad, dtype=np.float64),
                                       np.ascontiguousarray(angular_grad, dtype=np.float64), out)
        return out


class RefAni(_AniBase):
    _which = "ref"


class _NeighborsBase:
    _which = "oracle"

    def __init__(self, n_atoms, cutoff, periodic=False):
        self.api = _load(self._which)
        self.n_atoms, self.cutoff, self.periodic = int(n_atoms), float(cutoff), bool(periodic)
        self.handle = C.c_void_p(self.api["nb_create"](self.n_atoms, self.cutoff, int(self.periodic)))

    def __del__(self):
        if getattr(self, "handle", None):
            self.api["nb_destroy"](self.handle)
            self.handle = None

    def build(self, positions, box=None):
        pos = _f32(positions).reshape(self.n_atoms, 3)
        keep, bp = _boxptr(box if self.periodic else None)
        self.api["nb_build"](self.handle, pos, bp)

    def num_pairs(self):
        return int(self.api["nb_num_pairs"](self.handle))

    def export(self):
        """-> (start[n_atoms+1], other[P], dist[P]) of the half list {j>i}."""
        P = self.num_pairs()
        if self._which == "oracle":
            start = np.ctypeslib.as_array(self.api["nb_start"](self.handle), (self.n_atoms + 1,)).copy()
            if P == 0:
                return start, np.zeros(0, np.int32), np.zeros(0, np.float32)
            other = np.ctypeslib.as_array(self.api["nb_other"](self.handle), (P,)).copy()
            dist = np.ctypeslib.as_array(self.api["nb_dist"](self.handle), (P,)).copy()
            return start, other, dist
        start = np.empty(self.n_atoms + 1, np.int32)
        other = np.empty(max(P, 1), np.int32)
        dist = np.empty(max(P, 1), np.float32)
        self.api["nb_export"](self.handle, start, other, dist)
        return start, other[:P], dist[:P]


class CFConvNeighborsOracle(_NeighborsBase):
    _which = "oracle"


class RefCFConvNeighbors(_NeighborsBase):
    _which = "ref"


class _CFConvBase:
    """w1 is the core-level [W][G] array (see DESIGN.md for the binding-level [G,W] reinterpretation)."""
    _which = "oracle"

    def __init__(self, n_atoms, width, n_gauss, cutoff, sigma, activation, w1, b1, w2, b2, periodic=False):
        self.api = _load(self._which)
        self.n_atoms, self.width, self.n_gauss = int(n_atoms), int(width), int(n_gauss)
        self.periodic = bool(periodic)
        act = {"ssp": 0, "tanh": 1, 0: 0, 1: 1}[activation]
        self.w1, self.b1, self.w2, self.b2 = _f32(w1).reshape(-1), _f32(b1), _f32(w2).reshape(-1), _f32(b2)
        assert self.w1.size == width * n_gauss and self.w2.size == width * width
        self.handle = C.c_void_p(self.api["cf_create"](self.n_atoms, self.width, self.n_gauss, cutoff, int(self.periodic),
                                                       sigma, act, self.w1, self.b1, self.w2, self.b2))

    def __del__(self):
        if getattr(self, "handle", None):
            self.api["cf_destroy"](self.handle)
            self.handle = None

    def forward(self, neighbors, positions, x, box=None):
        pos = _f32(positions).reshape(self.n_atoms, 3)
        out = np.empty((self.n_atoms, self.width), np.float32)
        keep, bp = _boxptr(box if self.periodic else None)
        self.api["cf_forward"](self.handle, neighbors.handle, pos, bp, _f32(x), out)
        return out

    def backward(self, neighbors, positions, x, out_grad, box=None):
        pos = _f32(positions).reshape(self.n_atoms, 3)
        xg = np.empty((self.n_atoms, self.width), np.float32)
        pg = np.empty((self.n_atoms, 3), np.float32)
        keep, bp = _boxptr(box if self.periodic else None)
        self.api["cf_backward"](self.handle, neighbors.handle, pos, bp, _f32(x), _f32(out_grad), xg, pg)
        return xg, pg


class CFConvOracle(_CFConvBase):
    _which = "oracle"


class RefCFConv(_CFConvBase):
    _which = "ref"
