"""Synthetic inputs for the BASELINE.json configurations.

Pure numpy, deterministic in the seed, no file or network access.  Used by bench.py, the tests
and __graft_entry__.smoke() so that every leg of a comparison sees the same atoms.
"""
import math

import numpy as np

# ---------------------------------------------------------------------------------------------
# ANI-2x symmetry-function constants.
# Source of the numbers: the reference's own benchmark program, which embeds the ANI-2x set
# (src/ani/BenchmarkCudaANISymmetryFunctions.cu:101-153): 7 species (H C N O S F Cl),
# Rcr 5.1, Rca 3.5, 16 radial shifts eta 19.7, 8 x 4 angular (eta 12.5, zeta 14.1).
# Function order follows the torch binding's loop nest (src/pytorch/SymmetryFunctions.cpp:110-120):
# radial  = for eta in EtaR: for rs in ShfR
# angular = for eta in EtaA: for zeta in Zeta: for rs in ShfA: for thetas in ShfZ
# ---------------------------------------------------------------------------------------------
ANI2X = dict(
    num_species=7, Rcr=5.1, Rca=3.5,
    EtaR=[19.7], ShfR=[0.8 + 0.26875 * i for i in range(16)],
    EtaA=[12.5], Zeta=[14.1], ShfA=[0.8 + 0.3375 * i for i in range(8)],
    ShfZ=[(2 * i + 1) * math.pi / 8 for i in range(4)],
)


def expand_functions(EtaR, ShfR, EtaA, Zeta, ShfA, ShfZ, **_):
    """-> (radial_functions [nR,2], angular_functions [nA,4]) in the binding's loop order."""
    rf = np.array([[eta, rs] for eta in EtaR for rs in ShfR], dtype=np.float32)
    af = np.array([[eta, rs, zeta, ths] for eta in EtaA for zeta in Zeta for rs in ShfA for ths in ShfZ],
                  dtype=np.float32)
    return rf, af


def ani2x_functions():
    return expand_functions(**ANI2X)


# ---------------------------------------------------------------------------------------------
# geometry generators
# ---------------------------------------------------------------------------------------------
def _lattice_gas(n, box_len, min_dist, rng, periodic=True):
    """n points in a cube with a minimum separation: jittered simple-cubic sites (dense, fast,
    no rejection loop), which is what a liquid at 0.1 atoms/A^3 looks like at this level."""
    m = int(math.ceil(n ** (1.0 / 3.0)))
    while m ** 3 < n:
        m += 1
    a = box_len / m
    jitter = max(0.0, 0.5 * (a - min_dist))
    idx = rng.permutation(m ** 3)[:n]
    ijk = np.stack([idx // (m * m), (idx // m) % m, idx % m], axis=1).astype(np.float64)
    pos = (ijk + 0.5) * a + rng.uniform(-jitter, jitter, size=(n, 3))
    return pos.astype(np.float32)


def random_box(n_atoms, density=0.1, seed=0, min_dist=0.8, n_species=7, species_probs=None):
    """Periodic cubic box at the given number density (atoms / A^3).
    -> positions [N,3] f32, species [N] i32, box [3,3] f32."""
    rng = np.random.default_rng(seed)
    box_len = (n_atoms / density) ** (1.0 / 3.0)
    pos = _lattice_gas(n_atoms, box_len, min_dist, rng)
    if species_probs is None:
        species = rng.integers(0, n_species, size=n_atoms)
    else:
        species = rng.choice(len(species_probs), size=n_atoms, p=species_probs)
    box = np.eye(3, dtype=np.float32) * np.float32(box_len)
    return pos, species.astype(np.int32), box


def water_box(n_waters, density=0.1, seed=1):
    """Periodic cubic box of rigid waters (O-H 0.96 A, 104.5 deg, random orientation), atoms
    ordered O,H,H per molecule; ANI-2x species order H,C,N,O,... so O = 3, H = 0.
    -> positions [3*n_waters,3], species, box."""
    rng = np.random.default_rng(seed)
    n_atoms = 3 * n_waters
    box_len = (n_atoms / density) ** (1.0 / 3.0)
    oxy = _lattice_gas(n_waters, box_len, 2.4, rng).astype(np.float64)
    # random rotations from normalised quaternions
    q = rng.normal(size=(n_waters, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], axis=1),
                  np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], axis=1),
                  np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1)], axis=1)
    half = math.radians(104.5) / 2
    h1 = 0.96 * np.array([math.sin(half), 0.0, math.cos(half)])
    h2 = 0.96 * np.array([-math.sin(half), 0.0, math.cos(half)])
    pos = np.empty((n_waters, 3, 3))
    pos[:, 0] = oxy
    pos[:, 1] = oxy + R @ h1
    pos[:, 2] = oxy + R @ h2
    species = np.tile(np.array([3, 0, 0], dtype=np.int32), n_waters)
    box = np.eye(3, dtype=np.float32) * np.float32(box_len)
    return pos.reshape(-1, 3).astype(np.float32), species, box


def conformer(n_atoms, seed=0):
    """A compact organic-molecule-like cluster in vacuum: self-avoiding random walk with
    1.0-1.5 A bonds, minimum pair distance 0.9 A, species weighted towards H/C/N/O.
    -> positions [n,3], species [n]."""
    rng = np.random.default_rng(seed)
    pts = [np.zeros(3)]
    while len(pts) < n_atoms:
        anchor = pts[rng.integers(0, len(pts))]
        step = rng.normal(size=3)
        step *= rng.uniform(1.0, 1.5) / np.linalg.norm(step)
        cand = anchor + step
        if np.min(np.linalg.norm(np.array(pts) - cand, axis=1)) >= 0.9 and np.linalg.norm(cand) < 10.0:
            pts.append(cand)
    species = rng.choice(4, size=n_atoms, p=[0.5, 0.3, 0.1, 0.1]).astype(np.int32)   # H C N O
    return np.array(pts, dtype=np.float32), species


def triclinic_box(n_atoms, seed=0, density=0.1, n_species=7):
    """Reduced-form triclinic cell (a along x, b in xy) around the same lattice gas; used for the
    triclinic parity cases.  Box rows are the vectors, as the reference expects."""
    pos, species, box = random_box(n_atoms, density, seed, n_species=n_species)
    L = float(box[0, 0])
    box = np.array([[L, 0, 0], [0.15 * L, L, 0], [-0.05 * L, -0.1 * L, L]], dtype=np.float32)
    return pos, species, box


# ---------------------------------------------------------------------------------------------
# A stand-in for a TorchANI ANI-2x model object (torchani itself is not available offline): the same
# attributes NNPOps reads from it (reference src/pytorch/SymmetryFunctions.py:75-86, BatchedNN.py:55-59,
# EnergyShifter.py:34-45), random weights of the ANI-2x layer widths.
# ---------------------------------------------------------------------------------------------
Z_OF_SPECIES = (1, 6, 7, 8, 16, 9, 17)          # H C N O S F Cl, the ANI-2x species order
ANI2X_WIDTHS = {"H": (256, 192, 160), "C": (224, 192, 160), "N": (192, 160, 128), "O": (192, 160, 128),
                "S": (160, 128, 96), "F": (160, 128, 96), "Cl": (160, 128, 96)}


def torchani_like_model(n_models=8, seed=2, self_energies=None):
    """-> object with .species_converter, .aev_computer, .neural_networks (ModuleList of ModuleDicts of
    Sequential(Linear, CELU, ... Linear)), .energy_shifter, shaped like torchani.models.ANI2x."""
    import torch
    from torch import nn
    from types import SimpleNamespace

    class SpeciesConverter(nn.Module):
        def __init__(self):
            super().__init__()
            conv = torch.full((120,), -1, dtype=torch.long)
            for s, z in enumerate(Z_OF_SPECIES):
                conv[z] = s
            self.register_buffer("conv_tensor", conv)

        def forward(self, inp):
            numbers, coords = inp
            return SimpleNamespace(species=self.conv_tensor.to(numbers.device)[numbers], coordinates=coords)

    class ANIModel(nn.ModuleDict):
        pass

    c, t = ANI2X, torch.tensor
    aev_computer = SimpleNamespace(num_species=7, Rcr=c["Rcr"], Rca=c["Rca"],
                                   EtaR=t(c["EtaR"]).view(-1, 1), ShfR=t(c["ShfR"]).view(1, -1),
                                   EtaA=t(c["EtaA"]).view(-1, 1, 1, 1), Zeta=t(c["Zeta"]).view(1, -1, 1, 1),
                                   ShfA=t(c["ShfA"]).view(1, 1, -1, 1), ShfZ=t(c["ShfZ"]).view(1, 1, 1, -1))
    gen = torch.Generator().manual_seed(seed)
    models = []
    for _ in range(n_models):
        nets = {}
        for sym, (h1, h2, h3) in ANI2X_WIDTHS.items():
            net = nn.Sequential(nn.Linear(1008, h1), nn.CELU(0.1), nn.Linear(h1, h2), nn.CELU(0.1), nn.Linear(h2, h3),
                                nn.CELU(0.1), nn.Linear(h3, 1))
            for layer in net:
                if isinstance(layer, nn.Linear):                 # N(0, 1/sqrt(fan_in))
                    layer.weight.data = torch.randn(layer.weight.shape, generator=gen) / math.sqrt(layer.in_features)
                    layer.bias.data = 0.1 * torch.randn(layer.bias.shape, generator=gen)
            nets[sym] = net
        models.append(ANIModel(nets))
    sae = torch.zeros(7, dtype=torch.float64) if self_energies is None else torch.as_tensor(self_energies, dtype=torch.float64)
    shifter = SimpleNamespace(sae=lambda species: sae.to(species.device)[species].sum(dim=1))
    return SimpleNamespace(species_converter=SpeciesConverter(), aev_computer=aev_computer,
                           neural_networks=nn.ModuleList(models), energy_shifter=shifter)
