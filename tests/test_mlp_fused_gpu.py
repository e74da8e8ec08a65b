"""The fused atomic networks (mlp_fused.hip, C ABI nnpops_mlp_*) against the same networks evaluated in float64 on the host.

The function is the reference's TorchANIBatchedNN (src/pytorch/BatchedNN.py:100-111: four BatchedLinear with CELU(0.1) in
between, summed) and its gradient with respect to the AEV (BatchedNN.cpp:41-47 applied four times by autograd).  Bars:
north_star's 1e-5 on energies; gradients 1e-4 of the largest component (they become forces through the AEV backward).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _networks(widths, members, features, seed, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    h1, h2, h3 = widths
    def rnd(*shape, fan):
        return (scale * torch.randn(shape, generator=gen) / np.sqrt(fan)).float()
    return dict(w0=rnd(members, h1, features, fan=features), b0=0.1 * torch.randn((members, h1), generator=gen),
                w2=rnd(members, h2, h1, fan=h1), b2=0.1 * torch.randn((members, h2), generator=gen),
                w4=rnd(members, h3, h2, fan=h2), b4=0.1 * torch.randn((members, h3), generator=gen),
                w6=rnd(members, h3, fan=h3), b6=0.1 * torch.randn((members,), generator=gen))


def _host_reference(kinds, x):
    """-> per (grouped atom, member) energies [n, M] and dE_total/dx [atoms, F], float64."""
    xd = x.double().requires_grad_(True)
    outs = []
    for kd in kinds:
        xs = xd[kd["atoms"].long()]
        per_member = []
        for m in range(kd["w0"].shape[0]):
            y = torch.nn.functional.celu(xs @ kd["w0"][m].double().t() + kd["b0"][m].double(), alpha=0.1)
            y = torch.nn.functional.celu(y @ kd["w2"][m].double().t() + kd["b2"][m].double(), alpha=0.1)
            y = torch.nn.functional.celu(y @ kd["w4"][m].double().t() + kd["b4"][m].double(), alpha=0.1)
            per_member.append(y @ kd["w6"][m].double() + kd["b6"][m].double())
        outs.append(torch.stack(per_member, dim=1))
    e = torch.cat(outs, dim=0)
    e.sum().backward()
    return e.detach(), xd.grad


def _run(kinds_host, x_host, features, act_scale_log2=4):
    from nnpops_amd.capi import FusedMLP
    kinds_dev = [{k: v.to(DEV) for k, v in kd.items()} for kd in kinds_host]
    mlp = FusedMLP(kinds_dev, features, act_scale_log2=act_scale_log2)
    x = x_host.to(DEV).contiguous()
    e = mlp.forward(x, with_gradient=True).clone()
    dx = mlp.input_grad(x)
    torch.cuda.synchronize()
    e_ref, dx_ref = _host_reference(kinds_host, x_host)
    e, dx = e.cpu().double(), dx.cpu().double()
    assert torch.isfinite(e).all() and torch.isfinite(dx).all()
    assert float((e - e_ref).abs().max()) <= 1e-5 * float(e_ref.abs().max()), float((e - e_ref).abs().max())
    assert abs(float(e.sum() - e_ref.sum())) <= 1e-5 * float(e_ref.abs().sum())
    assert float((dx - dx_ref).abs().max()) <= 1e-4 * float(dx_ref.abs().max()), float((dx - dx_ref).abs().max() / dx_ref.abs().max())
    # energy-only launch: same energies; the mean over members in one launch
    e2 = mlp.forward(x, with_gradient=False).cpu().double()
    assert torch.equal(e2, e)
    mean = float(mlp.energy_mean(1.0 / e.shape[1]).cpu())
    assert abs(mean - float(e_ref.sum()) / e.shape[1]) <= 1e-5 * float(e_ref.abs().sum()) / e.shape[1]
    return e, dx


def test_water_box_shapes_of_config2():
    """BASELINE config 2's networks: 1334 H (256-192-160) + 667 O (192-160-128), 8 members, AEV width 1008; the atoms of the
    two species interleaved as in a water box (O H H), AEV-like non-negative inputs."""
    gen = torch.Generator().manual_seed(0)
    n = 2001
    x = torch.rand((n, 1008), generator=gen) * (torch.rand((n, 1008), generator=gen) < 0.3)      # sparse, >= 0, like an AEV
    species = torch.tensor([3, 0, 0] * 667)
    kinds = []
    for s, widths in ((0, (256, 192, 160)), (3, (192, 160, 128))):
        kd = _networks(widths, 8, 1008, seed=10 + s)
        kd["atoms"] = torch.nonzero(species == s).flatten().to(torch.int32)
        kinds.append(kd)
    _run(kinds, x, 1008)


def test_all_ani2x_species_ragged_tiles_and_unaligned_widths():
    """Seven kinds in one launch with the ANI-2x widths (224 and 160 are not multiples of 64: waves own unequal numbers of
    row blocks), atom counts that leave partial tiles (1, 63, 64, 65, 130 ...), one kind with no atoms, 3 members."""
    widths = [(256, 192, 160), (224, 192, 160), (192, 160, 128), (192, 160, 128), (160, 128, 96), (160, 128, 96), (160, 128, 96)]
    counts = [130, 65, 64, 63, 1, 0, 17]
    gen = torch.Generator().manual_seed(1)
    n = sum(counts)
    perm = torch.randperm(n, generator=gen)
    x = torch.rand((n, 1008), generator=gen)
    kinds, first = [], 0
    for s, (w, c) in enumerate(zip(widths, counts)):
        kd = _networks(w, 3, 1008, seed=20 + s)
        kd["atoms"] = perm[first:first + c].to(torch.int32)
        first += c
        kinds.append(kd)
    _run(kinds, x, 1008)


def test_small_widths_other_input_width_and_negative_inputs():
    """Not ANI-2x: input width 384 (12 K steps), widths 32 / 48 / 80 (padded to 32 / 64 / 96), one member, inputs of both signs
    and larger weights (activations of a few units)."""
    gen = torch.Generator().manual_seed(2)
    x = torch.randn((200, 384), generator=gen)
    kd = _networks((80, 48, 32), 1, 384, seed=5, scale=2.0)
    kd["atoms"] = torch.arange(199, -1, -1, dtype=torch.int32)[:150]          # a subset, reversed: rows are a map
    e, dx = _run([kd], x, 384)
    assert float(dx[150:].abs().max()) >= 0.0 and float(dx[torch.arange(0, 50)].abs().max()) == 0.0   # rows of no kind untouched


@pytest.mark.parametrize("k,weight_scale", [(4, 1.0), (8, 3.0), (12, 8.0)])
def test_activation_scale_is_an_argument(k, weight_scale):
    """nnpops_mlp_frame::act_scale_log2: activations are multiplied by 2^-k before their fp16 split (k = 4 up to round 4).  Larger
    weights (x8 per layer: activations ~500x) with the scale that holds them, same bars; values outside 4..12 are refused."""
    from nnpops_amd.capi import FusedMLP
    gen = torch.Generator().manual_seed(40 + k)
    x = torch.rand((300, 384), generator=gen)
    kinds = []
    for s, widths in enumerate(((96, 64, 32), (128, 96, 64))):
        kd = _networks(widths, 2, 384, seed=50 + s, scale=weight_scale)
        kd["atoms"] = torch.arange(s, 300, 2, dtype=torch.int32)
        kinds.append(kd)
    _run(kinds, x, 384, act_scale_log2=k)
    if k == 12:
        bad = FusedMLP([{key: v.to(DEV) for key, v in kd.items()} for kd in kinds], 384, act_scale_log2=13)
        with pytest.raises(Exception, match="act_scale_log2"):
            bad.forward(x.to(DEV), with_gradient=False)


def test_packer_layout():
    """nnpops_mlp_pack against the layout formula of mlp_fused.hip (fragment = [row block][K step][plane][lane][8])."""
    from nnpops_amd.capi import mlp_pack
    gen = torch.Generator().manual_seed(3)
    rows, cols = 40, 72
    w = torch.randn((rows, cols), generator=gen)
    for transpose in (False, True):
        for permute in (False, True):
            src = w.t().contiguous() if transpose else w
            packed = mlp_pack(src.to(DEV), rows, cols, transpose=transpose, permute=permute).cpu()
            nb, steps = (rows + 15) // 16, (cols + 31) // 32
            packed = packed.view(nb, steps, 2, 64, 8)
            for rb, s, lane, i in [(0, 0, 0, 0), (1, 2, 37, 5), (2, 1, 63, 7), (2, 2, 15, 3), (0, 1, 16, 4)]:
                r16, kg = lane & 15, lane >> 4
                k = 32 * s + ((4 * kg + i if i < 4 else 16 + 4 * kg + i - 4) if permute else 8 * kg + i)
                row = rb * 16 + r16
                v = float(w[row, k]) if row < rows and k < cols else 0.0
                hi = float(torch.tensor(v).half())
                assert float(packed[rb, s, 0, lane, i]) == hi
                assert float(packed[rb, s, 1, lane, i]) == float(torch.tensor((v - hi) * 2048.0).half())


def test_shifted_energy_mean_and_scale_by_scalar():
    """The two small launches around the one-node OptimizedTorchANI step (include/nnpops_hip.h): the ensemble mean promoted to
    float64 and shifted by the self energy exactly as `energies + self_energies` does it (EnergyShifter.py:52), and
    values * float(device scalar) for a float32 or a float64 factor -- bit for bit what the tensor expressions give."""
    from nnpops_amd import capi
    gen = torch.Generator().manual_seed(5)
    kd = _networks((64, 32, 32), 3, 96, seed=8)
    kd["atoms"] = torch.arange(37, dtype=torch.int32)
    mlp = capi.FusedMLP([{k: v.to(DEV) for k, v in kd.items()}], 96)
    x = torch.randn(37, 96, generator=gen).to(DEV)
    mlp.forward(x, with_gradient=False)
    shift = torch.tensor([-1234.56789012345], dtype=torch.float64, device=DEV)
    plain = mlp.energy_mean(1.0 / 3)
    shifted = mlp.energy_mean_shifted(shift, 1.0 / 3)
    assert shifted.dtype == torch.float64 and torch.equal(shifted, plain + shift)
    values = torch.randn(1001, 3, generator=gen).to(DEV)
    for factor in (torch.tensor([2.5000001], device=DEV), torch.tensor([-0.3333333333333], dtype=torch.float64, device=DEV)):
        assert torch.equal(capi.scale_by_scalar(values, factor), values * factor.float())


@pytest.mark.parametrize("live", [[0, 3, 7, 10, 13, 40, 41, 62], [0, 1, 2, 5, 6, 9, 11, 12, 13, 20, 21, 30, 31, 33, 34, 50, 51, 60], list(range(63)),
                                  [5], [62, 0, 31], list(range(1, 63, 4)), list(range(0, 63, 3)) + [61]])
def test_networks_over_the_live_column_blocks_only(live):
    """nnpops_mlp_frame::x_groups / dead_groups / dx_partial (include/nnpops_hip.h): with the AEV blocks of absent species zero, the
    networks packed over the live 16-column blocks give the energies and the input gradient of the networks packed over all 1008
    columns up to the ORDER of the fp32 additions (the live columns share K steps differently; the skipped terms are exact
    zeros): 1e-6 of the largest value, far inside the bars against the float64 reference; the gradient is zero in the dead
    columns.  With the gradient formed inside the forward launch (<= 256 live columns: 8 blocks), by the separate launch
    (18 blocks = 288 columns), and with every block live (the identity map: then bit for bit)."""
    from nnpops_amd.capi import FusedMLP
    gen = torch.Generator().manual_seed(17)
    F, n = 1008, 150
    kinds_host = []
    for s, (widths, atoms) in enumerate((((256, 192, 160), range(0, n, 3)), ((192, 160, 128), [a for a in range(n) if a % 3]))):
        kd = _networks(widths, 3, F, seed=40 + s)
        kd["atoms"] = torch.tensor(list(atoms), dtype=torch.int32)
        kinds_host.append(kd)
    x = torch.zeros(n, F)
    for g in live:
        x[:, 16 * g:16 * g + 16] = torch.rand(n, 16, generator=gen)
    kinds_dev = [{k: v.to(DEV) for k, v in kd.items()} for kd in kinds_host]
    full = FusedMLP(kinds_dev, F)
    part = FusedMLP(kinds_dev, F, live_groups=live)
    assert (getattr(part, "dx_partial", None) is not None) == (16 * len(live) <= 256)
    xd = x.to(DEV)
    e_full = full.forward(xd, with_gradient=True).clone()
    dx_full = full.input_grad(xd, scale=0.5)
    e_part = part.forward(xd, with_gradient=True).clone()
    dx_part = part.input_grad(xd, out=torch.full((n, F), float("nan"), device=DEV), scale=0.5)
    identity = len(live) == F // 16
    assert torch.equal(e_full, e_part) if identity else float((e_full - e_part).abs().max()) <= 1e-6 * float(e_full.abs().max())
    dead = [g for g in range(F // 16) if g not in live]
    for g in dead:
        assert bool((dx_part[:, 16 * g:16 * g + 16] == 0).all())
    cols = torch.tensor([16 * g + c for g in live for c in range(16)], device=DEV)
    ref = dx_full[:, cols]
    if identity:
        assert torch.equal(dx_part, dx_full)
    assert float((dx_part[:, cols] - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    e_ref, dx_ref = _host_reference(kinds_host, x)
    assert float((dx_part.cpu().double() - 0.5 * dx_ref)[:, cols.cpu()].abs().max()) <= 1e-4 * float(dx_ref.abs().max())
    assert torch.equal(part.forward(xd, with_gradient=False), e_part)
    assert float((e_part.cpu().double() - e_ref).abs().max()) <= 1e-5 * float(e_ref.abs().max())
