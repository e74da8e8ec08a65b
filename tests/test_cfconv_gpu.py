"""Parity of the HIP CFConv (through the C ABI) against the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

from nnpops_amd import workloads
from oracle import CFConvNeighborsOracle, CFConvOracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
OUT_RTOL, OUT_ATOL_FRAC = 2e-5, 2e-6       # |diff| <= rtol*|ref| + atol_frac*max|ref|
FORCE_RTOL = 1e-4                          # relative to the largest force component


def _case(pos, box, W, G, cutoff, sigma, act, seed=0, w=None, keep=None):
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    n = pos.shape[0]
    rng = np.random.default_rng(seed)
    if w is None:
        w1 = (0.3 * rng.standard_normal((W, G))).astype(np.float32)
        w2 = (0.2 * rng.standard_normal((W, W))).astype(np.float32)
        b1 = (0.3 * rng.standard_normal(W)).astype(np.float32)
        b2 = (0.3 * rng.standard_normal(W)).astype(np.float32)
        x = rng.standard_normal((n, W)).astype(np.float32)
    else:
        w1, b1, w2, b2, x = w
    gy = rng.standard_normal((n, W)).astype(np.float32)
    periodic = box is not None
    onb = CFConvNeighborsOracle(n, cutoff, periodic)
    onb.build(pos, box)
    ocf = CFConvOracle(n, W, G, cutoff, sigma, act, w1, b1, w2, b2, periodic=periodic)
    y_ref = ocf.forward(onb, pos, x, box)
    xg_ref, pg_ref = ocf.backward(onb, pos, x, gy, box)

    nb = CFConvNeighbors(n, cutoff, periodic)
    cf = CFConv(n, W, G, cutoff, sigma, act, w1, b1, w2, b2, periodic=periodic)
    tpos = torch.tensor(pos, device=DEV)
    tbox = torch.tensor(box, device=DEV) if periodic else None
    tx = torch.tensor(x, device=DEV)
    nb.build(tpos, tbox)
    y = cf.compute(nb, tpos, tx, tbox)
    xg, pg = cf.backprop(nb, tpos, tx, torch.tensor(gy, device=DEV), tbox)
    torch.cuda.synchronize()
    y, xg, pg = y.cpu().numpy(), xg.cpu().numpy(), pg.cpu().numpy()

    # the half list itself: same pairs, same distances
    start, other, dist = onb.export()
    atoms, d = nb.export()
    ref_i = np.repeat(np.arange(n), np.diff(start))
    assert np.array_equal(atoms[0], ref_i) and np.array_equal(atoms[1], other)
    np.testing.assert_allclose(d, dist, rtol=1e-6)

    np.testing.assert_allclose(y, y_ref, rtol=OUT_RTOL, atol=OUT_ATOL_FRAC * np.abs(y_ref).max())
    np.testing.assert_allclose(xg, xg_ref, rtol=OUT_RTOL, atol=OUT_ATOL_FRAC * np.abs(xg_ref).max())
    assert np.abs(pg - pg_ref).max() <= FORCE_RTOL * np.abs(pg_ref).max()
    e_ref = float((y_ref.astype(np.float64) * gy).sum())
    e = float((y.astype(np.float64) * gy).sum())
    assert abs(e - e_ref) <= 1e-5 * float(np.abs(y_ref.astype(np.float64) * gy).sum())
    if keep is not None:
        keep.update(y=y, xg=xg, pg=pg, y_ref=y_ref, xg_ref=xg_ref, pg_ref=pg_ref)
    return y


@pytest.mark.parametrize("tag", ["nonperiodic_ssp", "periodic_ssp", "triclinic_ssp", "nonperiodic_tanh"])
def test_water18_golden(golden_dir, tag):
    """The reference's own fixture (src/schnet/TestCFConv.h:81-247), SchNetPack-generated expectations."""
    g = np.load(f"{golden_dir}/cfconv_water18.npz")
    box = g[f"{tag}_box"] if f"{tag}_box" in g else None
    act = tag.split("_")[1]
    y = _case(g["positions"], box, 8, 5, 2.0, 0.5, act, w=(g["w1"], g["b1"], g["w2"], g["b2"], g["x"]))
    np.testing.assert_allclose(y, g[f"{tag}_output"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("W,G", [(128, 50), (112, 40), (96, 50), (80, 33), (64, 25), (48, 20), (32, 16), (16, 7), (100, 50), (5, 3)])
@pytest.mark.parametrize("act", ["ssp", "tanh"])
def test_random_cluster(W, G, act):
    pos, _ = workloads.conformer(120, seed=W + G)
    _case(pos, None, W, G, 5.0, 0.1 if G >= 25 else 0.4, act, seed=W)


def test_vector_kernels_at_matrix_widths(monkeypatch):
    """W = 128 normally runs on the matrix cores; $NNPOPS_CFCONV_VALU=1 (read at handle creation) keeps the
    vector kernels, which serve every width that is not a multiple of 16, under the same parity bar."""
    monkeypatch.setenv("NNPOPS_CFCONV_VALU", "1")
    pos, _ = workloads.conformer(90, seed=77)
    _case(pos, None, 128, 50, 5.0, 0.1, "ssp", seed=5)
    _case(pos, None, 64, 25, 5.0, 0.1, "tanh", seed=6)


@pytest.mark.parametrize("W,G,act", [(192, 70, "ssp"), (256, 130, "tanh"), (130, 20, "ssp")])
def test_wide_layers_stream_their_weights(W, G, act):
    """Beyond W = 128 the weights no longer fit in LDS next to a wave's tiles: the vector kernel then reads them through
    the caches (a functional path -- the reference accepts any width, so does the drop-in)."""
    pos, _ = workloads.conformer(70, seed=W)
    _case(pos, None, W, G, 4.5, 0.2, act, seed=W + 1)


def test_periodic_box_cells():
    """1500 atoms in a periodic box: the cell-grid neighbour search (core-level periodic variant of config 3)."""
    pos, _, box = workloads.random_box(1500, seed=31)
    _case(pos, box, 128, 50, 5.0, 0.1, "ssp", seed=1)


def test_cell_bins_grow_on_overflow(monkeypatch):
    """Same fixed-capacity bin protocol as the ANI handle (celllist.h): forced overflow, check() grows, rebuild."""
    monkeypatch.setenv("NNPOPS_CELL_BIN_CAP", "4")
    pos, _, box = workloads.random_box(1200, seed=41)
    _case(pos, box, 32, 16, 5.0, 0.4, "ssp", seed=7)


def test_nonperiodic_box_cells():
    pos, _, _ = workloads.random_box(1300, seed=32)
    _case(pos, None, 64, 50, 5.0, 0.1, "tanh", seed=2)


def test_triclinic_box():
    pos, _, box = workloads.triclinic_box(700, seed=33)
    _case(pos, box, 128, 50, 5.0, 0.1, "ssp", seed=3)


def test_half_and_full_list_paths_agree(monkeypatch):
    """The matrix-core widths evaluate the filter network once per pair (filters + owner-computes gather);
    $NNPOPS_CFCONV_HALF=0 (read at handle creation) keeps the kernels that evaluate every pair from both ends.
    Both meet the oracle, and each other far inside the parity bar."""
    pos, _, box = workloads.random_box(1500, seed=51)
    y_half = _case(pos, box, 128, 50, 5.0, 0.1, "ssp", seed=11)
    monkeypatch.setenv("NNPOPS_CFCONV_HALF", "0")
    y_full = _case(pos, box, 128, 50, 5.0, 0.1, "ssp", seed=11)
    assert np.abs(y_half - y_full).max() <= 2e-6 * np.abs(y_full).max()
    pos, _ = workloads.conformer(120, seed=52)              # all-pairs neighbour search, odd number of column blocks
    _case(pos, None, 48, 20, 5.0, 0.1, "tanh", seed=12)


def test_half_list_follows_rebuilds():
    """The pair slots are rebuilt with the rows: a second build on moved atoms (different pairs) with the same handles
    gives what fresh handles give."""
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    n, W, G = 1400, 64, 25
    pos, _, box = workloads.random_box(n, seed=61)
    rng = np.random.default_rng(62)
    moved = (pos + rng.normal(0, 0.8, pos.shape)).astype(np.float32)
    w1 = (0.3 * rng.standard_normal((W, G))).astype(np.float32); w2 = (0.2 * rng.standard_normal((W, W))).astype(np.float32)
    b1 = (0.3 * rng.standard_normal(W)).astype(np.float32); b2 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    x = torch.tensor(rng.standard_normal((n, W)).astype(np.float32), device=DEV)
    gy = torch.tensor(rng.standard_normal((n, W)).astype(np.float32), device=DEV)
    tbox = torch.tensor(box, device=DEV)

    def run(nb, cf, p):
        tp = torch.tensor(p, device=DEV)
        nb.build(tp, tbox)
        y = cf.compute(nb, tp, x, tbox)
        xg, pg = cf.backprop(nb, tp, x, gy, tbox)
        return y.cpu().numpy(), xg.cpu().numpy(), pg.cpu().numpy()

    nb = CFConvNeighbors(n, 5.0, True)
    cf = CFConv(n, W, G, 5.0, 0.1, "ssp", w1, b1, w2, b2, periodic=True)
    first = run(nb, cf, pos)
    second = run(nb, cf, moved)
    again = run(nb, cf, pos)
    fresh = run(CFConvNeighbors(n, 5.0, True), CFConv(n, W, G, 5.0, 0.1, "ssp", w1, b1, w2, b2, periodic=True), moved)
    for a, b in zip(second, fresh):
        assert np.array_equal(a, b)                          # same list, same arithmetic: bitwise
    for a, b in zip(first, again):
        assert np.array_equal(a, b)
    assert not np.array_equal(first[0], second[0])


def test_backward_reads_back_the_filter_rows_of_the_forward_call(monkeypatch):
    """A backward call that follows a forward call on the same build of the same list does not store the filter rows again (the
    gather reads the forward call's).  Same results as with $NNPOPS_CFCONV_REUSE_FILTERS=0 (to the rounding of one filter value);
    after a rebuild on moved atoms the kept rows are stale and are not used: bit for bit what always storing gives."""
    from nnpops_amd.capi import CFConv, CFConvNeighbors
    n, W, G = 1500, 128, 50
    pos, _, box = workloads.random_box(n, seed=71)
    rng = np.random.default_rng(72)
    w1, w2 = (0.3 * rng.standard_normal((W, G))).astype(np.float32), (0.2 * rng.standard_normal((W, W))).astype(np.float32)
    b1, b2 = (0.3 * rng.standard_normal(W)).astype(np.float32), (0.3 * rng.standard_normal(W)).astype(np.float32)
    tx, tg = torch.tensor(rng.standard_normal((n, W)).astype(np.float32), device=DEV), torch.tensor(rng.standard_normal((n, W)).astype(np.float32), device=DEV)
    tbox = torch.tensor(box, device=DEV)
    moved = (pos + 0.3 * rng.standard_normal(pos.shape)).astype(np.float32)
    t1, t2 = torch.tensor(pos, device=DEV), torch.tensor(moved, device=DEV)

    def run(reuse, forward_first):
        monkeypatch.setenv("NNPOPS_CFCONV_REUSE_FILTERS", "1" if reuse else "0")
        nb, cf = CFConvNeighbors(n, 5.0, True), CFConv(n, W, G, 5.0, 0.1, "ssp", w1, b1, w2, b2, periodic=True)
        out = []
        nb.build(t1, tbox)
        if forward_first:
            out.append(cf.compute(nb, t1, tx, tbox).clone())
        out += [t.clone() for t in cf.backprop(nb, t1, tx, tg, tbox)]
        nb.build(t2, tbox)                              # other pairs: the rows kept from the first build must not be read
        out += [t.clone() for t in cf.backprop(nb, t2, tx, tg, tbox)]
        out += [t.clone() for t in cf.backprop(nb, t2, tx, tg, tbox)]      # (and a second backward call on the same build may)
        # a NEW list (possibly at the freed one's address) after a forward call on the old one: another build, whatever its address
        cf.compute(nb, t2, tx, tbox)
        del nb
        nb = CFConvNeighbors(n, 5.0, True)
        nb.build(t1, tbox)
        out += [t.clone() for t in cf.backprop(nb, t1, tx, tg, tbox)]
        torch.cuda.synchronize()
        return out

    kept, stored, alone = run(True, True), run(False, True), run(True, False)

    def close(a, b):
        return float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())

    # (the forward kernel's rows and the backward kernel's are the same numbers up to the last bit: bias added at different points)
    assert torch.equal(kept[0], stored[0]) and close(kept[1], stored[1]) and close(kept[2], stored[2])
    # after the rebuild the first backward call stores its own rows, the second one reads those back: bit for bit what always storing gives
    for k in (3, 4, 5, 6, 7, 8):
        assert torch.equal(kept[k], stored[k]) and torch.equal(kept[k], alone[k - 1])
    assert torch.equal(kept[3], kept[5]) and torch.equal(kept[4], kept[6]) and not close(kept[1], kept[3])


def test_half_list_with_rows_longer_than_a_wave():
    """Twice the usual density: ~105 neighbours per atom overflow the 64-entry rows, check() grows them to 128 and the
    pair slots with them; the slot lookup and the gather then walk rows in two passes of 64."""
    pos, _, box = workloads.random_box(1100, density=0.2, seed=71)
    _case(pos, box, 32, 16, 5.0, 0.4, "ssp", seed=13)


def test_split_fp16_second_layer_is_as_accurate_as_fp32(monkeypatch):
    """Widths 32/64/96/128 run the W x W layer as split-fp16 matrix products (x = hi + 2^-11 lo', three
    v_mfma_f32_16x16x32_f16 per block, fp32 accumulation); $NNPOPS_CFCONV_SPLIT=0 keeps the fp32 matrix instruction.
    Both meet the oracle under the same bar, and the split form is not the less accurate of the two."""
    pos, _, box = workloads.random_box(1500, seed=81)
    split, plain = {}, {}
    _case(pos, box, 128, 50, 5.0, 0.1, "ssp", seed=21, keep=split)
    monkeypatch.setenv("NNPOPS_CFCONV_SPLIT", "0")
    _case(pos, box, 128, 50, 5.0, 0.1, "ssp", seed=21, keep=plain)
    for key in ("y", "xg", "pg"):
        ref = split[key + "_ref"].astype(np.float64)
        err_split = np.abs(split[key] - ref).max() / np.abs(ref).max()
        err_plain = np.abs(plain[key] - ref).max() / np.abs(ref).max()
        assert err_split <= 2.0 * err_plain + 2e-7, (key, err_split, err_plain)
    monkeypatch.delenv("NNPOPS_CFCONV_SPLIT")
    pos, _ = workloads.conformer(150, seed=82)
    _case(pos, None, 96, 33, 5.0, 0.2, "tanh", seed=22)
    _case(pos, None, 32, 9, 4.0, 0.3, "ssp", seed=23)


@pytest.mark.parametrize("W,G,act", [(128, 50, "ssp"), (96, 33, "tanh"), (64, 20, "ssp"), (32, 9, "tanh")])
def test_register_fed_filters_kernels_agree_with_the_plane_kernels(monkeypatch, W, G, act):
    """Round 6: where both layers run as split products the forward filters kernel takes 32 pairs per wave and feeds layer 2 from registers
    (cfconv_filters_h2x2), the backward one runs one matrix pass per layer on register-fed operands (cfconv_filters_h2b, one wave per
    SIMD at widths 96 / 128, two below); $NNPOPS_CFCONV_FWD32=0 / $NNPOPS_CFCONV_BWD1=0 keep the kernels that go through LDS planes
    (still the ones for more than 63 Gaussians).  Same arithmetic, the 32 products of a matrix step added in another order: both
    meet the oracle, and each other far inside the parity bar -- also with the backward kernel's two wave counts swapped."""
    pos, _, box = workloads.random_box(1400, seed=91)
    new, old, swapped = {}, {}, {}
    _case(pos, box, W, G, 5.0, 0.1, act, seed=31, keep=new)
    monkeypatch.setenv("NNPOPS_CFCONV_BWD_WAVES", "8" if W >= 96 else "4")
    _case(pos, box, W, G, 5.0, 0.1, act, seed=31, keep=swapped)
    monkeypatch.delenv("NNPOPS_CFCONV_BWD_WAVES")
    monkeypatch.setenv("NNPOPS_CFCONV_FWD32", "0")
    monkeypatch.setenv("NNPOPS_CFCONV_BWD1", "0")
    _case(pos, box, W, G, 5.0, 0.1, act, seed=31, keep=old)
    for key in ("y", "xg", "pg"):
        scale = np.abs(old[key]).max()
        assert np.abs(new[key] - old[key]).max() <= 3e-6 * scale, key
        assert np.abs(swapped[key] - new[key]).max() <= 3e-6 * scale, key
    # Round 6: dY1 goes into its fp16 planes scaled to the top of the fp16 range (ConvParams::dy_scale).  Unscaled, the forces under tanh --
    # whose saturated neurons leave most of dY1 orders of magnitude below its largest entries -- sat 2e-5 ... 7e-5 of the largest force from
    # the oracle, in every split-fp16 kernel since round 3; now they sit where the fp32 matrix kernel sits (tools/cfconv_split_error.py).
    for keep in (new, old):
        assert np.abs(keep["pg"] - keep["pg_ref"]).max() <= 8e-6 * np.abs(keep["pg_ref"]).max()
    # a ragged last pass (pairs not a multiple of 32) and a molecule (all-pairs list)
    monkeypatch.delenv("NNPOPS_CFCONV_FWD32")
    monkeypatch.delenv("NNPOPS_CFCONV_BWD1")
    mol, _ = workloads.conformer(61, seed=92)
    _case(mol, None, W, G, 5.0, 0.2, act, seed=32)


def test_weights_outside_the_fp16_range_keep_the_fp32_layer():
    """The split form is only taken when the weights bound every operand below the fp16 range (checked when the handle is
    created); layers with very large weights go through the fp32 matrix instruction and still match."""
    pos, _ = workloads.conformer(130, seed=83)
    n, W, G = len(pos), 64, 20
    rng = np.random.default_rng(84)
    w1 = (0.3 * rng.standard_normal((W, G))).astype(np.float32)
    w2 = (2.0e4 * rng.standard_normal((W, W))).astype(np.float32)          # max |W2| ~ 7e4: beyond fp16
    b1 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    b2 = (0.3 * rng.standard_normal(W)).astype(np.float32)
    x = rng.standard_normal((n, W)).astype(np.float32)
    _case(pos, None, W, G, 5.0, 0.3, "tanh", seed=24, w=(w1, b1, w2, b2, x))


def test_many_gaussians_at_a_matrix_width():
    """W = 128 with 200 Gaussians: the two weight matrices no longer fit in LDS beside a tile, the matrix-core kernels
    step aside for the vector kernel with streamed weights (any G up to 256 is accepted)."""
    pos, _ = workloads.conformer(60, seed=91)
    _case(pos, None, 128, 200, 5.0, 0.05, "ssp", seed=31)


def test_reference_benchmark_parameters():
    """The parameters of the reference's own CFConv benchmark (src/schnet/BenchmarkCudaCFConv.cu:64-67,87: width 128,
    50 Gaussians, cutoff 10 A, Gaussian width 0.2, shifted softplus, N(0, 1) weights) against the oracle, on a periodic box just
    large enough for that cutoff (1 000 atoms, 21.5 A: ~420 neighbours per atom, rows far longer than a wave)."""
    pos, _, box = workloads.random_box(1000, density=0.1, seed=13)
    rng = np.random.default_rng(0)
    W, G = 128, 50
    w = (rng.standard_normal((W, G)).astype(np.float32), rng.standard_normal(W).astype(np.float32),
         rng.standard_normal((W, W)).astype(np.float32), rng.standard_normal(W).astype(np.float32),
         rng.standard_normal((1000, W)).astype(np.float32))
    _case(pos, box, W, G, 10.0, 0.2, "ssp", seed=1, w=w)
