"""GPU parity: every HIP stage (called through the C ABI, arfe_amd.functional) against the CPU oracle on the
same seeded inputs, and against the golden vectors generated from the reference (tests/golden).

Tolerance: north_star asks for 1e-3 relative in fp32 (max|a-b|/max|b|).  The streaming kernels are exact fp32
and the dense contractions run as bf16x3 split MFMA (~1e-5 relative), so the asserts are set 3-10x tighter than
the north-star bound (TOL / TOL_GRAD) to catch indexing bugs; integer/index results must be bit-exact."""
import os

import pytest
import torch

from conftest import load_golden, relerr, relerr_l2

pytestmark = pytest.mark.gpu

TOL = 1e-4          # fp32 streaming kernels are ~1e-6; the bf16x3 split-MFMA contractions ~1e-5 per op
TOL_GRAD = 3e-4     # gradients (chains of contractions, atomics, long reductions)
NORTH_STAR_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def Fn():
    from arfe_amd import functional
    from arfe_amd import hip
    hip.lib()       # fail loudly if the HIP library is missing
    return functional


@pytest.fixture(scope="module")
def O():
    from oracle import arfe_oracle
    return arfe_oracle


def cu(ts, dev):
    return [t.to(dev).contiguous() for t in ts]


PYRAMIDS = {
    'tiny': ((16, 24), (8, 12), (4, 6), (2, 3), (1, 2)),
    'ragged': ((40, 67), (20, 34), (10, 17), (5, 9), (3, 5)),
    'mid': ((64, 96), (32, 48), (16, 24), (8, 12), (4, 6)),
    'odd': ((50, 83), (25, 42), (13, 21), (7, 11), (4, 6)),
}


@pytest.mark.parametrize('shape', list(PYRAMIDS))
@pytest.mark.parametrize('r', [0, 2, 3, 4])
def test_gather_matches_oracle(Fn, O, dev, shape, r):
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(2, 12, h, w, generator=g) for h, w in PYRAMIDS[shape]]
    m, argmax = Fn.fpn_gather_op(cu(xs, dev), r)
    ref = O.fpn_gather(xs, r, explicit=True)
    assert torch.equal(m.cpu(), ref), "gather is max/copy/add/div: must be bit-exact"
    hr, wr = xs[r].shape[2:]
    for i in range(r):
        _, idx = torch.nn.functional.adaptive_max_pool2d(xs[i], (hr, wr), return_indices=True)
        assert torch.equal(argmax[i].cpu().long(), idx), f"argmax level {i}"


@pytest.mark.parametrize('shape', list(PYRAMIDS))
def test_gatemap_matches_oracle(Fn, O, dev, shape):
    g = torch.Generator().manual_seed(2)
    C = 20
    xs = [torch.randn(2, C, h, w, generator=g) for h, w in PYRAMIDS[shape]]
    w1 = [torch.randn(1, C, 3, 3, generator=g) * 0.1 for _ in xs]
    w2 = [torch.randn(1, C, 3, 3, generator=g) * 0.1 for _ in xs]
    b1 = [torch.randn(1, generator=g) * 0.1 for _ in xs]
    b2 = [torch.randn(1, generator=g) * 0.1 for _ in xs]
    a, b = Fn.fpn_gatemap_op(cu(xs, dev), cu(w1, dev), cu(b1, dev), cu(w2, dev), cu(b2, dev))
    for i, x in enumerate(xs):
        ra, rb = O.gate_maps(x, w1[i], b1[i], w2[i], b2[i])
        assert relerr(a[i], ra) <= TOL and relerr(b[i], rb) <= TOL, i
        assert (a[i].cpu() > 0).eq(ra > 0).float().mean() > 0.999


@pytest.mark.parametrize('shape,r,C', [('tiny', 2, 12), ('mid', 2, 20), ('mid', 1, 8), ('ragged', 2, 12), ('mid', 2, 6),
                                       ('bench', 2, 8)])
def test_gather_gate_fused_matches_oracle(Fn, O, dev, shape, r, C):
    """arfe_fpn_gather_gate_fwd: gather and both gate maps from one read of the pyramid.  m and argmax bit-exact (the fused
    kernel's window maximum is a lane-parallel fold that must keep adaptive_max_pool2d's first-maximum scan order), gate maps
    within TOL; 'ragged' and C = 6 are pyramids the fused kernel refuses (the call runs the two separate ops inside), 'bench'
    is the 1333x800 pyramid (4x4 and 2x2 windows, a generic top level)."""
    sizes = O.PYRAMID_1333x800 if shape == 'bench' else PYRAMIDS[shape]
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(2, C, h, w, generator=g) for h, w in sizes]
    xs[0][0, 1, :4, :4] = 0.25                 # a window of ties: the FIRST maximum's index must win
    xs[1][1, 0, 2:4, 2:4] = 1.5
    w1 = [torch.randn(1, C, 3, 3, generator=g) * 0.1 for _ in xs]
    w2 = [torch.randn(1, C, 3, 3, generator=g) * 0.1 for _ in xs]
    b1 = [torch.randn(1, generator=g) * 0.1 for _ in xs]
    b2 = [torch.randn(1, generator=g) * 0.1 for _ in xs]
    m, argmax, a, b = Fn.fpn_gather_gate_op(cu(xs, dev), cu(w1, dev), cu(b1, dev), cu(w2, dev), cu(b2, dev), r)
    ref = O.fpn_gather(xs, r, explicit=True)
    assert torch.equal(m.cpu(), ref)
    hr, wr = xs[r].shape[2:]
    for i in range(r):
        _, idx = torch.nn.functional.adaptive_max_pool2d(xs[i], (hr, wr), return_indices=True)
        assert torch.equal(argmax[i].cpu().long(), idx), f"argmax level {i}"
    for i, x in enumerate(xs):
        ra, rb = O.gate_maps(x, w1[i], b1[i], w2[i], b2[i])
        assert relerr(a[i], ra) <= TOL and relerr(b[i], rb) <= TOL, i


@pytest.mark.parametrize('shape', list(PYRAMIDS))
@pytest.mark.parametrize('r', [2, 3])
def test_scatter_matches_oracle(Fn, O, dev, shape, r):
    g = torch.Generator().manual_seed(3)
    C = 9
    xs = [torch.randn(2, C, h, w, generator=g) for h, w in PYRAMIDS[shape]]
    bsf = torch.randn(2, C, *PYRAMIDS[shape][r], generator=g)
    a = [torch.rand(2, 1, h, w, generator=g) for h, w in PYRAMIDS[shape]]
    b = [torch.rand(2, 1, h, w, generator=g) for h, w in PYRAMIDS[shape]]
    outs = Fn.fpn_scatter_op(cu(xs, dev), bsf.to(dev), cu(a, dev), cu(b, dev), r)
    for i in range(5):
        ref = O.fpn_scatter(xs[i], bsf, a[i], b[i], explicit=True)
        assert relerr(outs[i], ref) <= 1e-6, i


@pytest.mark.parametrize('shape', list(PYRAMIDS))
@pytest.mark.parametrize('r,C', [(0, 9), (1, 70), (2, 9), (2, 70), (2, 256), (3, 33), (4, 9)])
def test_scatter_bwd_matches_oracle_autograd(Fn, O, dev, shape, r, C):
    """arfe_fpn_scatter_bwd (dpre_a, dpre_b, dbsf) against autograd of the oracle's scatter on every refine level: the
    single-pass kernel (`fpn_scatter_bwd_fused_kernel<0|1|2>`: 'tiny' / 'mid' pyramids with r <= 2, all three channel
    chunkings) and the two-kernel form (ragged / odd pyramids, r = 3, 4)."""
    g = torch.Generator().manual_seed(40 + r)
    sizes = PYRAMIDS[shape]
    bsf = torch.randn(2, C, *sizes[r], generator=g).requires_grad_(True)
    pa = [torch.randn(2, 1, h, w, generator=g).requires_grad_(True) for h, w in sizes]
    pb = [torch.randn(2, 1, h, w, generator=g).requires_grad_(True) for h, w in sizes]
    a = [torch.tanh(torch.relu(t)) for t in pa]
    b = [torch.tanh(torch.relu(t)) for t in pb]
    cots = [torch.randn(2, C, h, w, generator=g) for h, w in sizes]
    loss = sum((O.fpn_scatter(torch.zeros(2, C, h, w), bsf, a[i], b[i], explicit=True) * cots[i]).sum()
               for i, (h, w) in enumerate(sizes))
    loss.backward()
    dpa, dpb, dbsf = Fn.fpn_scatter_bwd_op(cu(cots, dev), bsf.detach().to(dev), cu([t.detach() for t in a], dev),
                                           cu([t.detach() for t in b], dev), r)
    assert relerr(dbsf, bsf.grad) <= 2e-5
    for i in range(5):
        assert relerr(dpa[i], pa[i].grad) <= 2e-5 and relerr(dpb[i], pb[i].grad) <= 2e-5, i


@pytest.mark.parametrize('B,C,h,w', [(1, 8, 4, 6), (2, 16, 10, 17), (1, 256, 13, 21), (2, 64, 25, 42)])
def test_nonlocal_fwd_bwd_matches_oracle(Fn, O, dev, B, C, h, w):
    g = torch.Generator().manual_seed(4)
    p = {k: v for k, v in O.arfpn_params(C, 5, seed=5).items() if k.startswith('refine')}
    for k in list(p):
        if k.endswith('bias'):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
        p[k] = p[k].requires_grad_(True)
    m = torch.randn(B, C, h, w, generator=g, requires_grad=True)
    cot = torch.randn(B, C, h, w, generator=g)
    ref = O.nonlocal_refine(m, p)
    (ref * cot).sum().backward()
    names = ['theta', 'phi', 'g', 'conv_out']
    W = [p[f'refine.{n}.conv.weight'].detach().view(C, C).to(dev) for n in names]
    Bv = [p[f'refine.{n}.conv.bias'].detach().to(dev) for n in names]
    md = m.detach().to(dev)
    bsf, saved = Fn.nonlocal_fwd_op(md, W[0], Bv[0], W[1], Bv[1], W[2], Bv[2], W[3], Bv[3])
    assert relerr(bsf, ref) <= TOL
    dm, grads = Fn.nonlocal_bwd_op(cot.to(dev), md, W[0], W[1], W[2], W[3], saved)
    assert relerr(dm, m.grad) <= TOL_GRAD
    dwt, dbt, dwp, dbp, dwg, dbg, dwo, dbo = grads
    for got, n in ((dwt, 'theta'), (dwp, 'phi'), (dwg, 'g'), (dwo, 'conv_out')):
        assert relerr(got.view(C, C, 1, 1), p[f'refine.{n}.conv.weight'].grad) <= TOL_GRAD, n
    for got, n in ((dbt, 'theta'), (dbg, 'g'), (dbo, 'conv_out')):
        assert relerr(got, p[f'refine.{n}.conv.bias'].grad) <= TOL_GRAD, n
    # d/d(b_phi) is identically zero (row-constant logit shift): only rounding noise on both sides
    assert dbp.abs().max().item() <= 1e-3 * max(dbt.abs().max().item(), 1e-12)


def _run_arfpn(Fn, dev, d, r):
    xs = [d[f'x{i}'].to(dev).requires_grad_(True) for i in range(5)]
    p = {k[2:]: v.to(dev).requires_grad_(True) for k, v in d.items() if k.startswith('p.')}
    C = xs[0].shape[1]
    w1 = [p[f'reduce_convs.{i}.conv.weight'] for i in range(5)]
    b1 = [p[f'reduce_convs.{i}.conv.bias'] for i in range(5)]
    w2 = [p[f'reduce_convs2.{i}.conv.weight'] for i in range(5)]
    b2 = [p[f'reduce_convs2.{i}.conv.bias'] for i in range(5)]
    nl = []
    for n in ('theta', 'phi', 'g', 'conv_out'):
        nl += [p[f'refine.{n}.conv.weight'].view(C, C), p[f'refine.{n}.conv.bias']]
    outs = Fn.arfpn(xs, w1, b1, w2, b2, *nl, refine_level=r)
    return xs, p, outs


@pytest.mark.parametrize('name', ['arfpn_tiny_r2.npz', 'arfpn_tiny_r3.npz', 'arfpn_ragged_r2.npz',
                                  'arfpn_ragged_r3.npz', 'arfpn_peaky_r2.npz'])
def test_arfpn_golden_fwd_bwd(Fn, dev, name):
    d = load_golden(name)
    r = int(d['refine_level'])
    xs, p, outs = _run_arfpn(Fn, dev, d, r)
    for i in range(5):
        assert relerr(outs[i], d[f'out{i}']) <= TOL, (name, i)
    sum((o * d[f'cot{i}'].to(dev)).sum() for i, o in enumerate(outs)).backward()
    for i in range(5):
        assert relerr(xs[i].grad, d[f'dx{i}']) <= TOL_GRAD, (name, 'dx', i)
    ref_scale = d['dp.refine.theta.conv.bias'].abs().max().item()
    for k, v in p.items():
        if k == 'refine.phi.conv.bias':
            assert v.grad.abs().max().item() <= 1e-3 * ref_scale
            continue
        assert relerr(v.grad, d['dp.' + k]) <= TOL_GRAD, (name, k)


def test_arfpn_full_size_checksums(Fn, O, dev):
    """BASELINE size (B=1, 256 ch, 1333x800 pyramid) against checksums of the reference's own run."""
    import numpy as np
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'arfpn_full_r2_checksum.npz'))
    p = {k: v.to(dev).requires_grad_(True) for k, v in O.arfpn_params(256, 5, seed=11).items()}
    xs = [x.to(dev).requires_grad_(True) for x in O.synth_pyramid(1, 256, seed=1234)]
    cots = [c.to(dev) for c in O.synth_pyramid(1, 256, seed=4321)]
    w1 = [p[f'reduce_convs.{i}.conv.weight'] for i in range(5)]
    b1 = [p[f'reduce_convs.{i}.conv.bias'] for i in range(5)]
    w2 = [p[f'reduce_convs2.{i}.conv.weight'] for i in range(5)]
    b2 = [p[f'reduce_convs2.{i}.conv.bias'] for i in range(5)]
    nl = []
    for n in ('theta', 'phi', 'g', 'conv_out'):
        nl += [p[f'refine.{n}.conv.weight'].view(256, 256), p[f'refine.{n}.conv.bias']]
    outs = Fn.arfpn(xs, w1, b1, w2, b2, *nl, refine_level=2)
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()

    def check(t, key, tol):
        t = t.detach().double().flatten().cpu()
        s = z[key + '.samples']
        got = t[torch.from_numpy(z[key + '.idx'])].numpy()
        assert np.abs(got - s).max() <= tol * max(np.abs(s).max(), 1e-30), key
        assert abs(t.abs().sum().item() - z[key + '.abssum']) <= tol * z[key + '.abssum'], key

    for i in range(5):
        check(outs[i], f'out{i}', NORTH_STAR_TOL)
        check(xs[i].grad, f'dx{i}', NORTH_STAR_TOL)
    for k, v in p.items():
        if k != 'refine.phi.conv.bias':
            check(v.grad, 'dp.' + k, NORTH_STAR_TOL)


def test_make_rois_bit_exact(Fn, dev):
    d = load_golden('rois_make.npz')
    h, w = Fn.make_rois_op(d['rois'].to(dev))
    assert torch.equal(h.cpu(), d['h_rois']) and torch.equal(w.cpu(), d['w_rois'])


def test_map_roi_levels_bit_exact(Fn, O, dev):
    d = load_golden('roi_levels.npz')
    for nl, key in ((5, 'levels5'), (4, 'levels4')):
        lv = Fn.map_roi_levels_op(d['rois'].to(dev), nl)
        assert lv.dtype == torch.int64 and torch.equal(lv.cpu(), d[key])
    # against the oracle (torch-CPU log2 on THIS host) on the bench's synthetic RoIs and their enlarged sets
    rois = O.synth_rois(2, 512)
    h, w = O.make_rois(rois)
    for r in (rois, h, w):
        assert torch.equal(Fn.map_roi_levels_op(r.to(dev), 5).cpu(), O.map_roi_levels(r, 5))
    assert Fn.map_roi_levels_op(torch.zeros(0, 5, device=dev), 5).numel() == 0


def test_rff_roi_sets_single_launch_is_bit_exact(Fn, O, dev):
    """arfe_rff_roi_sets == arfe_rff_make_rois + 3 x arfe_map_roi_levels, against the golden RoIs and the oracle."""
    d = load_golden('rois_make.npz')
    h, w, lv = Fn.rff_roi_sets_op(d['rois'].to(dev), 5)
    assert torch.equal(h.cpu(), d['h_rois']) and torch.equal(w.cpu(), d['w_rois'])
    rois = torch.cat([O.synth_rois(2, 512), load_golden('roi_levels.npz')['rois']])
    ho, wo = O.make_rois(rois)
    for nl in (5, 4):
        h, w, lv = Fn.rff_roi_sets_op(rois.to(dev), nl)
        assert torch.equal(h.cpu(), ho) and torch.equal(w.cpu(), wo)
        for got, r in zip(lv, (rois, wo, ho)):          # level order: ori, w, h
            assert got.dtype == torch.int64 and torch.equal(got.cpu(), O.map_roi_levels(r, nl))
    h, w, lv = Fn.rff_roi_sets_op(torch.zeros(0, 5, device=dev), 5)
    assert h.shape == (0, 5) and all(l.numel() == 0 for l in lv)


def test_roialign_bwd_prepared_route_equals_single_call(Fn, O, dev):
    """The backward split into _prepare (side stream, during the forward) + _prepared gives the gradients of
    arfe_roialign_bwd (to summation order), and the autograd node that uses it matches the node fed with separately computed levels."""
    g = torch.Generator().manual_seed(18)
    C = 24
    feats = [torch.randn(2, C, h, w, generator=g).to(dev) for h, w in PYRAMIDS['mid']]
    rois = O.synth_rois(2, 90, img_w=384, img_h=256, seed=4).to(dev)
    h_r, w_r, lv = Fn.rff_roi_sets_op(rois, 5)
    sets = [rois, w_r, h_r]
    gout = torch.randn(rois.shape[0], 3 * C, 7, 7, generator=g).to(dev)
    shapes = [f.shape for f in feats]
    one = Fn.roialign_bwd_op(shapes, O.STRIDES, sets, lv, gout)
    ws = Fn.roialign_bwd_prepare_op(feats, O.STRIDES, sets, lv)
    two = Fn.roialign_bwd_op(shapes, O.STRIDES, sets, lv, gout, prepared=ws)
    for a, b in zip(one, two):
        # the tile lists are filled through atomic cursors and then put in RoI order (roi_list_sort_kernel): two runs add the
        # RoIs of a tile in the same order, and with no list sliced (the default) nothing is summed atomically
        assert torch.equal(a, b) or os.environ.get('ARFE_RB_SEG')
        assert relerr(a, b) <= 1e-6
    # the same in three calls: the atomically summed tiles zeroed ahead of time into early-allocated gradients
    pz = Fn.roialign_bwd_prezero_op(shapes, O.STRIDES, sets, ws)
    three = Fn.roialign_bwd_op(shapes, O.STRIDES, sets, lv, gout, prepared=ws, prezeroed=pz)
    for a, c in zip(one, three):
        assert relerr(a, c) <= 1e-6
    fa = [f.clone().requires_grad_(True) for f in feats]
    fb = [f.clone().requires_grad_(True) for f in feats]
    ya = Fn.roi_extract(fa, sets, O.STRIDES)
    yb = Fn.roi_extract(fb, sets, O.STRIDES, levels=lv)
    assert torch.equal(ya, yb)
    (ya * gout).sum().backward()
    (yb * gout).sum().backward()
    for a, b, c in zip(fa, fb, one):
        assert relerr(a.grad, b.grad) <= 1e-6 and relerr(a.grad, c) <= 1e-6
    # no gradient wanted: no prepare, forward still right
    with torch.no_grad():
        assert torch.equal(Fn.roi_extract(feats, sets, O.STRIDES, levels=lv), ya)


def test_roialign_forward_in_groups_equals_single_call(Fn, O, dev):
    """arfe_roialign_fwd_prepare + arfe_roialign_fwd_group (one RoI set per launch, processing groups w, h, ori) fill the same
    [R, 3C, 7, 7] concat as the single call, bit for bit; each group writes only its own channel slice."""
    C = 8
    g = torch.Generator().manual_seed(29)
    feats = [torch.randn(2, C, h, w, generator=g).to(dev) for h, w in PYRAMIDS['mid']]
    rois = O.synth_rois(2, 40, img_w=384, img_h=256, seed=6).to(dev)
    h_r, w_r, lv = Fn.rff_roi_sets_op(rois, 5)
    sets = [rois, w_r, h_r]
    ref = torch.empty(rois.shape[0], 3 * C, 7, 7, device=dev)
    Fn.roialign_fwd_op(feats, O.STRIDES, sets, lv, ref)
    ws = Fn.roialign_fwd_prepare_op(feats, O.STRIDES, sets, lv)
    out = torch.full_like(ref, float('nan'))
    for grp, s in ((0, 1), (1, 2), (2, 0)):                 # group g holds RoI set (g + 1) % 3
        Fn.roialign_fwd_group_op(feats, O.STRIDES, sets, lv, out, ws, grp, 1)
        assert torch.equal(out[:, s * C:(s + 1) * C], ref[:, s * C:(s + 1) * C])
        untouched = [t for t in range(3) if t not in [(k + 1) % 3 for k in range(grp + 1)]]
        assert all(torch.isnan(out[:, t * C:(t + 1) * C]).all() for t in untouched)
    assert torch.equal(out, ref)


def test_roialign_golden(Fn, dev):
    d = load_golden('roialign.npz')
    feats = [d[f'feat{i}'].to(dev).requires_grad_(True) for i in range(5)]
    rois = d['rois'].to(dev)
    h_r, w_r = Fn.make_rois_op(rois)
    cat = Fn.roi_extract(feats, [rois, w_r, h_r], (4, 8, 16, 32, 64))
    assert relerr(cat, d['cat']) <= TOL
    (cat * d['cot'].to(dev)).sum().backward()
    for i in range(5):
        g = feats[i].grad if feats[i].grad is not None else torch.zeros_like(feats[i])
        assert relerr(g, d[f'dfeat{i}']) <= TOL_GRAD, i
    # fixed sample_num through the single-level layer
    from arfe_amd import RoIAlign
    layer = RoIAlign(7, 1.0 / 8, sample_num=2)
    assert relerr(layer(d['feat1'].to(dev), rois), d['single_lvl1_sn2']) <= TOL


def test_roialign_realistic_geometry_matches_oracle(Fn, O, dev):
    """The bench's synthetic RoIs (16..512 px, clipped to 1344x800) and their enlarged sets on the full-size
    pyramid geometry (12 channels): forward concat and the tile-binned backward against the C oracle."""
    g = torch.Generator().manual_seed(8)
    C = 12
    feats_cpu = [torch.randn(2, C, h, w, generator=g) for h, w in O.PYRAMID_1333x800]
    rois = O.synth_rois(2, 160, seed=5)
    # + the whole image, thin long boxes whose windows exceed a wave's 64 columns (table path in 64-column chunks, up to
    # 159) or the 160-entry weight tables altogether (per-sample fallback), a tall one, a degenerate one
    rois = torch.cat([rois, torch.tensor([[0, 0., 0., 1343., 799.], [1, 2., 790., 1340., 799.],
                                          [1, 1300., 0., 1343., 799.], [0, 640., 400., 640., 400.],
                                          [0, 100., 300., 800., 320.], [1, 40., 500., 1100., 530.],
                                          [0, 10., 10., 30., 700.], [1, 600., 5., 1250., 12.]])])
    fo = [f.clone().requires_grad_(True) for f in feats_cpu]
    ref = O.roi_extract3(fo, rois)
    cot = torch.randn(ref.shape, generator=g)
    (ref * cot).sum().backward()
    fd = [f.to(dev).requires_grad_(True) for f in feats_cpu]
    rd = rois.to(dev)
    h_r, w_r = Fn.make_rois_op(rd)
    got = Fn.roi_extract(fd, [rd, w_r, h_r], O.STRIDES)
    assert relerr(got, ref) <= TOL
    (got * cot.to(dev)).sum().backward()
    for i in range(5):
        assert relerr(fd[i].grad, fo[i].grad if fo[i].grad is not None else torch.zeros_like(fo[i])) <= TOL_GRAD, i


@pytest.mark.parametrize("R,in_f,out_f", [(1024, 1024, 401), (96, 3136, 64), (200, 70, 33), (1024, 1024, 1024)])
def test_linear_ops_with_and_without_splitk_scratch(Fn, dev, R, in_f, out_f):
    """arfe_linear_fwd / _bwd (nn.Linear of the shared FCs and the stacked predictors): the slab split-K route (scratch
    given: bias, ReLU, mask and accumulate are applied by the fixed-order reduce) and the scratch-less route (atomic partial
    sums + separate activation pass) against an fp64 reference."""
    import ctypes
    from arfe_amd import hip
    from arfe_amd.hip import ptr, stream, check
    g = torch.Generator().manual_seed(R + in_f)
    x = torch.randn(R, in_f, generator=g).relu().to(dev)           # post-ReLU activations: the mask is not trivial
    w = (torch.randn(out_f, in_f, generator=g) * 0.05).to(dev)
    b = torch.randn(out_f, generator=g).to(dev)
    dy = torch.randn(R, out_f, generator=g).to(dev)
    base = torch.randn(R, in_f, generator=g).to(dev)
    xd, wd, bd, dyd = x.double(), w.double(), b.double(), dy.double()
    y_ref = (xd @ wd.t() + bd).relu()
    dx_ref = (base.double() + dyd @ wd) * (xd > 0)
    dw_ref, db_ref = dyd.t() @ xd, dyd.sum(0)
    L = hip.lib()
    nbytes = L.arfe_linear_workspace_bytes(R, in_f, out_f)
    for ws in (torch.empty(nbytes, dtype=torch.uint8, device=dev), None):
        wp, wn = (ptr(ws), ws.numel()) if ws is not None else (0, 0)
        y = torch.empty(R, out_f, device=dev)
        check(L.arfe_linear_fwd(R, in_f, out_f, ptr(x), ptr(w), ptr(b), 1, ptr(y), wp, wn, stream()), "fwd")
        assert relerr(y, y_ref.float()) <= TOL
        dx = base.clone()
        dw, db = torch.empty_like(w), torch.empty(out_f, device=dev)
        check(L.arfe_linear_bwd(R, in_f, out_f, ptr(x), ptr(w), ptr(dy), ptr(dx), 1, 1, ptr(dw), ptr(db), wp, wn, stream()), "bwd")
        assert relerr(dx, dx_ref.float()) <= TOL and relerr(dw, dw_ref.float()) <= TOL and relerr(db, db_ref.float()) <= TOL


@pytest.mark.parametrize("classes,agnostic,R", [(7, True, 33), (3, False, 64), (80, True, 40)])
def test_rff_head_predictor_shapes_match_oracle(O, dev, classes, agnostic, R):
    """The stacked predictor contraction ([classes + 1; 4 or 4 * classes] rows) for class counts and RoI counts other than the
    bench's, class-agnostic regression included: MultiBBoxHead forward + backward against the oracle's head."""
    import arfe_amd
    torch.manual_seed(classes + R)
    C, fc = 32, 40
    head = arfe_amd.build_head(dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=fc,
                                    roi_feat_size=7, num_classes=classes, reg_class_agnostic=agnostic)).to(dev)
    for p in head.parameters():
        p.data.normal_(0, 0.05)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(R, 3 * C, 7, 7, generator=g)
    xo = x.clone().requires_grad_(True)
    po = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in head.state_dict().items()}
    co, ro = O.rff_head(xo, po)
    assert co.shape == (R, classes + 1) and ro.shape == (R, 4 if agnostic else 4 * classes)
    gc, gr = torch.randn(co.shape, generator=g), torch.randn(ro.shape, generator=g)
    ((co * gc).sum() + (ro * gr).sum()).backward()
    xd = x.to(dev).requires_grad_(True)
    cd, rd = head(xd)
    assert relerr(cd, co.detach()) <= TOL and relerr(rd, ro.detach()) <= TOL
    ((cd * gc.to(dev)).sum() + (rd * gr.to(dev)).sum()).backward()
    # gradients in L2: with random weights a pre-activation within rounding of zero may flip its ReLU between the two
    # implementations, which moves single elements by a whole term (max-norm 5e-3 in one of these cases) but not the tensor
    assert relerr_l2(xd.grad, xo.grad) <= 2e-3
    for k, v in head.named_parameters():
        assert relerr_l2(v.grad, po[k].grad) <= 2e-3, k


def test_rff_head_with_prepacked_weights_equals_plain_call(Fn, dev):
    """MultiBBoxHead.prepack() (weight re-layout launched early on the side stream) changes scheduling only."""
    import arfe_amd
    torch.manual_seed(5)
    head = arfe_amd.build_head(dict(type='MultiRoIsBBoxHead', in_channels=32, conv_out_channels=32, fc_out_channels=48,
                                    roi_feat_size=7, num_classes=80)).to(dev)
    head.init_weights()
    x = torch.randn(70, 96, 7, 7, device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ca, ra = head(xa)
    cb, rb = head(xb, head.prepack())
    # split-K atomics: two plain calls differ by up to ~5e-6 (tools/probe/head_repeat.py); the bound is that noise
    assert relerr(ca, cb) <= 3e-5 and relerr(ra, rb) <= 3e-5
    ga = torch.autograd.grad((ca.sum() + ra.sum()), [xa] + list(head.parameters()))
    gb = torch.autograd.grad((cb.sum() + rb.sum()), [xb] + list(head.parameters()))
    for a, b in zip(ga, gb):
        assert relerr(a, b) <= 1e-4


@pytest.mark.parametrize('name', ['rff_head_r8.npz', 'rff_head_r40.npz'])
def test_rff_head_golden(Fn, dev, name):
    d = load_golden(name)
    x = d['x'].to(dev).requires_grad_(True)
    p = {k[2:]: v.to(dev).requires_grad_(True) for k, v in d.items() if k.startswith('p.')}
    cls, reg = Fn.rff_head(x, p['wh_conv.conv.weight'], p['wh_conv.conv.bias'], p['hh_conv.conv.weight'],
                           p['hh_conv.conv.bias'], p['final_conv.conv.weight'], p['final_conv.conv.bias'],
                           p['shared_fcs.0.weight'], p['shared_fcs.0.bias'], p['shared_fcs.1.weight'],
                           p['shared_fcs.1.bias'], p['fc_cls.weight'], p['fc_cls.bias'], p['fc_reg.weight'],
                           p['fc_reg.bias'])
    assert relerr(cls, d['cls']) <= TOL and relerr(reg, d['reg']) <= TOL
    ((cls * d['cot_cls'].to(dev)).sum() + (reg * d['cot_reg'].to(dev)).sum()).backward()
    assert relerr(x.grad, d['dx']) <= TOL_GRAD
    for k, v in p.items():
        assert relerr(v.grad, d['dp.' + k]) <= TOL_GRAD, k


def test_end_to_end_golden_through_plugin_surface(dev):
    """Registry-built modules with the reference's state_dict keys: neck -> 3-RoI extractor -> head."""
    import arfe_amd
    d = load_golden('e2e_small.npz')
    C = d['x0'].shape[1]
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=C, num_levels=5)).to(dev)
    roi_head = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=C, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=24,
                       roi_feat_size=7, num_classes=80))).to(dev)
    neck.load_state_dict({k[5:]: v for k, v in d.items() if k.startswith('neck.')})
    roi_head.bbox_head.load_state_dict({k[5:]: v for k, v in d.items() if k.startswith('head.')})
    xs = [d[f'x{i}'].to(dev).requires_grad_(True) for i in range(5)]
    feats = neck(xs)
    for i in range(5):
        assert relerr(feats[i], d[f'feat{i}']) <= TOL, i
    res = roi_head._bbox_forward(feats, d['rois'].to(dev))
    assert relerr(res['bbox_feats'], d['bbox_feats']) <= TOL
    assert relerr(res['cls_score'], d['cls']) <= TOL and relerr(res['bbox_pred'], d['reg']) <= TOL
    ((res['cls_score'] * d['cot_cls'].to(dev)).sum() + (res['bbox_pred'] * d['cot_reg'].to(dev)).sum()).backward()
    for i in range(5):
        assert relerr(xs[i].grad, d[f'dx{i}']) <= TOL_GRAD, i
    for k, v in neck.named_parameters():
        if k == 'refine.phi.conv.bias':
            continue
        assert relerr(v.grad, d['dneck.' + k]) <= TOL_GRAD, k
    for k, v in roi_head.bbox_head.named_parameters():
        assert relerr(v.grad, d['dhead.' + k]) <= TOL_GRAD, k


def test_end_to_end_medium_matches_oracle(O, dev):
    """A mid-size configuration between the golden toys and the bench (B=3, C=64, odd pyramid, R=330, RoI count not a
    multiple of 4/32): other tile / split-K / per-tap-K planner branches than either; whole path against the oracle."""
    import arfe_amd
    g = torch.Generator().manual_seed(41)
    B, C, fc = 3, 64, 96
    sizes = ((100, 152), (50, 76), (25, 38), (13, 19), (7, 10))
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=C, num_levels=5)).to(dev)
    roi_head = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=C, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=fc,
                       roi_feat_size=7, num_classes=80))).to(dev)
    pn = O.arfpn_params(C, 5, seed=5)
    ph = O.rff_params(C, fc, 80, 7, seed=6)
    for d in (pn, ph):
        for k in d:
            if k.endswith('bias'):
                d[k] = torch.randn(d[k].shape, generator=g) * 0.1
    neck.load_state_dict({k: v.clone() for k, v in pn.items()})
    roi_head.bbox_head.load_state_dict({k: v.clone() for k, v in ph.items()})
    xs_cpu = [torch.randn(B, C, h, w, generator=g) for h, w in sizes]
    rois = O.synth_rois(B, 110, img_w=608, img_h=400, seed=9)
    from oracle import parity as P
    cot_c, cot_r = torch.randn(rois.shape[0], 81, generator=g), torch.randn(rois.shape[0], 320, generator=g)
    xs = [x.clone().to(dev).requires_grad_(True) for x in xs_cpu]
    got = P.device_step(neck, roi_head, xs, rois.to(dev), cot_c.to(dev), cot_r.to(dev))
    rep = P.compare(got, xs_cpu, rois, pn, ph, cot_c, cot_r)
    print(rep)
    # forward in max-norm; the ReLU / argmax choices that differ are counted; gradients in max-norm (north_star's bound) against
    # the oracle evaluated under the device's choices - no L2 allowance for flipped masks any more
    assert rep['max_rel_fwd'] <= TOL, rep
    assert rep['mask_flips'] <= max(3, 20e-6 * rep['decisions']), rep
    assert rep['max_rel_grad_agreeing'] <= NORTH_STAR_TOL, rep


def test_bench_workload_full_size_matches_oracle(O, dev):
    """The exact bench.py step (BASELINE configs[2] hot path: B=2, C=256, 1333x800 pyramid, R=1024, seeds 1234 / 99,
    init_weights parameters) on the HIP path against the oracle, as a whole: cls/reg in max-norm against the free-running
    oracle; the ReLU / argmax choices that differ between the two forward passes COUNTED; every gradient in max-norm
    (north_star's 1e-3) against the oracle evaluated under the device's choices (oracle/parity.py)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import parity as P
    B, per_img = 2, 512
    neck, roi_head = bench.build_modules(dev)
    pn = {k: v.detach().cpu().clone() for k, v in neck.state_dict().items()}
    ph = {k: v.detach().cpu().clone() for k, v in roi_head.bbox_head.state_dict().items()}
    xs_cpu = O.synth_pyramid(B, 256, seed=1234)
    rois = O.synth_rois(B, per_img, seed=1234)
    g = torch.Generator().manual_seed(99)
    cc, cr = torch.randn(B * per_img, 81, generator=g), torch.randn(B * per_img, 320, generator=g)
    xs = [x.to(dev).requires_grad_(True) for x in xs_cpu]
    got = P.device_step(neck, roi_head, xs, rois.to(dev), cc.to(dev), cr.to(dev))
    rep = P.compare(got, xs_cpu, rois, pn, ph, cc, cr)
    print(rep)
    assert rep['max_rel_fwd'] <= TOL, rep
    # two fp32 evaluations disagree on a handful of choices per million (pre-activations within rounding of zero)
    assert rep['mask_flips'] <= 20e-6 * rep['decisions'], rep
    assert rep['max_rel_grad_agreeing'] <= NORTH_STAR_TOL, rep


def test_full_size_properties(Fn, O, dev):
    """BASELINE-size (B=2) properties that need no CPU reference run:
    scatter with zero gates is the identity; the path is linear in the upstream cotangent; empty RoI sets."""
    xs = [x.to(dev) for x in O.synth_pyramid(2, 256, seed=77)]
    bsf = torch.randn(2, 256, 50, 84, device=dev)
    zeros = [torch.zeros(2, 1, *x.shape[2:], device=dev) for x in xs]
    outs = Fn.fpn_scatter_op(xs, bsf, zeros, zeros, 2)
    assert all(torch.equal(o, x) for o, x in zip(outs, xs))
    ones = [torch.full((2, 1, *x.shape[2:]), 0.5, device=dev) for x in xs]
    outs = Fn.fpn_scatter_op(xs, bsf, ones, ones, 2)       # a + b == 1 -> out - x == nearest(bsf)
    up = outs[0] - xs[0]
    assert relerr(up[:, :, ::4, ::4], bsf) <= 1e-6
    # backward linearity: dbsf(2g) == 2 dbsf(g)
    g = [torch.randn_like(x) for x in xs]
    _, _, d1 = Fn.fpn_scatter_bwd_op(g, bsf, ones, ones, 2)
    _, _, d2 = Fn.fpn_scatter_bwd_op([2 * t for t in g], bsf, ones, ones, 2)
    assert relerr(d2, 2 * d1) <= 1e-6
    # empty RoI set
    out = Fn.roi_extract(xs, [torch.zeros(0, 5, device=dev)], (4, 8, 16, 32, 64))
    assert out.shape == (0, 256, 7, 7)


@pytest.mark.parametrize('R', [1024, 1000])
def test_conv3x3_bench_size_matches_torch_cpu(Fn, dev, R):
    """The 3x3 conv GEMMs at the bench size (R=1024: N = 50 176 = 224 x 224 takes the 256x224 tile; R=1000 the 256x256
    one): forward, data gradient and weight gradient against torch-CPU fp32 conv2d (tolerances of the bf16x3 split)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    C = 256
    x = torch.randn(R, C, 7, 7, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.03
    b = torch.randn(C, generator=g) * 0.1
    dy = torch.randn(R, C, 7, 7, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, padding=1)
    y_ref.backward(dy)
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    pk = Fn.conv3x3_pack_op(wd)
    y = Fn.conv3x3_fwd_op(xd.data_ptr(), C * 49, R, C, pk, bd, False)
    assert relerr(y, y_ref.detach()) <= TOL
    dx = torch.empty_like(xd)
    Fn.conv3x3_bwd_data_op(R, C, pk, dyd, dx.data_ptr(), C * 49)
    assert relerr(dx, xr.grad) <= TOL_GRAD
    import ctypes
    from arfe_amd import hip
    ws = torch.empty(int(hip.lib().arfe_conv3x3_workspace_bytes(C)), dtype=torch.uint8, device=dev)
    dw, db = Fn.conv3x3_bwd_weight_op(xd.data_ptr(), C * 49, R, C, wd, dyd, ws)
    assert relerr(dw, wr.grad) <= TOL_GRAD
    assert relerr(db, br.grad) <= TOL_GRAD


@pytest.mark.parametrize('R,C', [(32, 64), (96, 96), (64, 320), (50, 64), (5, 96), (131, 64), (515, 32)])
def test_conv3x3_stage_once_kernel_matches_torch_cpu(Fn, dev, R, C):
    """The stage-once conv kernel (conv_stage_once.hpp; taken when R*49 >= 224 and C is a multiple of 32, a ragged last
    tile included: R = 50 and R = 5): 224-pixel tiles cut RoIs, so the 8-pixel halo, the zero row of padding lanes, ReLU + bias, a
    strided channel-slice input (the wh/hh convs read slices of the [R,3C,7,7] concat) and a second M tile (C = 320) are
    all exercised; forward and data gradient against torch-CPU fp32 conv2d."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(50 + R)
    xcat = torch.randn(R, 3 * C, 7, 7, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    b = torch.randn(C, generator=g) * 0.1
    dy = torch.randn(R, C, 7, 7, generator=g)
    xs = xcat[:, C:2 * C].clone().requires_grad_(True)          # the middle slice of the concat
    wr = w.clone().requires_grad_(True)
    y_ref = F.relu(F.conv2d(xs, wr, b, padding=1))
    y_ref.backward(dy)
    xd, wd, bd, dyd = xcat.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    pk = Fn.conv3x3_pack_op(wd)
    y = Fn.conv3x3_fwd_op(xd.data_ptr() + 4 * C * 49, 3 * C * 49, R, C, pk, bd, True)
    assert relerr(y, y_ref.detach()) <= TOL
    # data gradient of the un-activated conv (ReLU backward is the producer's job): dx = conv^T(dy * [y > 0])
    # (the REFERENCE's activation mask: an output within rounding of zero may flip sign between the two - one such element
    # in 1 M with ARFE_GEMM_F32=1 at (64, 320) - and a flipped mask element is a 3 % error in max-norm, not a kernel error)
    dym = dyd * (y_ref.detach() > 0).to(dev)
    dx = torch.full_like(xd, float('nan'))
    Fn.conv3x3_bwd_data_op(R, C, pk, dym.contiguous(), dx.data_ptr() + 4 * C * 49, 3 * C * 49)
    assert relerr(dx[:, C:2 * C], xs.grad) <= TOL_GRAD
    assert torch.isnan(dx[:, :C]).all() and torch.isnan(dx[:, 2 * C:]).all()      # neighbouring slices untouched


@pytest.mark.parametrize('R,C', [(50, 64), (5, 96), (3, 32)])
def test_conv3x3_fwd_fuse_equals_conv_then_fusion(Fn, dev, R, C):
    """arfe_conv3x3_fwd_fuse (the fusion z = ori + ori (lw + lh) in the epilogue of the conv that produces lh) against the two
    calls it stands for, on a channel slice of a [R, 3C, 7, 7] concat; (3, 32) takes the fallback inside the entry point."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(R, 3 * C, 7, 7, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
    b = (torch.randn(C, generator=g) * 0.1).to(dev)
    lw = torch.relu(torch.randn(R, C, 7, 7, generator=g)).to(dev)
    pk = Fn.conv3x3_pack_op(w)
    rs = 3 * C * 49
    lh_ref = Fn.conv3x3_fwd_op(x.data_ptr() + 4 * 2 * C * 49, rs, R, C, pk, b, True)
    z_ref = torch.empty(R, C, 7, 7, device=dev)
    from arfe_amd import hip
    hip.check(hip.lib().arfe_rff_fuse_fwd(R, C, x.data_ptr(), rs, hip.ptr(lw), hip.ptr(lh_ref), hip.ptr(z_ref), hip.stream()), "fuse")
    lh, z = Fn.conv3x3_fwd_fuse_op(x.data_ptr() + 4 * 2 * C * 49, rs, R, C, pk, b, x.data_ptr(), rs, lw)
    assert torch.equal(lh, lh_ref)
    assert (z - z_ref).abs().max().item() <= 1e-6 * max(1.0, z_ref.abs().max().item())


@pytest.mark.parametrize('R,C', [(64, 128), (65, 128), (66, 128), (67, 128), (200, 256), (515, 128)])
def test_conv3x3_wgrad_stage_once_kernel_matches_torch_cpu(Fn, dev, R, C):
    """The stage-once weight-gradient kernel (conv_wgrad_stage_once.hpp; C a multiple of 128, R >= 64): row slices of a
    single K-step, a last slice that ends inside a K-step (R = 65: 455 pixel rows), the row halo across slice and RoI
    boundaries, the three column-shifted copies, and a strided x slice of the [R,3C,7,7] concat; dw and db against
    torch-CPU fp32 conv2d."""
    import torch.nn.functional as F
    from arfe_amd import hip
    g = torch.Generator().manual_seed(70 + R)
    xcat = torch.randn(R, 3 * C, 7, 7, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    b = torch.zeros(C)
    dy = torch.randn(R, C, 7, 7, generator=g)
    xs = xcat[:, 2 * C:].clone()
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.conv2d(xs, wr, br, padding=1).backward(dy)
    xd, wd, dyd = xcat.to(dev), w.to(dev), dy.to(dev)
    ws = torch.empty(int(hip.lib().arfe_conv3x3_workspace_bytes(C)), dtype=torch.uint8, device=dev)
    dw, db = Fn.conv3x3_bwd_weight_op(xd.data_ptr() + 4 * 2 * C * 49, 3 * C * 49, R, C, wd, dyd, ws)
    assert relerr(dw, wr.grad) <= TOL_GRAD
    assert relerr(db, br.grad) <= TOL_GRAD


def _run_fpn(dev, xs_cpu, params_cpu, num_outs, cots_cpu):
    import arfe_amd
    in_channels = [x.shape[1] for x in xs_cpu]
    out_channels = params_cpu['lateral_convs.0.conv.weight'].shape[0]
    neck = arfe_amd.build_neck(dict(type='FPN', in_channels=in_channels, out_channels=out_channels, num_outs=num_outs)).to(dev)
    neck.load_state_dict({k: v.clone() for k, v in params_cpu.items()})
    xs = [x.clone().to(dev).requires_grad_(True) for x in xs_cpu]
    outs = neck(xs)
    sum((o * c.to(dev)).sum() for o, c in zip(outs, cots_cpu)).backward()
    return neck, xs, outs


@pytest.mark.parametrize('name', ['fpn_tiny.npz', 'fpn_ragged.npz'])
def test_base_fpn_golden_fwd_bwd(dev, name):
    """registry-built FPN on the HIP kernels against the reference's own outputs and gradients."""
    d = load_golden(name)
    n = sum(1 for k in d if k[0] == 'x' and k[1:].isdigit())
    num_outs = int(d['num_outs'])
    xs_cpu = [d[f'x{i}'] for i in range(n)]
    params = {k[2:]: v for k, v in d.items() if k.startswith('p.')}
    cots = [d[f'cot{i}'] for i in range(num_outs)]
    neck, xs, outs = _run_fpn(dev, xs_cpu, params, num_outs, cots)
    for i, o in enumerate(outs):
        assert o.shape == d[f'out{i}'].shape
        assert relerr(o, d[f'out{i}']) <= TOL, i
    for i in range(n):
        assert relerr(xs[i].grad, d[f'dx{i}']) <= TOL_GRAD, i
    for k, v in neck.named_parameters():
        assert relerr(v.grad, d['dp.' + k]) <= TOL_GRAD, k


def test_base_fpn_r50_sizes_match_oracle(O, dev):
    """ResNet-50 channel widths (256..2048 -> 256) at a quarter of the 1333x800 pyramid, B=2: large-tile GEMM paths of
    the 1x1 / 3x3 map convolutions against the CPU oracle."""
    g = torch.Generator().manual_seed(21)
    in_channels, sizes = (256, 512, 1024, 2048), ((50, 84), (25, 42), (13, 21), (7, 11))
    xs_cpu = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(in_channels, sizes)]
    params = O.fpn_params(in_channels, 256, seed=3)
    for k in params:
        if k.endswith('bias'):
            params[k] = torch.randn(params[k].shape, generator=g) * 0.1
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xr = [x.clone().requires_grad_(True) for x in xs_cpu]
    ref = O.fpn_forward(xr, pr, 5)
    cots = [torch.randn(o.shape, generator=g) for o in ref]
    sum((o * c).sum() for o, c in zip(ref, cots)).backward()
    neck, xs, outs = _run_fpn(dev, xs_cpu, params, 5, cots)
    for i, o in enumerate(outs):
        assert relerr(o, ref[i].detach()) <= TOL, i
    for i in range(len(xs)):
        assert relerr(xs[i].grad, xr[i].grad) <= TOL_GRAD, i
    for k, v in neck.named_parameters():
        assert relerr(v.grad, pr[k].grad) <= TOL_GRAD, k


def test_neck_sequence_fpn_then_arfpn_matches_oracle(O, dev):
    """The flagship config's neck list [FPN, WFPNDualSpatial] built through the registry (nn.Sequential, builder.py:14-18):
    backbone-shaped inputs -> base FPN -> AR-FPN, forward and gradients down to the backbone features.

    The forward is compared end to end with the composition of the two oracles.  The gradients are compared stage by
    stage on the GPU's own intermediates (oracle AR-FPN evaluated AT the GPU pyramid, oracle FPN backward fed the GPU's
    pyramid gradient): AR-FPN is not smooth - adaptive-max argmax ties and the ReLU of the gate pre-activations - and
    the ~1e-5 relative difference of the bf16x3 FPN output from the CPU's flips about one such decision in every tenth
    random draw (measured with tools/dbg_neck_seq.py), which moves a 3x3xC patch (or a pair of pixels) of the gradient
    by its full value although every kernel is right.  With identical inputs on both sides the decisions agree."""
    import arfe_amd
    torch.manual_seed(31)
    torch.cuda.manual_seed(31)
    g = torch.Generator().manual_seed(31)
    in_channels, sizes = (8, 16, 24, 32), ((40, 64), (20, 32), (10, 16), (5, 8))
    neck = arfe_amd.build_neck([dict(type='FPN', in_channels=list(in_channels), out_channels=16, num_outs=5),
                                dict(type='WFPNDualSpatial', in_channels=16, num_levels=5)]).to(dev)
    assert isinstance(neck, torch.nn.Sequential)
    for m in neck:
        m.init_weights()
    with torch.no_grad():
        for n, p in neck.named_parameters():
            if n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g).to(dev) * 0.1)
    xs_cpu = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(in_channels, sizes)]
    pf = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in neck[0].state_dict().items()}
    pa = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in neck[1].state_dict().items()}

    # forward, end to end through the Sequential
    ref = O.arfpn_forward(list(O.fpn_forward(xs_cpu, pf, 5)), pa, refine_level=2)
    cots = [torch.randn(o.shape, generator=g) for o in ref]
    with torch.no_grad():
        outs_seq = neck([x.to(dev) for x in xs_cpu])
    for i, o in enumerate(outs_seq):
        assert relerr(o, ref[i].detach()) <= TOL, i

    # backward through both modules in one autograd graph
    xs = [x.clone().to(dev).requires_grad_(True) for x in xs_cpu]
    pyr = neck[0](xs)
    for t in pyr:
        t.retain_grad()
    outs = neck[1](pyr)
    for o, s in zip(outs, outs_seq):
        assert relerr(o, s) <= 1e-6       # bit-equal in the default mode; ARFE_GEMM_F32=1 still sums its K slices with atomics
    sum((o * c.to(dev)).sum() for o, c in zip(outs, cots)).backward()

    # stage 2 (AR-FPN) at the GPU pyramid
    # (the 5th level is FPN's stride-2 subsample of the 4th, a copy: kept in the oracle graph so that its gradient
    # flows into level 3 as it does in the product's graph)
    pyr_at = [t.detach().cpu().clone().requires_grad_(True) for t in pyr[:4]]
    pyr_at.append(torch.nn.functional.max_pool2d(pyr_at[3], 1, stride=2))
    assert torch.equal(pyr_at[4].detach(), pyr[4].detach().cpu())
    pyr_at[4].retain_grad()
    sum((o * c).sum() for o, c in zip(O.arfpn_forward(pyr_at, pa, refine_level=2), cots)).backward()
    for i in range(5):
        assert relerr(pyr[i].grad, pyr_at[i].grad) <= TOL_GRAD, i
    for k, v in neck[1].named_parameters():
        if k != 'refine.phi.conv.bias':
            assert relerr(v.grad, pa[k].grad) <= TOL_GRAD, k
    # stage 1 (FPN) fed the GPU's pyramid gradient
    # (level 3's retained gradient already holds the share that arrived through the subsampled level 4: take it out,
    # the oracle graph below routes it again)
    ups = [t.grad.cpu().clone() for t in pyr]
    ups[3][:, :, ::2, ::2] -= ups[4]
    xr = [x.clone().requires_grad_(True) for x in xs_cpu]
    sum((o * u).sum() for o, u in zip(O.fpn_forward(xr, pf, 5), ups)).backward()
    for i in range(len(xs)):
        assert relerr(xs[i].grad, xr[i].grad) <= TOL_GRAD, i
    for k, v in neck[0].named_parameters():
        assert relerr(v.grad, pf[k].grad) <= TOL_GRAD, k


def test_cascade_roi_head_stages_match_oracle(O, dev):
    """CascadeRoIHead._bbox_forward(stage, x, rois) (cascade_roi_head.py:126-149, intended AR-RFF body): three stages with
    their own weights on the e2e fixture's pyramid features and RoIs, each against the oracle, forward and gradients."""
    import arfe_amd
    d = load_golden('e2e_small.npz')
    C = d['x0'].shape[1]
    ext = dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0), out_channels=C,
               featmap_strides=[4, 8, 16, 32, 64])
    head = dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=24, roi_feat_size=7,
                num_classes=80)
    rh = arfe_amd.build_head(dict(type='CascadeRoIHead', num_stages=3, stage_loss_weights=[1, 0.5, 0.25],
                                  bbox_roi_extractor=ext, bbox_head=head)).to(dev)
    g = torch.Generator().manual_seed(9)
    rh.bbox_head[0].load_state_dict({k[5:]: v for k, v in d.items() if k.startswith('head.')})
    for st in (1, 2):                                   # the other stages: the same shapes, their own values
        rh.bbox_head[st].load_state_dict({k[5:]: v + 0.05 * st * torch.randn(v.shape, generator=g)
                                          for k, v in d.items() if k.startswith('head.')})
    feats_cpu = [d[f'feat{i}'] for i in range(5)]
    rois = d['rois']
    for st in range(3):
        feats = [f.to(dev).requires_grad_(True) for f in feats_cpu]
        # later stages see shifted boxes (the reference refines them between stages; any boxes exercise the path)
        r = rois.clone()
        r[:, 1:] += 3.0 * st
        res = rh._bbox_forward(st, feats, r.to(dev))
        fr = [f.clone().requires_grad_(True) for f in feats_cpu]
        ph = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in rh.bbox_head[st].state_dict().items()}
        bf = O.roi_extract3(fr, r)
        cls, reg = O.rff_head(bf, ph)
        assert relerr(res['bbox_feats'], bf.detach()) <= TOL, st
        assert relerr(res['cls_score'], cls.detach()) <= TOL and relerr(res['bbox_pred'], reg.detach()) <= TOL, st
        c1, c2 = torch.randn(cls.shape, generator=g), torch.randn(reg.shape, generator=g)
        ((cls * c1).sum() + (reg * c2).sum()).backward()
        ((res['cls_score'] * c1.to(dev)).sum() + (res['bbox_pred'] * c2.to(dev)).sum()).backward()
        for i in range(5):
            ref_g = fr[i].grad if fr[i].grad is not None else torch.zeros_like(fr[i])    # a level no RoI maps to
            assert relerr(feats[i].grad, ref_g) <= TOL_GRAD, (st, i)
        for k, v in rh.bbox_head[st].named_parameters():
            assert relerr(v.grad, ph[k].grad) <= TOL_GRAD, (st, k)
    outs = rh.forward_dummy([f.to(dev) for f in feats_cpu], rois[:12, 1:].to(dev))
    assert len(outs) == 6 and outs[0].shape == (12, 81) and outs[5].shape == (12, 320)


def test_extractor_fork_options_match_oracle(O, dev):
    """single_level.py:110-141, the fork's extra arguments: `replace_rois` (levels from other boxes), `lvl` (levels shifted
    and clamped) and `roi_scale_factor` (boxes rescaled after the levels were chosen), each against the oracle's RoIAlign on
    the level the reference would pick."""
    import arfe_amd
    C = 6
    g = torch.Generator().manual_seed(23)
    feats = [torch.randn(2, C, h, w, generator=g) for h, w in PYRAMIDS['mid']]
    strides = [4, 8, 16, 32, 64]
    ext = arfe_amd.build_roi_extractor(dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                            out_channels=C, featmap_strides=strides))
    rois = O.synth_rois(2, 20, img_w=384, img_h=256, seed=4)
    other = O.synth_rois(2, 20, img_w=384, img_h=256, seed=5)
    fd = [f.to(dev) for f in feats]

    def ref(boxes, levels):
        out = torch.zeros(boxes.shape[0], C, 7, 7)
        for i in range(5):
            idx = (levels == i).nonzero(as_tuple=True)[0]
            if idx.numel():
                out[idx] = O.roi_align_fwd(feats[i], boxes[idx], 1.0 / strides[i])
        return out

    lv = O.map_roi_levels(rois, 5)
    for kw, boxes, levels in (
            (dict(lvl=1), rois, (lv + 1).clamp(0, 4)),
            (dict(lvl=-2), rois, (lv - 2).clamp(0, 4)),
            (dict(replace_rois=other.to(dev)), rois, O.map_roi_levels(other, 5)),
            (dict(roi_scale_factor=1.5), ext.roi_rescale(rois, 1.5), lv),
            (dict(roi_scale_factor=0.7, lvl=1, replace_rois=other.to(dev)), ext.roi_rescale(rois, 0.7),
             (O.map_roi_levels(other, 5) + 1).clamp(0, 4))):
        got = ext(fd, rois.to(dev), **kw)
        assert relerr(got, ref(boxes, levels)) <= 1e-5, kw


def test_empty_roi_set_flows_through_extractor_and_head(dev):
    """R = 0 (an image batch without proposals) must give empty outputs and zero gradients, not a crash."""
    import arfe_amd
    C = 8
    xs = [torch.randn(1, C, h, w, device=dev, requires_grad=True) for h, w in PYRAMIDS['tiny']]
    roi_head = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=C, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=16,
                       roi_feat_size=7, num_classes=80))).to(dev)
    roi_head.init_weights()
    res = roi_head._bbox_forward(xs, torch.zeros(0, 5, device=dev))
    assert res['cls_score'].shape == (0, 81) and res['bbox_pred'].shape == (0, 320)
    (res['cls_score'].sum() + res['bbox_pred'].sum()).backward()
    for p in roi_head.bbox_head.parameters():
        assert p.grad is not None and p.grad.abs().max().item() == 0.0
    for x in xs:
        assert x.grad is None or x.grad.abs().max().item() == 0.0


def test_exact_fp32_mode_in_a_fresh_process():
    """ARFE_GEMM_F32=1 (read once per process) routes every contraction through the exact-fp32 MFMA kernel:
    the head golden case must then match to summation-order accuracy."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = '''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import load_golden, relerr, relerr_l2
from arfe_amd import functional as Fn
dev = torch.device("cuda:0")
d = load_golden("rff_head_r40.npz")
x = d["x"].to(dev)
p = {k[2:]: v.to(dev) for k, v in d.items() if k.startswith("p.")}
cls, reg = Fn.rff_head(x, p["wh_conv.conv.weight"], p["wh_conv.conv.bias"], p["hh_conv.conv.weight"], p["hh_conv.conv.bias"],
                       p["final_conv.conv.weight"], p["final_conv.conv.bias"], p["shared_fcs.0.weight"], p["shared_fcs.0.bias"],
                       p["shared_fcs.1.weight"], p["shared_fcs.1.bias"], p["fc_cls.weight"], p["fc_cls.bias"],
                       p["fc_reg.weight"], p["fc_reg.bias"])
e = max(relerr(cls, d["cls"]), relerr(reg, d["reg"]))
print("ERR", e)
assert e <= 2e-6, e
''' % (ROOT, ROOT)
    env = dict(os.environ, ARFE_GEMM_F32='1')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize('R', [1024, 96])
def test_half_operand_mode_is_exact_for_half_values(Fn, dev, R):
    """arfe_set_operand_precision(1) - the contraction arithmetic of the reference's fp16 mode: operands go to the matrix pipe as
    IEEE halves, ONE f16 MFMA per product, fp32 accumulation.  For tensors whose values are half-representable every product
    is exact, so 3x3 conv (stage-once kernel at R = 1024, generic GEMM kernel at R = 96), its two gradients and a linear layer must
    match an fp64 evaluation to fp32 accumulation error; the mode is per thread and off again after the block."""
    import torch.nn.functional as F
    from arfe_amd import hip
    g = torch.Generator().manual_seed(31)
    C = 128
    h = lambda t: t.half().float()
    x = h(torch.randn(R, C, 7, 7, generator=g))
    w = h(torch.randn(C, C, 3, 3, generator=g) * 0.05)
    b = h(torch.randn(C, generator=g) * 0.1)
    dy = h(torch.randn(R, C, 7, 7, generator=g))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b.double(), padding=1)
    y_ref.backward(dy.double())
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    assert hip.lib().arfe_get_operand_precision() == 0
    with Fn.lib_precision(True):
        assert hip.lib().arfe_get_operand_precision() == 1
        pk = Fn.conv3x3_pack_op(wd)
        y = Fn.conv3x3_fwd_op(xd.data_ptr(), C * 49, R, C, pk, bd, False)
        dx = torch.empty_like(xd)
        Fn.conv3x3_bwd_data_op(R, C, pk, dyd, dx.data_ptr(), C * 49)
        ws = torch.empty(int(hip.lib().arfe_conv3x3_workspace_bytes(C)), dtype=torch.uint8, device=dev)
        dw, db = Fn.conv3x3_bwd_weight_op(xd.data_ptr(), C * 49, R, C, wd, dyd, ws)
        lw = h(torch.randn(200, C * 49, generator=g) * 0.02)
        lb = h(torch.randn(200, generator=g))
        ly = Fn.linear_fwd_op(xd.view(R, C * 49), lw.to(dev), lb.to(dev), False)
    assert hip.lib().arfe_get_operand_precision() == 0
    assert relerr(y, y_ref.detach().float()) <= 2e-6
    assert relerr(dx, xr.grad.float()) <= 2e-6
    assert relerr(dw, wr.grad.float()) <= 5e-6
    assert relerr(ly, F.linear(x.double().view(R, -1), lw.double(), lb.double()).float()) <= 2e-6
    # the default mode on the same inputs agrees to its own (bf16x3) accuracy
    y3 = Fn.conv3x3_fwd_op(xd.data_ptr(), C * 49, R, C, pk, bd, False)
    assert relerr(y3, y_ref.detach().float()) <= TOL


@pytest.mark.parametrize('shape,C', [('mid', 12), ('bench', 8), ('tiny', 8), ('ragged', 8)])
def test_arfpn_half_storage_equals_the_conversion_route(Fn, O, dev, shape, C):
    """fp16 mode of the neck: on eligible pyramids the streaming kernels read / write the half pyramid themselves
    (arfe_pyramid_t::storage = F16).  The arithmetic is the fp32 one, so outputs and input gradients must equal, bit for bit,
    the round-2 route (half -> float pass, fp32 kernels, float -> half pass); parameter gradients (fp32 either way) too."""
    sizes = O.PYRAMID_1333x800 if shape == 'bench' else PYRAMIDS[shape]
    g = torch.Generator().manual_seed(13)
    xs_h = [torch.randn(2, C, h, w, generator=g).half().to(dev) for h, w in sizes]
    assert Fn.fpn_half_storage_supported(xs_h, 2) == (shape != 'ragged')     # 'ragged': pooled levels that are not 4x / 2x the refine level
    p = {k: v.half().float().to(dev) for k, v in O.arfpn_params(C, 5, seed=3).items()}
    cots = [torch.randn(2, C, h, w, generator=g).half().to(dev) for h, w in sizes]

    def run(native):
        xs = [x.clone().requires_grad_(True) for x in xs_h]
        pp = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        w1 = [pp[f'reduce_convs.{i}.conv.weight'] for i in range(5)]
        b1 = [pp[f'reduce_convs.{i}.conv.bias'] for i in range(5)]
        w2 = [pp[f'reduce_convs2.{i}.conv.weight'] for i in range(5)]
        b2 = [pp[f'reduce_convs2.{i}.conv.bias'] for i in range(5)]
        nl = []
        for n in ('theta', 'phi', 'g', 'conv_out'):
            nl += [pp[f'refine.{n}.conv.weight'].view(C, C), pp[f'refine.{n}.conv.bias']]
        if native:
            outs = Fn.arfpn(xs, w1, b1, w2, b2, *nl, refine_level=2, half_ops=True)
        else:
            outs = [Fn.to_f16(o) for o in Fn.arfpn([Fn.to_f32(x) for x in xs], w1, b1, w2, b2, *nl, refine_level=2, half_ops=True)]
        assert all(o.dtype == torch.float16 for o in outs)
        torch.autograd.backward(list(outs), cots)
        return outs, [x.grad for x in xs], {k: v.grad for k, v in pp.items()}

    if shape == 'ragged':
        with pytest.raises(Exception):
            run(True)                     # not eligible: the library refuses half storage (the module converts instead)
        return
    o1, dx1, dp1 = run(True)
    o2, dx2, dp2 = run(False)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    for a, b in zip(dx1, dx2):
        assert a.dtype == torch.float16 and torch.equal(a, b)
    for k in dp1:
        if k == 'refine.phi.conv.bias':
            continue
        # (the gate-conv weight gradients are summed with float atomics: equal up to their order)
        assert relerr(dp1[k], dp2[k]) <= 1e-5, k


def test_cast_kernels_match_tensor_to(Fn, dev):
    """arfe_cast_*: round to nearest even, inf/nan/subnormals, unaligned views and ragged tails like Tensor.to."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100003, generator=g) * torch.exp(torch.randn(100003, generator=g) * 6)
    x[:6] = torch.tensor([float('inf'), -float('inf'), float('nan'), 65520.0, 1e-8, -6.1e-5])
    xd = x.to(dev)
    for off in (0, 1, 3):
        h = Fn.to_f16(xd[off:])
        ref = x[off:].half()
        assert h.dtype == torch.float16 and torch.equal(h.cpu().view(torch.int16), ref.view(torch.int16)), off
        f = Fn.to_f32(h)
        assert f.dtype == torch.float32 and torch.equal(f.cpu().view(torch.int32), ref.float().view(torch.int32)), off
    t = xd[:1000].clone().requires_grad_(True)
    Fn.to_f32(Fn.to_f16(t)).sum().backward()
    assert t.grad.dtype == torch.float32 and torch.equal(t.grad.cpu(), torch.ones(1000))
    assert Fn.to_f16(torch.empty(0, device=dev)).numel() == 0


@pytest.mark.parametrize('products', ['half', 'f32'])
def test_fp16_mode_through_plugin_surface(O, dev, products, monkeypatch):
    """wrap_fp16_model on registry-built FPN -> WFPNDualSpatial -> StandardRoIHead: half parameters and activations at
    the module boundaries (dtypes as in the reference), compared (a) tightly with the oracle's restatement of the same
    storage contract and (b) with the fixture the reference produced in its own fp16 mode, within the bounds of
    conftest (the reference's half kernels are the noisier side).
    products: 'half' (default) - the contractions take half operands, one f16 MFMA per product as in the reference's half
    kernels (intermediates that feed a contraction are rounded to half, so gradients sit ~1 % from the fp32 evaluation - still
    2-5x closer than the reference's own half run); 'f32' (ARFE_FP16_F32_PRODUCTS=1) - fp32-grade products, round 2's form."""
    import arfe_amd
    monkeypatch.setenv('ARFE_FP16_F32_PRODUCTS', '1' if products == 'f32' else '0')
    GRAD_A = 5e-3 if products == 'f32' else 2.5e-2
    from conftest import FP16_FWD_TOL, check_fp16_grads, fp16_fixture
    d, xs_cpu, P = fp16_fixture()
    C = 8
    fpn = arfe_amd.build_neck(dict(type='FPN', in_channels=[4, 6, 8, 10], out_channels=C, num_outs=5))
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=C, num_levels=5))
    roi_head = arfe_amd.build_head(dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0),
                                out_channels=C, featmap_strides=[4, 8, 16, 32, 64]),
        bbox_head=dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=24,
                       roi_feat_size=7, num_classes=80)))
    fpn.load_state_dict({k: v.detach() for k, v in P['fpn.'].items()})
    neck.load_state_dict({k: v.detach() for k, v in P['neck.'].items()})
    roi_head.bbox_head.load_state_dict({k: v.detach() for k, v in P['head.'].items()})
    mods = [fpn.to(dev), neck.to(dev), roi_head.to(dev)]
    for m in mods:
        arfe_amd.wrap_fp16_model(m)
    assert all(p.dtype == torch.float16 for m in mods for p in m.parameters())
    assert fpn.fp16_enabled and roi_head.bbox_roi_extractor.fp16_enabled and roi_head.bbox_head.fp16_enabled
    xs = [x.detach().to(dev).requires_grad_(True) for x in xs_cpu]
    from arfe_amd import functional as FnT
    from oracle import parity as PT
    tap = {}
    FnT.tap_decisions(tap)
    try:
        pyr = fpn(xs)                                        # fp32 in, half out (@auto_fp16)
        feats = neck(pyr)
        res = roi_head._bbox_forward(feats, d['rois'].to(dev))
    finally:
        FnT.tap_decisions(None)
    dz = PT.decisions_from_tap(tap, 2)
    cls, reg, bbox_feats = res['cls_score'], res['bbox_pred'], res['bbox_feats']
    assert all(t.dtype == torch.float16 for t in (*pyr, *feats, bbox_feats, cls, reg))
    ((cls * d['cot_cls'].to(dev)).sum() + (reg * d['cot_reg'].to(dev)).sum()).backward()
    grads = {'fpn.': {k: v.grad for k, v in fpn.named_parameters()},
             'neck.': {k: v.grad for k, v in neck.named_parameters()},
             'head.': {k: v.grad for k, v in roi_head.bbox_head.named_parameters()}}
    assert all(g is None or g.dtype == torch.float16 for gs in grads.values() for g in gs.values())
    assert all(x.grad.dtype == torch.float32 for x in xs)

    # (a) same storage contract, fp32 oracle evaluated under the device's ReLU / argmax choices (a ReLU that opens on one
    # side only moves every gradient of this small fixture by percents: oracle/arfe_oracle.Decisions): differences are
    # single half-ulp rounding flips (2^-11 relative)
    o_pyr, o_feats, o_bbox, o_cls, o_reg = O.fp16_mode_forward(xs_cpu, P['fpn.'], P['neck.'], P['head.'], d['rois'],
                                                               decisions=dz)
    HALF_ULP = 2e-3
    for i in range(5):
        assert relerr(pyr[i], o_pyr[i]) <= HALF_ULP and relerr(feats[i], o_feats[i]) <= HALF_ULP, i
    assert relerr(bbox_feats, o_bbox) <= HALF_ULP
    assert relerr(cls, o_cls) <= HALF_ULP and relerr(reg, o_reg) <= HALF_ULP
    ((o_cls * d['cot_cls'].float()).sum() + (o_reg * d['cot_reg'].float()).sum()).backward()
    for i in range(4):
        assert relerr_l2(xs[i].grad, xs_cpu[i].grad) <= GRAD_A, i
    for tag, gs in grads.items():
        for k, g in gs.items():
            og = P[tag][k].grad
            if og is None or k == 'refine.phi.conv.bias':
                continue
            # the product rounds each parameter gradient to half once (storage); the oracle's stays fp32.
            # NonLocal bias gradients with half operands: the softmax backward dS = P (dP - delta) keeps a row-sum residue once
            # P and dP are rounded on their way into the MFMA, and these plain sums over L^2 entries collect it.  The
            # reference's own half kernels show the same (conftest.FP16_GRAD_TOL).
            from conftest import FP16_GRAD_TOL
            tol = FP16_GRAD_TOL if (products == 'half' and k.startswith('refine.') and k.endswith('bias')) else GRAD_A
            assert relerr_l2(g, og) <= tol, (tag + k, relerr_l2(g, og))

    # (b) the reference's own half run
    for i in range(5):
        assert relerr(pyr[i], d[f'pyr{i}']) <= FP16_FWD_TOL and relerr(feats[i], d[f'feat{i}']) <= FP16_FWD_TOL, i
    assert relerr(bbox_feats, d['bbox_feats']) <= FP16_FWD_TOL
    assert relerr(cls, d['cls']) <= FP16_FWD_TOL and relerr(reg, d['reg']) <= FP16_FWD_TOL
    check_fp16_grads(d, [x.grad for x in xs], grads)


def test_cascade_roi_head_in_fp16_mode_matches_oracle(O, dev):
    """BASELINE configs[4] at test size: `[FPN, WFPNDualSpatial]` + CascadeRoIHead (3 AR-RFF stages) under wrap_fp16_model
    (mmdet/core/fp16/hooks.py:86-105; cascade_roi_head.py:126-149 per stage) against the oracle's fp16-mode restatement,
    every stage with its own head weights and shifted boxes: forward within a half ulp of the storage contract, gradients of
    all three heads and of the shared neck / FPN in L2."""
    import arfe_amd
    from conftest import fp16_fixture
    d, xs_cpu, P = fp16_fixture()
    C = 8
    fpn = arfe_amd.build_neck(dict(type='FPN', in_channels=[4, 6, 8, 10], out_channels=C, num_outs=5))
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=C, num_levels=5))
    ext = dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=7, sample_num=0), out_channels=C,
               featmap_strides=[4, 8, 16, 32, 64])
    head = dict(type='MultiRoIsBBoxHead', in_channels=C, conv_out_channels=C, fc_out_channels=24, roi_feat_size=7,
                num_classes=80, reg_class_agnostic=True)
    rh = arfe_amd.build_head(dict(type='CascadeRoIHead', num_stages=3, stage_loss_weights=[1, 0.5, 0.25],
                                  bbox_roi_extractor=ext, bbox_head=head))
    fpn.load_state_dict({k: v.detach() for k, v in P['fpn.'].items()})
    neck.load_state_dict({k: v.detach() for k, v in P['neck.'].items()})
    g = torch.Generator().manual_seed(17)
    stage_params = []
    for st in range(3):        # half-representable weights per stage (the fixture's for stage 0, perturbed copies after it)
        sp = {k: (v.detach() + 0.03 * st * torch.randn(v.shape, generator=g)).half().float() for k, v in P['head.'].items()}
        if st == 0:
            sp = {k: v.detach().clone() for k, v in P['head.'].items()}
        # class-agnostic regression (every AR cascade config): fc_reg has 4 outputs
        sp['fc_reg.weight'] = sp['fc_reg.weight'][:4].clone()
        sp['fc_reg.bias'] = sp['fc_reg.bias'][:4].clone()
        rh.bbox_head[st].load_state_dict(sp)
        stage_params.append({k: v.clone().requires_grad_(True) for k, v in sp.items()})
    mods = [fpn.to(dev), neck.to(dev), rh.to(dev)]
    for m in mods:
        arfe_amd.wrap_fp16_model(m)
    assert all(p.dtype == torch.float16 for m in mods for p in m.parameters())
    xs = [x.detach().to(dev).requires_grad_(True) for x in xs_cpu]
    from arfe_amd import functional as FnT
    from oracle import parity as PT
    tap = {}
    FnT.tap_decisions(tap)
    try:
        feats = neck(fpn(xs))
    finally:
        FnT.tap_decisions(None)
    neck_tap = dict(tap)
    rois = d['rois']
    # oracle: shared trunk once, one head per stage
    xo = [x.detach().clone().requires_grad_(True) for x in xs_cpu]
    Pf = {k: v.detach().clone().requires_grad_(True) for k, v in P['fpn.'].items()}
    Pn = {k: v.detach().clone().requires_grad_(True) for k, v in P['neck.'].items()}
    o_pyr = [O.half_store(t) for t in O.fpn_forward([O.half_store(x) for x in xo], Pf, 5)]
    # the oracle runs under the device's ReLU / argmax choices (oracle/arfe_oracle.Decisions): a mask that flips under half
    # arithmetic moves the gradients of this small fixture by percents and is not an arithmetic error
    dz_neck = O.Decisions(argmax=[neck_tap['argmax'][i].cpu().long() for i in range(2)],
                          gate_a=[t.cpu() > 0 for t in neck_tap['gate_a']], gate_b=[t.cpu() > 0 for t in neck_tap['gate_b']])
    o_feats = [O.half_store(t) for t in O.arfpn_forward(o_pyr, Pn, 2, decisions=dz_neck)]
    HALF_ULP = 2e-3
    loss, o_loss = 0., 0.
    for st in range(3):
        r = rois.clone()
        r[:, 1:] += 2.0 * st
        tap = {}
        FnT.tap_decisions(tap)
        try:
            res = rh._bbox_forward(st, feats, r.to(dev))
        finally:
            FnT.tap_decisions(None)
        dz_head = O.Decisions(**{k: tap[k].cpu() > 0 for k in ('lw', 'lh', 'z2', 'f1', 'f2')})
        assert res['cls_score'].dtype == torch.float16 and res['bbox_pred'].dtype == torch.float16
        o_bbox = O.half_store(O.roi_extract3(o_feats, r))
        o_cls, o_reg = O.rff_head(o_bbox, stage_params[st], decisions=dz_head)
        o_cls, o_reg = O.half_store(o_cls), O.half_store(o_reg)
        assert relerr(res['bbox_feats'], o_bbox) <= HALF_ULP, st
        assert relerr(res['cls_score'], o_cls) <= HALF_ULP and relerr(res['bbox_pred'], o_reg) <= HALF_ULP, st
        c1 = torch.randn(o_cls.shape, generator=g).half().float()
        c2 = torch.randn(o_reg.shape, generator=g).half().float()
        loss = loss + (res['cls_score'] * c1.to(dev)).sum() + (res['bbox_pred'] * c2.to(dev)).sum()
        o_loss = o_loss + (o_cls * c1).sum() + (o_reg * c2).sum()
    loss.backward()
    o_loss.backward()
    GR = 2.5e-2          # half-operand contractions (see test_fp16_mode_through_plugin_surface)
    for i in range(4):
        assert relerr_l2(xs[i].grad, xo[i].grad) <= GR, i
    for st in range(3):
        from conftest import FP16_GRAD_TOL
        for k, v in rh.bbox_head[st].named_parameters():
            assert v.grad.dtype == torch.float16
            assert relerr_l2(v.grad, stage_params[st][k].grad) <= GR, (st, k, relerr_l2(v.grad, stage_params[st][k].grad))
    for k, v in neck.named_parameters():
        if k == 'refine.phi.conv.bias':
            continue
        if Pn[k].grad is None:          # the gate conv of a level no RoI maps to: no gradient in the reference, none / zero here
            assert v.grad is None or float(v.grad.float().abs().max()) == 0.0, k
            continue
        from conftest import FP16_GRAD_TOL
        tol = FP16_GRAD_TOL if (k.startswith('refine.') and k.endswith('bias')) else GR   # see test_fp16_mode_through_plugin_surface
        assert relerr_l2(v.grad, Pn[k].grad) <= tol, (k, relerr_l2(v.grad, Pn[k].grad))


def test_fp16_mode_dtype_mismatch_raises_like_the_reference(dev):
    import arfe_amd
    neck = arfe_amd.build_neck(dict(type='WFPNDualSpatial', in_channels=8, num_levels=5)).to(dev)
    arfe_amd.wrap_fp16_model(neck)
    xs = [torch.randn(1, 8, h, w, device=dev) for h, w in PYRAMIDS['tiny']]
    with pytest.raises(RuntimeError, match='should be the same'):
        neck(xs)        # float activations meeting half parameters: conv2d's dtype error in the reference


def test_shape_sweep_in_exact_fp32_mode():
    """tools/fuzz_shapes.py: the whole path (neck -> 3-RoI extractor -> head, fwd+bwd) on five unusual configurations
    (C = 20 / 36 / 128 / 8 / 320, B = 1..3, R = 1..200, every refine level, odd map sizes) against the oracle, run with
    ARFE_GEMM_F32=1 in a fresh process: with exact-fp32 contractions the indexing of every kernel shows at ~1e-6.  The
    script counts the ReLU / argmax choices that differ from the oracle's and bounds every gradient in max-norm (1e-3)
    against the oracle evaluated under the device's choices (oracle/parity.py)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ARFE_GEMM_F32='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_shapes.py')], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count('case ') == 5


def test_shape_sweep_in_default_bf16x3_mode():
    """The same sweep through the kernels that SHIP (bf16x3 contractions, generic tiles for these channel counts and
    RoI counts that are not multiples of 32): their tails and indexing are exercised here, not only by the exact-fp32
    run above.  Same criterion: forward 1e-4, gradients 1e-3 in max-norm under the device's choices, flipped choices counted."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('ARFE_GEMM_F32', 'ARFE_CONV_STAGE_ONCE', 'ARFE_GEMM_PLAN')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_shapes.py')], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count('case ') == 5


def test_two_kernel_arfpn_backward_with_the_fused_pass_switched_off():
    """ARFE_FPN_BWD_FUSED=0 keeps the round-1 form (maps kernel + dbsf kernel) that ragged pyramids and refine levels 3 / 4
    still take; the golden AR-FPN cases and the neck sequence must pass through it as well (fresh process: the switch is
    read once)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ARFE_FPN_BWD_FUSED='0')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                          '-k', 'arfpn_golden or neck_sequence or end_to_end_golden'], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout


def test_generic_conv_kernels_with_stage_once_switched_off():
    """ARFE_CONV_STAGE_ONCE=0 (read once per process) routes the 3x3 convs through the generic implicit-GEMM kernels, which
    remain the path for RoI counts that are not a multiple of 32 and channel counts that are not a multiple of 128: the
    bench-size conv cases must pass there too (fresh process)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ARFE_CONV_STAGE_ONCE='0')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                          '-k', 'conv3x3_bench_size or rff_head_golden'], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout


def test_roialign_backward_with_sliced_lists_switched_on():
    """ARFE_RB_SEG=<min length>,<slices> (read once per process) cuts RoI lists longer than `min` over grid.z workgroups that
    add atomically into tiles zeroed by arfe_roialign_bwd_prezero - off by default since round 2, still the route for very
    long lists: the RoIAlign cases must pass through it as well (fresh process)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, ARFE_RB_SEG='6,4')
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-q', '-x', '-m', 'gpu',
                          '-k', 'roialign_golden or roialign_realistic or roialign_bwd_prepared or end_to_end_golden'],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert ' passed' in out.stdout


def test_two_rank_gloo_step_with_early_gradient_sink():
    """The N > 1 code path of bench.py on ONE GPU: two ranks over gloo run the step with GradAllReducer(early=[fc1.weight]); the
    fused AR-RFF node writes fc1's weight gradient straight into the reducer's bucket and puts it on the wire from the side
    stream (arfe_amd.functional.register_grad_sink).  Both ranks see the same data, so every averaged head gradient must equal
    the single-process gradient (tools/probe/dist_sink_check.py asserts 1e-4)."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ARFE_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                          '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'tools', 'probe', 'dist_sink_check.py')],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count('sink used for fc1: True') == 2


def test_rccl_one_rank_reducer_smoke():
    """RCCL itself on the one GPU there is: a 1-rank nccl process group under the reducer's N > 1 bookkeeping (world override):
    communicator set-up, ReduceOp.AVG, fc1's bucket launched from the side stream through the gradient sink, the other buckets
    from the hooks, waits on the main stream.  With one rank the collectives are identities: the gradients must come out as
    in a plain single-process step (tools/probe/rccl_one_rank.py; fresh process, own master port)."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'probe', 'rccl_one_rank.py')], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    if 'RCCL_UNAVAILABLE' in out.stdout:
        pytest.skip('no RCCL bootstrap here: ' + out.stdout[-300:])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'avg op True' in out.stdout


def test_bench_prints_one_json_line_with_the_contract_fields():
    """bench.py's contract with the driver: ONE JSON line on stdout with the metric of BASELINE.json, the whole-job value, the
    `roofline` object of the dominant kernel (HIP-event launch time inside the timed region) and the stage break-down; the
    first-generation switches (ARFE_WGRAD_ROWS=0, ARFE_X6_V1=1: fresh process, read once) must produce a step as well."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    for extra in ({}, {'ARFE_WGRAD_ROWS': '0', 'ARFE_X6_V1': '1'}):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-cpu-baseline',
                              '--no-fp16-leg'], env=dict(os.environ, **extra), cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines
        d = json.loads(lines[0])
        assert d['metric'].startswith('imgs/s AR-FPN+AR-RFF') and d['unit'] == 'imgs/s' and d['n_gpus'] == 1
        assert d['steps'] == 3 and d['warmup'] == 2 and d['higher_is_better'] is True and d['scaling'] == 'weak'
        assert d['vs_baseline'] is None and d['data'] == 'synthetic' and 'workload' in d['config']
        assert abs(d['value'] - 2 * 1e3 / d['ms_per_step']) <= 1e-2 * d['value']            # B = 2 images per step
        r = d['roofline']
        assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 2500.0 and r['launches'] == 3 * 6
        assert abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-4 and 0.02 < r['frac'] < 0.33
        assert set(d['stage_ms_per_step']) == {'arfpn_hbm_stages', 'nonlocal_fwd_bwd', 'conv3x3_fwd_and_dgrad'}
        assert d['dist'] == {'backend': None, 'world_size': 1, 'launcher': False}
