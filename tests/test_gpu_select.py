"""gh_select_rows / renderer.select_gaussians: the Gaussian selection of forward_single_batch (renderer_one_shot.py:468-477)
against the reference's own boolean-mask indexing (oracle.oracle_torch.select_gaussians_reference): bit-equal rows in the same
order, edge cases (nothing / everything selected, N not a multiple of the block, thresholds hit exactly, NaN scores, C = 0
or wide rows), and gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from guassianhand_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _case(N, C, seed, lo=0.1, hi=0.9, special=None):
    g = torch.Generator().manual_seed(seed)
    score = torch.rand(N, 1, generator=g)
    if special == "none":
        score = score * 0.05
    elif special == "all":
        score = 0.95 + 0.05 * score
    elif special == "exact":                           # values exactly on the thresholds are NOT selected (strict >)
        score[::3] = lo
        score[1::3] = hi
    elif special == "nan":
        score[::7] = float("nan")
    return score, torch.randn(N, 3, generator=g), torch.randn(N, C, generator=g)


@pytest.mark.parametrize("N,C,special", [(1, 5, None), (63, 131, None), (64, 131, "all"), (257, 131, "none"), (1000, 131, "exact"),
                                         (4099, 7, "nan"), (98562, 131, None), (5000, 1, None), (300, 200, None)])
def test_selection_equals_boolean_mask_indexing(dev, N, C, special):
    from guassianhand_amd.renderer import select_gaussians
    from oracle.oracle_torch import select_gaussians_reference
    score, pts, feat = _case(N, C, 3 * N + C, special=special)
    want = select_gaussians_reference(score, pts, feat, 0.1, 0.9)
    got = select_gaussians(score.to(dev), pts.to(dev), feat.to(dev), 0.1, 0.9)
    for w, gt in zip(want, got):
        assert gt.shape == w.shape and torch.equal(gt.cpu(), w)


def test_selection_gradients(dev):
    from guassianhand_amd.renderer import select_gaussians
    from oracle.oracle_torch import select_gaussians_reference
    score, pts, feat = _case(777, 33, 5)
    wts = [torch.randn(777, 3), torch.randn(777, 33), torch.randn(777, 3), torch.randn(777, 33)]

    def run(fn, to):
        p, f = pts.clone().to(to).requires_grad_(True), feat.clone().to(to).requires_grad_(True)
        outs = fn(score.to(to), p, f, 0.1, 0.9)
        loss = sum((o * w.to(to)[:o.shape[0]]).sum() for o, w in zip(outs, wts))
        loss.backward()
        return p.grad.cpu(), f.grad.cpu()
    gp_ref, gf_ref = run(select_gaussians_reference, "cpu")
    gp, gf = run(select_gaussians, dev)
    assert torch.allclose(gp, gp_ref, rtol=0, atol=1e-6) and torch.allclose(gf, gf_ref, rtol=0, atol=1e-6)


def test_c_abi_argument_checks(dev):
    import ctypes as C
    from guassianhand_amd import _abi, _lib
    L = _lib.lib()
    counts = torch.full((2,), 7, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    # N = 0: counts are zeroed, nothing else is touched
    assert L.gh_select_rows(None, 0, 0.1, 0.9, None, None, 4, None, None, None, None, None, None, p(counts), None, 0, None) == 0
    torch.cuda.synchronize()
    assert counts.tolist() == [0, 0]
    sc = torch.rand(10, device=dev)
    pts = torch.rand(10, 3, device=dev)
    assert L.gh_select_rows(p(sc), 10, 0.1, 0.9, p(pts), None, 4, p(pts), None, p(pts), None, None, None, p(counts), None, 0, None) == _abi.GH_ERR_INVALID_ARG
    ws = torch.empty(4, dtype=torch.uint8, device=dev)
    feat = torch.rand(10, 4, device=dev)
    o = [torch.empty(10, 3, device=dev), torch.empty(10, 4, device=dev), torch.empty(10, 3, device=dev), torch.empty(10, 4, device=dev)]
    assert L.gh_select_rows(p(sc), 10, 0.1, 0.9, p(pts), p(feat), 4, p(o[0]), p(o[1]), p(o[2]), p(o[3]), None, None, p(counts), p(ws), 4,
                            None) == _abi.GH_ERR_WORKSPACE_SMALL
