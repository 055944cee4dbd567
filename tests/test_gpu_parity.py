"""Parity tests proper: the HIP path (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json): RGB L_inf <= 1e-4 and gradient rtol <= 1e-3. The shared arithmetic contract
 makes the forward bit-exact, so the tests assert equality for every integer/decision array
(radii, tile rects, sorted keys, ranges, n_contrib) AND for the fp32 forward outputs, and keep the stated
tolerance for the gradients, whose cross-pixel sums are ordered differently (oracle: double, sequential)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import dimg_like, max_rel, rel_l2, scene_kwargs

pytestmark = pytest.mark.gpu

GRAD_RTOL = 1e-3      # BASELINE.json north_star: "1e-3 grad rtol"
IMG_LINF = 1e-4       # BASELINE.json north_star: "1e-4 RGB L_inf"


def gpu_render(sc, dev, sync=True, **over):
    from guassianhand_amd.rasterizer import raster_forward
    s = sc.to(dev)
    kw, bl = scene_kwargs(s)
    kw.update(over)
    cams = sc.cams().to(dev)       # same camera bytes as the oracle (tan/atan2 differ by 1 ulp between CPU and GPU libm)
    return raster_forward(cams, s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=sync, **kw, **bl)


def oracle_render(sc, debug=True):
    from oracle.oracle_c import OracleRender
    kw, bl = scene_kwargs(sc)
    o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=debug, **kw, **bl)
    if debug and o.num_rendered > o.debug["sorted_keys"].numel():       # huge footprints: list longer than the default debug arrays
        n = o.num_rendered
        o.close()
        o = OracleRender(sc.cams(), sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=sc.H, W=sc.W, debug=True, debug_capacity=n, **kw, **bl)
    return o


def compare(sc, dev, check_stages=True, grad_l2=1e-5, grad_rtol=GRAD_RTOL, check_grads=True):
    from guassianhand_amd.rasterizer import raster_backward, workspace_views
    img, radii, ctx = gpu_render(sc, dev)
    orc = oracle_render(sc)
    torch.cuda.synchronize()
    wv = workspace_views(ctx)
    D = int(wv["counters"][0].item())
    # exact tile culling: the instance list is the oracle's (3-sigma rect) list minus tiles the alpha >= 1/255 ellipse
    # cannot reach; what is dropped blends nowhere, so everything downstream of the lists must still be bit-equal
    assert 0 <= D <= orc.num_rendered
    if bool((orc.debug["n_contrib"] > 0).any()):
        assert D > 0
    assert torch.equal(radii.cpu(), orc.radii)
    if check_stages:
        N = sc.P * sc.w2c.shape[0]
        vis = orc.radii.reshape(-1) > 0
        g0, g1, gb = wv["g0"].cpu(), wv["g1"].cpu(), wv["gb"].cpu()
        tt = wv["tiles_touched"].cpu().long()
        oo = orc.debug["offsets"].reshape(N).long()
        rect_tiles = oo - torch.cat([oo.new_zeros(1), oo[:-1]])
        assert bool((tt <= rect_tiles).all()) and int(tt.sum()) == D
        inst = tt > 0                                    # Gaussians with at least one instance carry the full record
        assert torch.equal(g0[inst, :2], orc.debug["xy"].reshape(N, 2)[inst])
        assert torch.equal(torch.stack([g0[inst, 2], g0[inst, 3], g1[inst, 0], g1[inst, 1]], 1), orc.debug["conic_opacity"].reshape(N, 4)[inst])
        assert torch.equal(torch.stack([g1[inst, 2], g1[inst, 3], gb[inst]], 1), orc.debug["rgb"].reshape(N, 3)[inst])
        assert torch.equal(wv["rect"].cpu()[inst], orc.debug["rect"].reshape(N)[inst])
        # two-level sort (per-view depth order of the Gaussians, then a stable partition by tile): lists are laid out
        # tile-major / view-minor, so compare per tile: read in ascending global tile id (stable), the (tile, gaussian)
        # sequence is a subsequence, in order, of the oracle's stable sort by tile<<32|depth
        g_tile_raw = wv["sorted_tile"][:D].cpu().long()
        g_slot = wv["sorted_slot"][:D].cpu().long()
        g_gid_raw = wv["sorted_gid"][:D].cpu().long()
        assert torch.equal(g_slot.sort().values, torch.arange(D))               # emit slots are a permutation of 0..D-1
        by_tile = torch.argsort(g_tile_raw, stable=True)
        g_tile, g_gid = g_tile_raw[by_tile], g_gid_raw[by_tile]
        o_tile, o_gid = (orc.debug["sorted_keys"] >> 32).long(), orc.debug["sorted_gid"].long()
        o_pair, g_pair = o_tile * N + o_gid, g_tile * N + g_gid
        order = torch.argsort(o_pair)                                            # pairs are unique
        so = o_pair[order]
        idx = torch.searchsorted(so, g_pair).clamp(max=max(so.numel() - 1, 0))
        assert torch.equal(so[idx], g_pair)                                      # every GPU instance is an oracle instance ...
        where = order[idx]
        assert bool((where[1:] > where[:-1]).all())                              # ... and they come in the oracle's order
        # ranges index the culled list consistently: every tile's range is one contiguous run of its own instances
        rng = wv["ranges"].cpu().long()
        cnt = torch.bincount(g_tile_raw, minlength=rng.shape[0])
        assert torch.equal(rng[:, 1] - rng[:, 0], cnt)
        nz = cnt > 0
        first = torch.full((rng.shape[0],), D, dtype=torch.long).scatter_reduce(0, g_tile_raw, torch.arange(D), reduce="amin")
        last = torch.full((rng.shape[0],), -1, dtype=torch.long).scatter_reduce(0, g_tile_raw, torch.arange(D), reduce="amax")
        assert torch.equal(rng[nz, 0], first[nz]) and torch.equal(rng[nz, 1], last[nz] + 1)
        # per pixel: same transmittance, and the last blended entry is the same Gaussian as in the oracle's list
        assert torch.equal(wv["final_T"].cpu(), orc.debug["final_T"])
        NV, H, W = sc.w2c.shape[0], sc.H, sc.W
        gx, gy = (W + 15) // 16, (H + 15) // 16
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        tile_of = (torch.arange(NV)[:, None, None] * (gx * gy) + (yy // 16 * gx + xx // 16)[None]).reshape(-1)
        nc_g, nc_o = wv["n_contrib"].cpu().long().reshape(-1), orc.debug["n_contrib"].long().reshape(-1)
        assert torch.equal(nc_g > 0, nc_o > 0)
        o_rng = orc.debug["ranges"].long()
        sel = nc_g > 0
        last_g = g_gid_raw[(rng[tile_of, 0] + nc_g - 1)[sel]]
        last_o = o_gid[(o_rng[tile_of, 0] + nc_o - 1)[sel]]
        assert torch.equal(last_g, last_o)
    assert (img.cpu() - orc.image).abs().max().item() <= IMG_LINF
    assert torch.equal(img.cpu(), orc.image), "forward is expected to be bit-exact under the arithmetic contract"
    dimg = dimg_like(sc.w2c.shape[0], sc.H, sc.W)
    g = raster_backward(ctx, dimg.to(dev))
    og = orc.backward(dimg)
    assert set(g) == set(og)
    for k in og:
        assert bool(torch.isfinite(g[k]).all()), k
        if check_grads:
            rtol = grad_rtol.get(k, GRAD_RTOL) if isinstance(grad_rtol, dict) else grad_rtol
            assert rel_l2(g[k].cpu(), og[k]) <= grad_l2, k
            assert max_rel(g[k].cpu(), og[k]) <= rtol, (k, max_rel(g[k].cpu(), og[k]))
    print(f"instances: {D} after exact tile culling, {orc.num_rendered} in the 3-sigma rects")
    orc.close()
    return D


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from guassianhand_amd import _lib
    _lib.lib()                      # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


@pytest.mark.parametrize("use_rgb,blend,nv", [(True, False, 1), (False, False, 1), (True, True, 1), (False, True, 2), (True, True, 3)])
def test_config0_random1k(dev, use_rgb, blend, nv):
    """BASELINE configs[0]: 1k random Gaussians, 128x128."""
    from guassianhand_amd.scenes import make_scene
    compare(make_scene("random1k", n_views=nv, use_rgb=use_rgb, blend=blend), dev)


def test_per_gaussian_color_w_and_nonzero_xyz_b(dev):
    from guassianhand_amd.scenes import make_scene
    for rgb in (True, False):
        sc = make_scene("random1k", n_views=2, P=800, use_rgb=rgb, blend=True)
        g = torch.Generator().manual_seed(4)
        sc.color_w = 1 + 0.05 * torch.randn(sc.P, 48, generator=g)      # edit renderer's (P,48) form
        sc.xyz_b = 0.005 * torch.randn(3, generator=g)
        compare(sc, dev)


def test_sh_w_only_blend(dev):
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=500, use_rgb=False, blend=True)
    sc.color_b = None
    compare(sc, dev)


@pytest.mark.parametrize("deg,M", [(0, 1), (1, 4), (2, 9), (3, 16), (1, 16), (2, 16)])
def test_sh_degrees(dev, deg, M):
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=400, use_rgb=False, blend=False)
    sc.shs = sc.shs[:, :M].contiguous()
    sc.sh_degree = deg
    compare(sc, dev)


def test_sh_negative_colours_are_clamped_with_zero_grad(dev):
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=400, use_rgb=False, blend=False)
    sc.shs[:, 0, :] -= 1.2           # many channels go negative -> clamp flags exercised
    compare(sc, dev)


@pytest.mark.parametrize("H,W", [(16, 16), (17, 33), (40, 250), (334, 512)])
def test_ragged_image_sizes(dev, H, W):
    """Partial tiles on both edges; tiny images give an odd radix pass count (tile_bits <= 8)."""
    from guassianhand_amd.scenes import make_scene, ring_cameras
    sc = make_scene("random1k", n_views=2, P=600)
    sc.H, sc.W = H, W
    sc.w2c, sc.K = ring_cameras(torch.zeros(3), 2, H, W, 1.3 * max(H, W))
    compare(sc, dev)


def test_empty_and_fully_culled(dev):
    from guassianhand_amd.rasterizer import raster_backward, raster_forward
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=64)
    sc.xyz = sc.xyz - torch.tensor([0.0, 0.0, 5.0])          # everything behind the camera
    s = sc.to(dev)
    img, radii, ctx = raster_forward(sc.cams().to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W,
                                     colors_precomp=s.shs.squeeze(1))
    assert int(radii.abs().sum()) == 0 and float(img.abs().max()) == 0.0
    g = raster_backward(ctx, torch.ones(1, 3, sc.H, sc.W, device=dev))
    assert all(float(v.abs().max()) == 0.0 for v in g.values())
    # white background shows through untouched
    sc.bg = torch.tensor([1.0, 0.5, 0.25])
    img2, _, _ = gpu_render(sc, dev)
    assert torch.equal(img2[0, :, 0, 0].cpu(), sc.bg)
    # P = 0
    z = torch.zeros(0, 3, device=dev)
    img3, r3, ctx3 = raster_forward(s.cams(), z, torch.zeros(0, 1, device=dev), z, torch.zeros(0, 4, device=dev),
                                    H=sc.H, W=sc.W, colors_precomp=z)
    assert r3.shape == (1, 0) and float(img3.abs().max()) == 0.0


def test_edge_semantics_match_oracle(dev):
    """Near-plane cull edge, alpha clamp (opacity >= 1 and > 1 after opacity_b), negative opacity, equal-depth
    ties, early termination through 10^4 stacked Gaussians, sub-pixel Gaussians (0.3 dilation)."""
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=12000)
    P = sc.P
    g = torch.Generator().manual_seed(8)
    sc.xyz[:10000] = torch.tensor([0.01, -0.02, 0.0]) + 1e-4 * torch.randn(10000, 3, generator=g)   # stacked
    sc.xyz[10000:10400, 2] = sc.xyz[10000, 2]                                                       # equal depth ties
    sc.xyz[10400:10500] = sc.xyz[10400:10401]                                                      # fully identical
    sc.opacity[10500:10700] = 1.0
    sc.opacity[10700:10800] = 1.4
    sc.opacity[10800:10900] = -0.3
    sc.scaling[10900:11100] = 1e-6
    sc.xyz[11100:11200, 2] = -0.8 + 0.2 + 0.01 * torch.randn(100, generator=g)                      # straddle z = 0.2 (camera at z=-1)
    D = compare(sc, dev)
    assert D > 0


def test_culling_is_conservative_for_thin_faint_and_border_gaussians(dev):
    """The render kernels skip list entries whose alpha >= 1/255 ellipse misses the wave's 4x4 pixel block. A wrong
    skip would change pixels, so stress the test: needle-thin (1000:1) Gaussians at all angles, opacities hugging
    1/255, centres exactly on block / tile borders and far outside the image, giant and sub-pixel footprints —
    the image must still equal the oracle (which has no culling) bit for bit."""
    import math
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=6000)
    g = torch.Generator().manual_seed(12)
    P = sc.P
    sc.scaling[:2000] = torch.stack([torch.full((2000,), 0.03), torch.full((2000,), 3e-5), torch.full((2000,), 3e-5)], 1)  # needles
    sc.scaling[2000:2300] = 0.2                                                   # giants covering many tiles
    sc.scaling[2300:2600] = 1e-7                                                  # sub-pixel (dilation only)
    sc.opacity[2600:3600] = (1 / 255) * (1 + 0.02 * torch.randn(1000, 1, generator=g))   # around the alpha threshold
    sc.opacity[3600:3800] = 1 / 255
    sc.opacity[:1000] = 0.01 + 0.02 * torch.rand(1000, 1, generator=g)           # faint needles: tiny ellipses
    # centres that project exactly onto block borders of view 0 (f=325, 128x128, camera at z=-1 looking at +z)
    k = torch.arange(4000, 5000)
    sc.xyz[k, 0] = ((k % 33) * 4 - 64 + 0.5).float() / 325.0
    sc.xyz[k, 1] = (((k // 33) % 33) * 4 - 64 + 0.5).float() / 325.0
    sc.xyz[k, 2] = 0.0
    sc.xyz[5000:5200, 0] += 0.5                                                   # far outside the frustum
    # The subject here is the FORWARD (bit-exact, asserted inside compare). Every gradient keeps the 1e-3 bar except
    # dL/dscales and dL/drotations: the chain rule dSigma2D -> dSigma3D -> (dscale, dR) of a 1000:1 needle subtracts terms
    # ~(s_max / s_min)^2 = 1e6 times larger than the result, in fp32 on BOTH sides (the oracle's per-Gaussian stage is fp32 as
    # well; only its pixel sums are double). Measured against the float64 dense autograd oracle on this scene at a third of
    # its size (tools/needle_accuracy.py, profiles/r3_needle_accuracy.txt): scales / rotations max-rel 2.0e-3 / 4.1e-3 for
    # the HIP path (float64 record sums), 2.5e-3 / 4.7e-3 for the C oracle — neither float32 implementation is within 1e-3
    # of the truth here, and the HIP path is the closer one; the two differ from each other by ≈1e-3 (1.1e-3 on scales in that
    # run, 1.2e-3 on rotations at this test's size).
    compare(sc, dev, grad_l2=1e-4, grad_rtol={"scales": 3e-3, "rotations": 3e-3})


def test_frustum_clamp_edge_gradients(dev):
    """Gaussians beyond 1.3x the frustum: the clamped view-space coordinate carries no x/y gradient
    (App. A.4-3); a huge scale keeps them on screen so the clamp branch is actually exercised."""
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=1, P=300)
    sc.xyz[:150, 0] += 0.9       # far outside in x (tan = 0.197 at f=325/128px, z~1)
    sc.scaling[:150] = 0.3
    sc.opacity[:150] = 0.05
    compare(sc, dev)


def test_many_views_chain_rule_groups(dev):
    """More views than one lane group holds (the chain-rule kernel splits a Gaussian's views over <= 64 lanes and loops
    for the rest) and a view count that is not a power of two (padded lanes must contribute nothing)."""
    from guassianhand_amd.scenes import make_scene
    for nv in (70, 6):
        compare(make_scene("random1k", n_views=nv, P=300, use_rgb=True, blend=True), dev, check_stages=False)
    compare(make_scene("random1k", n_views=67, P=120, use_rgb=False, blend=True), dev, check_stages=False)
    # SH mode with more views than a workgroup has lanes: the SH backward falls back to its 16-lanes-per-Gaussian form
    # (gh_sh_colour_bwd_kernel); 3 and 100 views: view counts that do not divide the two-phase kernel's 256 lanes
    compare(make_scene("random1k", n_views=260, P=40, use_rgb=False, blend=True), dev, check_stages=False)
    for nv in (3, 100):
        compare(make_scene("random1k", n_views=nv, P=90, use_rgb=False, blend=True), dev, check_stages=False)


def test_config1_one_hand(dev):
    """BASELINE configs[1]: single right hand, 49,281 Gaussians, 512x334, forward + backward."""
    from guassianhand_amd.scenes import make_scene
    D = compare(make_scene("one_hand", n_views=1), dev)
    assert D > 20000


def test_config2_two_hands_blend(dev):
    """BASELINE configs[2] (the headline workload): 98,562 Gaussians, interaction-aware blend, 512x334."""
    from guassianhand_amd.scenes import make_scene
    D = compare(make_scene("two_hands", n_views=2), dev)
    assert D > 98562


def test_two_hands_at_the_reference_render_size_256(dev):
    """The only render size the reference itself uses: H = W = 256 hard-coded at its one call site (infer_one_shot.py:283-284;
    512x334 and 1024x1024 are BASELINE.json's). Two hands, 98,562 Gaussians, blend on, the same field of view at half the
    resolution (f = 650), two ring views."""
    from guassianhand_amd.scenes import make_scene, ring_cameras
    sc = make_scene("two_hands", n_views=2)
    sc.H = sc.W = 256
    sc.w2c, sc.K = ring_cameras(sc.xyz.mean(0), 2, 256, 256, 650.0)
    D = compare(sc, dev)
    assert D > 98562 // 2


def test_depth_ranges_wider_than_the_three_pass_sort(dev):
    """GH_FLAG_DEPTH24 (three depth-sort passes) covers visible depths whose float bit patterns share their top byte, e.g.
    everything in [0.5, 2) m. The flag is a speculation the device can only answer with a NaN image, so it is used
    only for a call shape a read-back has shown it to hold for: every full forward reports whether the top byte varied
    (GhCounters.overflow bit 4), with or without the flag. A scene that straddles 0.5 m and 2 m comes out bit-exact through four
    passes and the shape learns False; a shape that was LEARNED to hold and whose scene then moves across a boundary is flagged
    (bit 3): a sync call re-runs with four passes transparently, a sync-free call returns NaN and raises at the check."""
    from guassianhand_amd import _abi
    from guassianhand_amd import rasterizer as R
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=1500)
    sc.xyz[:, 2] = sc.xyz[:, 2] * 9.0 + 0.25                  # camera at z = -1: depths 0.35 .. 2.15 m
    sc.scaling = sc.scaling * 2.0
    key = R.capacity_key(sc.P, 2, sc.H, sc.W)
    R._depth24.pop(key, None)
    compare(sc, dev)                                          # unknown shape: four passes, bit-exact against the oracle
    assert R._depth24.get(key) is False                       # ... and the call reported that its top byte varies
    # the verdict forced to "holds" (as if learned on an earlier, narrower scene of this shape): sync=True re-runs transparently
    R._depth24[key] = True
    compare(sc, dev)
    assert R._depth24.get(key) is False
    # ... and sync-free: NaN image + an error at the check, and the shape is demoted again
    R._depth24[key] = True
    s = sc.to(dev)
    img, _, ctx = R.raster_forward(sc.cams().to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, sync=False,
                                   colors_precomp=s.shs.squeeze(1))
    assert ctx.dims.flags & _abi.GH_FLAG_DEPTH24
    with pytest.raises(R.GhOverflowError, match="24 key bits"):
        R.check_overflow()
    assert bool(torch.isnan(img).all()) and R._depth24.get(key) is False
    # a scene inside one factor-4 range: the first call (four passes) learns that three suffice, the second uses them — both bit-exact
    sc2 = make_scene("random1k", n_views=2, P=1500, seed=5)
    sc2.H, sc2.W = 96, 80
    key2 = R.capacity_key(sc2.P, 2, 96, 80)
    R._depth24.pop(key2, None)
    compare(sc2, dev)
    assert R._depth24.get(key2) is True
    s2 = sc2.to(dev)
    _, _, ctx2 = R.raster_forward(sc2.cams().to(dev), s2.xyz, s2.opacity, s2.scaling, s2.rotation, H=96, W=80, sync=False,
                                  colors_precomp=s2.shs.squeeze(1))
    assert ctx2.dims.flags & _abi.GH_FLAG_DEPTH24             # (also for an explicit sync=False call: the knowledge is there now)
    R.check_overflow()
    compare(sc2, dev)
    # a sync-free loop that never read anything back knows nothing: four passes
    R._depth24.pop(key2, None)
    _, _, ctx3 = R.raster_forward(sc2.cams().to(dev), s2.xyz, s2.opacity, s2.scaling, s2.rotation, H=96, W=80, sync=False,
                                  colors_precomp=s2.shs.squeeze(1))
    assert not (ctx3.dims.flags & _abi.GH_FLAG_DEPTH24)
    R.check_overflow()                                        # ... until a read-back arrives: the verdict is learned from it
    assert R._depth24.get(key2) is True


def test_reference_init_scale(dev):
    """The reference's initial Gaussian size exp(-5) = 6.7 mm (renderer_one_shot.py:165): ~10x more instances."""
    from guassianhand_amd.scenes import make_scene
    compare(make_scene("one_hand", n_views=1, P=20000, scale_mean=-5.0), dev)


def test_golden_vectors(dev, golden_dir):
    from guassianhand_amd.rasterizer import raster_backward, raster_forward
    z = np.load(os.path.join(golden_dir, "raster_golden.npz"))
    for n in [str(c) for c in z["cases"]]:
        t = {k[len(n) + 4:]: torch.tensor(z[k]).to(dev) for k in z.files if k.startswith(n + "_in_")}
        H, W = [int(v) for v in z[f"{n}_HW"]]
        kw = {k: t[k] for k in ("colors_precomp", "shs", "xyz_b", "opacity_b", "color_w", "color_b") if k in t}
        if "shs" in t:
            kw["sh_degree"] = 3
        img, radii, ctx = raster_forward(t["cams"], t["means3D"], t["opacities"], t["scales"], t["rotations"], H=H, W=W, **kw)
        assert np.array_equal(img.cpu().numpy(), z[f"{n}_out_image"]), n
        assert np.array_equal(radii.cpu().numpy(), z[f"{n}_out_radii"])
        g = raster_backward(ctx, t["dL_dimage"])
        for k, v in g.items():
            ref = torch.tensor(z[f"{n}_grad_{k}"])
            assert max_rel(v.cpu(), ref) <= GRAD_RTOL and rel_l2(v.cpu(), ref) <= 1e-5, (n, k)


def test_overflow_is_reported_not_silent(dev):
    from guassianhand_amd.rasterizer import GhOverflowError, raster_forward
    from guassianhand_amd.scenes import make_scene
    s = make_scene("random1k", n_views=1).to(dev)
    with pytest.raises(GhOverflowError):
        raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=s.H, W=s.W, colors_precomp=s.shs.squeeze(1),
                       max_instances=100)


def test_footprints_larger_than_the_64_tile_hit_mask(dev):
    """Rects of more than 64 tiles do not fit the per-Gaussian tile hit mask: the emit / ranges kernels fall back to
    re-running the ellipse/tile test. 512x334 (21x32 tiles) with Gaussians from a few tiles to the whole image,
    elongated diagonal ones included (large rect, fewer hit tiles)."""
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("one_hand", n_views=2, P=3000)
    g = torch.Generator().manual_seed(17)
    sc.scaling[:40] = 0.25                                                        # whole image
    sc.scaling[40:200] = torch.stack([torch.full((160,), 0.09), torch.full((160,), 0.015), torch.full((160,), 0.03)], 1)   # large, 6:1
    sc.scaling[200:400] = 0.02 + 0.03 * torch.rand(200, 3, generator=g)          # 10 .. 25 tiles across
    sc.opacity[:400] = 0.02 + 0.1 * torch.rand(400, 1, generator=g)              # keep the image from saturating
    # The subject is the instance lists and the (bit-exact) forward. A whole-image Gaussian's gradient is a signed sum
    # of 1.7e5 pixel terms that largely cancel: fp32 partial sums vs the oracle's double accumulation differ by up to a
    # few % on the near-cancelled components (rel-L2 stays <= 2e-4); well-conditioned scenes keep the 1e-3 bar.
    D = compare(sc, dev, grad_l2=2e-4, grad_rtol=GRAD_RTOL)
    assert D > 20000


def test_culling_margins_hold_for_extreme_anisotropy_and_far_offscreen_centres(dev):
    """q = A dx^2 + 2B dx dy + C dy^2 of the tile / block tests is evaluated in fp32; for strongly elongated
    conics whose centre lies far outside the image the terms cancel at ~1e6..1e8 against a threshold of ~11, so the
    margin scales with the term magnitude (gh_block_hit / gh_block_mask16). Long needles (up to 3000:1) whose centres
    sit hundreds to thousands of pixels off screen but whose long axis crosses the image must keep every pixel they
    blend: the forward stays bit-equal to the oracle (which has no culling)."""
    import math
    from guassianhand_amd.scenes import make_scene
    sc = make_scene("random1k", n_views=2, P=3000)
    g = torch.Generator().manual_seed(23)
    n = 2400
    # needles pointing at the image centre from far away: centre at distance r along direction phi, long axis along phi
    r = 0.5 + 6.0 * torch.rand(n, generator=g)                      # metres off axis at z ~ 1 m (f = 325 px/m: 160 .. 2100 px)
    phi = 2 * math.pi * torch.rand(n, generator=g)
    sc.xyz[:n, 0] = r * torch.cos(phi)
    sc.xyz[:n, 1] = r * torch.sin(phi)
    sc.xyz[:n, 2] = 0.05 * torch.randn(n, generator=g)
    L = r * (0.35 + 0.3 * torch.rand(n, generator=g))                 # 1-sigma half length: the 3-sigma tip reaches the image
    sc.scaling[:n] = torch.stack([L, L / 3000.0 * (1 + 9 * torch.rand(n, generator=g)), torch.full((n,), 1e-5)], 1)
    q = torch.stack([torch.cos(phi / 2), torch.zeros(n), torch.zeros(n), torch.sin(phi / 2)], 1)   # rotation about z by phi
    sc.rotation[:n] = q
    sc.opacity[:n] = 0.05 + 0.9 * torch.rand(n, 1, generator=g)
    # The FORWARD (lists, bit-exact image, n_contrib) is the subject. No gradient parity is claimed for these degenerate
    # footprints: the chain rule of a 3000:1 needle hundreds of sigma off screen cancels to nothing in fp32 on either side
    # (the oracle's fp32 per-Gaussian stage included); gradients are only required to be finite.
    compare(sc, dev, check_grads=False)


def test_conics_that_are_not_positive_definite_are_never_culled(dev):
    """Needles tens of metres long lying across the image at ~45 degrees: a c and b^2 of the dilated 2-D covariance agree to
    more digits than fp32 holds, so `det = a c - b^2` comes out with a random sign (the reference only skips det == 0,
    App. A.1-5). With det < 0 the conic is indefinite and `alpha >= 1/255` holds on two unbounded wedges; the convexity
    argument of the culling tests does not apply, so gh_block_hit / gh_block_mask16 answer "hit" for every tile / block of
    the rect. The forward must stay bit-equal to the oracle, which has no culling."""
    import math
    from guassianhand_amd.rasterizer import raster_forward, workspace_views
    from guassianhand_amd.scenes import make_scene
    from tests.helpers import scene_kwargs
    sc = make_scene("random1k", n_views=1, P=1500)
    g = torch.Generator().manual_seed(31)
    n = 1200
    sc.xyz[:n, :2] = 0.15 * (torch.rand(n, 2, generator=g) - 0.5)
    sc.xyz[:n, 2] = 0.05 * torch.randn(n, generator=g)
    L = 20.0 * 10 ** torch.rand(n, generator=g)                                    # 20 .. 200 m (6,500 .. 65,000 px at f = 325)
    sc.scaling[:n] = torch.stack([L, L * 1e-6, torch.full((n,), 1e-6)], 1)
    phi = math.pi / 4 + 0.5 * (torch.rand(n, generator=g) - 0.5)
    sc.rotation[:n] = torch.stack([torch.cos(phi / 2), torch.zeros(n), torch.zeros(n), torch.sin(phi / 2)], 1)
    sc.opacity[:n] = 0.02 + 0.5 * torch.rand(n, 1, generator=g)
    s = sc.to(dev)
    kw, bl = scene_kwargs(s)
    _, _, ctx = raster_forward(s.cams(), s.xyz, s.opacity, s.scaling, s.rotation, H=sc.H, W=sc.W, **kw, **bl)
    wv = workspace_views(ctx)
    inst = wv["tiles_touched"][:n] > 0
    A, B, C = wv["g0"][:n, 2], wv["g0"][:n, 3], wv["g1"][:n, 0]
    not_pd = inst & ~((C > 0) & (A * C - B * B > 0))
    assert int(not_pd.sum()) >= 50, "the scene no longer produces indefinite conics"
    compare(sc, dev, check_grads=False)
