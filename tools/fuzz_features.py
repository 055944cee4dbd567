"""Feature-matrix fuzz of the HIP path against the C oracle: random scene shapes x colour modes x blend subsets x call variants.

Every iteration draws a scene (P, views, image size off the 16-pixel grid, SH degree / RGB, a random subset of the blend terms in
their (48,) / (P,48) / (P,3) forms, stress distributions of scale and opacity) and ONE call variant
    plain | split streams | static lists + refresh with other opacities and colours | second call over shared geometry (mask pass) |
    occlusion bound from an earlier call of slightly different positions | pose batch (per-view Gaussians)
and checks: image and radii bit-equal to the oracle's, fused alpha bit-equal to the oracle's mask render, every gradient within the
north star's tolerances (rel-L2 <= 1e-5, element-wise <= 1e-3 with the scale-aware floor of tests.helpers.max_rel), and a `want`
subset of the gradients bit-equal to the same entries of the full set.   usage: fuzz_features.py [n_iterations] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.scenes import make_scene
from oracle.oracle_c import OracleRender
from tests.helpers import float64_grads, rel_l2, max_rel

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
only = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--only=")]
dev = torch.device("cuda:0")
GRAD_L2, GRAD_RTOL = 1e-5, 1e-3
VARIANTS = ("plain", "split", "static_refresh", "shared_mask", "depth_bound", "pose_batch")
count = {v: 0 for v in VARIANTS}
bad = []
worst_by = {m: [0.0, 0.0] for m in ("needles", "threshold", "giants", "stack", "mix", "default")}
cur = {"tag": ""}


def draw_scene(it):
    P = rnd.choice([1, 7, 64, 300, 1000, 2500])
    nv = rnd.randint(1, 4)
    rgb = rnd.random() < 0.5
    sc = make_scene("random1k", n_views=nv, P=P, use_rgb=rgb, blend=True, seed=rnd.randint(0, 10 ** 6))
    g = torch.Generator().manual_seed(1000 + it)
    mode = rnd.randrange(6)
    if mode == 0:      # needles
        a = 10 ** (-1.5 - 3 * torch.rand(P, generator=g)); b = a * 10 ** (-3 * torch.rand(P, generator=g))
        sc.scaling = torch.stack([a, b, b], 1)
    elif mode == 1:    # opacities around the 1/255 threshold
        sc.opacity = ((1 / 255) * (1 + 0.1 * torch.randn(P, 1, generator=g))).clamp(min=0.5 / 255)
    elif mode == 2:    # giants
        sc.scaling = 10 ** (-2.0 + 1.5 * torch.rand(P, 3, generator=g)); sc.opacity = 0.02 + 0.2 * torch.rand(P, 1, generator=g)
    elif mode == 3:    # dense opaque stack: early stops everywhere
        sc.scaling = 10 ** (-2.3 + 0.3 * torch.rand(P, 3, generator=g)); sc.opacity = 0.6 + 0.39 * torch.rand(P, 1, generator=g)
    elif mode == 4:    # wide mix
        sc.scaling = 10 ** (-4.5 + 3.5 * torch.rand(P, 3, generator=g)); sc.opacity = torch.sigmoid(3 * torch.randn(P, 1, generator=g))
    sc.H, sc.W = rnd.randint(16, 150), rnd.randint(16, 150)
    if not rgb:
        sc.sh_degree = rnd.randint(0, 3)
        M = rnd.choice([m for m in (1, 4, 9, 16) if m >= (sc.sh_degree + 1) ** 2])
        sc.shs = sc.shs[:, :M].contiguous()
    # a random subset of the blend terms, in their alternative forms
    if rnd.random() < 0.4: sc.xyz_b = None
    else: sc.xyz_b = 0.004 * torch.randn(3, generator=g)
    if rnd.random() < 0.4: sc.opacity_b = None
    if rnd.random() < 0.4: sc.color_w = None
    elif rnd.random() < 0.5: sc.color_w = 1 + 0.05 * torch.randn(P, 48, generator=g)
    if rnd.random() < 0.4: sc.color_b = None
    if not rgb and sc.shs.shape[1] != 16:          # the blend's 48 columns are the 16 SH coefficients: only with all of them present
        sc.color_w = sc.color_b = None
    if not rgb and sc.color_w is None:             # SH mode: the bias line multiplies by color_w again (renderer_one_shot.py:334)
        sc.color_b = None
    sc.bg = torch.rand(3, generator=g) if rnd.random() < 0.5 else torch.zeros(3)
    cur["mode"] = ("needles", "threshold", "giants", "stack", "mix", "default")[mode]
    return sc


def colour_kw(sc, t=lambda x: x):
    return dict(colors_precomp=t(sc.shs.squeeze(1))) if sc.use_rgb else dict(shs=t(sc.shs), sh_degree=sc.sh_degree)


def blend_kw(sc, t=lambda x: x, b3=False):
    out = {k: t(getattr(sc, k)) for k in ("xyz_b", "opacity_b", "color_w", "color_b") if getattr(sc, k) is not None}
    if b3 and "color_b" in out:
        out["color_b"] = out["color_b"][:, :3].contiguous()          # GH_FLAG_BLEND_COLOR_B_RGB: the three columns RGB mode reads
    return out


refereed = [0, 0]          # [gradient tensors sent to the referee, of which the HIP path was the worse of the two float32 programs]


def check_grads(tag, g, og, want=None):
    keys = set(og) if want is None else set(want)
    assert keys <= set(g), (tag, sorted(keys), sorted(g))
    for k in keys:
        a, b = g[k].cpu(), og[k]
        if k == "color_b" and a.shape != b.shape:                   # (P,3) form against the oracle's (P,48)
            assert float(b[:, 3:].abs().max()) == 0.0
            b = b[:, :3]
        a = a.reshape(b.shape)
        assert bool(torch.isfinite(a).all()), (tag, k)
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, (tag, k)
            continue
        l2, mr = rel_l2(a, b), max_rel(a, b)
        strict = cur["mode"] in ("default", "stack") and b.numel() >= 300
        w = worst_by[cur["mode"]]
        w[0], w[1] = max(w[0], l2), max(w[1], mr)
        # the oracle is a float32 program too: on the stress distributions (needles of aspect 1000, footprints of thousands of
        # pixels, alphas at the 1/255 threshold) two float32 summation orders differ by far more than on the scenes the north star's
        # tolerances are quoted for (tools/needle_accuracy.py: both sit equally far from a float64 evaluation) — there the check is
        # a sanity bar that a wrong formula or a misplaced row would still break by orders of magnitude
        ok = (l2 <= GRAD_L2 and mr <= GRAD_RTOL) if strict else l2 <= 3e-3
        if not ok and cur["referee"] is not None:
            # two float32 programs disagree: ask the float64 one which of them is off. A finding = the HIP path is far from the
            # float64 gradient AND several times farther than the float32 oracle is (on needles of aspect 1000 both sit 1e-3 away
            # from float64, either one the farther by a factor of up to ~5 depending on the scene: that is conditioning, not a bug)
            if not isinstance(cur["referee"], dict):
                cur["referee"] = cur["referee"]()
            ref = cur["referee"][k].reshape(b.shape).double()
            e_hip, e_orc = rel_l2(a, ref), rel_l2(b, ref)
            refereed[0] += 1
            print(f"referee {tag} {k}: HIP vs float32 oracle rel-L2 {l2:.2e} / element-wise {mr:.2e}; against float64: HIP {e_hip:.2e}, oracle {e_orc:.2e}", flush=True)
            ok = e_hip <= (2e-5 if strict else 1e-2) or e_hip <= 5.0 * e_orc
            refereed[1] += 0 if ok else 1
            if not ok or only:                                   # where the difference sits: the worst rows, all three values
                a2, b2, r2 = (x.reshape(x.shape[0], -1).double() if x.dim() > 1 else x.reshape(1, -1).double() for x in (a, b, ref))
                rows = (a2 - r2).norm(dim=1).topk(min(3, a2.shape[0])).indices.tolist()
                for i in rows:
                    extra = f" scaling {cur['scaling'][i].tolist()} opacity {float(cur['opacity'][i]):.4f}" if cur.get("scaling") is not None and a2.shape[0] == cur["scaling"].shape[0] else ""
                    print(f"    row {i}: HIP {a2[i].tolist()} oracle {b2[i].tolist()} float64 {r2[i].tolist()}{extra}", flush=True)
        elif not ok:
            ok = l2 <= 1e-2 and not strict                      # (no referee for this tensor: sums over all Gaussians, the mask pass)
        assert ok, (tag, k, l2, mr)


def one(it):
    rnd.seed(seed * 1000003 + it)                              # every iteration can be replayed on its own (--only)
    cur["referee"] = None
    sc = draw_scene(it)
    variant = rnd.choice(VARIANTS)
    if variant == "split" and sc.w2c.shape[0] < 2:
        variant = "plain"
    if variant == "shared_mask" and not sc.use_rgb:
        variant = "plain"
    count[variant] += 1
    cur["scaling"], cur["opacity"] = sc.scaling, sc.opacity.reshape(-1)
    P, NV, H, W = sc.P, sc.w2c.shape[0], sc.H, sc.W
    cams = sc.cams()
    b3 = sc.use_rgb and rnd.random() < 0.5
    cur["tag"] = tag = f"it {it} {variant}/{cur['mode']} P={P} NV={NV} {H}x{W} {'rgb' if sc.use_rgb else 'sh%d/M%d' % (sc.sh_degree, sc.shs.shape[1])} blend={sorted(blend_kw(sc))} b3={b3}"
    s = sc.to(dev)
    cg = cams.to(dev)
    gdev = lambda x: x.to(dev)
    dimg = torch.randn(NV, 3, H, W, generator=torch.Generator().manual_seed(it))
    ident = lambda x: x

    def oracle(scn, cams_, **over):
        kw = colour_kw(scn); kw.update(over)
        return OracleRender(cams_, scn.xyz, scn.opacity, scn.scaling, scn.rotation, H=H, W=W, **kw, **blend_kw(scn))

    if variant in ("plain", "split"):
        o = oracle(sc, cams)
        img, radii, ctx = R.raster_forward(cg, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, return_alpha=True,
                                           split_streams=(variant == "split"), **colour_kw(s), **blend_kw(s, b3=b3))
        assert torch.equal(img.cpu(), o.image) and torch.equal(radii.cpu(), o.radii), tag
        # the fused alpha channel == the oracle's mask render (colour 1, background 0) of the same geometry
        cm = cams.clone(); cm[:, 37:40] = 0
        ones = torch.ones(P, 3)
        om = OracleRender(cm, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=H, W=W, colors_precomp=ones,
                          **{k: v for k, v in blend_kw(sc).items() if k in ("xyz_b", "opacity_b")})
        assert torch.equal(ctx.alpha.cpu(), om.image[:, 0]), tag + " (alpha)"
        og = o.backward(dimg)
        og.pop("means2D", None)
        cur["referee"] = lambda: float64_grads(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, sc.shs, sc.use_rgb, sc.sh_degree, blend_kw(sc), dimg, H, W)
        full = {k: v.clone() for k, v in R.raster_backward(ctx, dimg.to(dev), want_means2D=False).items()}
        check_grads(tag, full, og)
        # a subset of the gradients: the same bits as in the full set
        names = sorted(og)
        sub = set(rnd.sample(names, rnd.randint(1, len(names))))
        part = R.raster_backward(ctx, dimg.to(dev), want_means2D=False, want=sub)
        for k in sub:
            assert torch.equal(part[k], full[k]), tag + f" (want subset: {k})"
        if variant == "plain":
            # the fused image loss (GhOutputs.l1_*): same image, the gradient of mean|image - target| bit for bit, the loss against float64
            tgt = torch.rand(NV, 3, H, W, generator=torch.Generator().manual_seed(it + 7)).to(dev)
            if rnd.random() < 0.3:
                tgt[:, :, : H // 2] = img[:, :, : H // 2]             # exact zeros: no gradient there (torch.abs' backward)
            img1, radii1, ctx1 = R.raster_forward(cg, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, l1_target=tgt,
                                                  **colour_kw(s), **blend_kw(s, b3=b3))
            assert ctx1.l1 is not None and torch.equal(img1, img) and torch.equal(radii1, radii), tag + " (fused loss: image)"
            inv_n = torch.tensor(1.0 / img.numel(), dtype=torch.float64).to(torch.float32).to(dev)       # (float)(1.0 / n), as the library rounds it
            assert torch.equal(ctx1.l1[1], torch.sign(img - tgt) * inv_n), tag + " (fused loss: gradient)"
            ref = float((img.double() - tgt.double()).abs().mean())
            assert abs(float(ctx1.l1[0]) - ref) <= 3e-6 * ref + 1e-12, tag + f" (fused loss: {float(ctx1.l1[0])} vs {ref})"
        om.close(); o.close()
    elif variant == "static_refresh":
        img0, radii0, ctx0 = R.raster_forward(cg, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, static_lists=True,
                                              **colour_kw(s), **blend_kw(s, b3=b3))
        o0 = oracle(sc, cams)
        assert torch.equal(img0.cpu(), o0.image) and torch.equal(radii0.cpu(), o0.radii), tag + " (static forward)"
        o0.close()
        g = torch.Generator().manual_seed(7 + it)
        sc2 = sc.to("cpu")
        sc2.opacity = (sc.opacity * (0.2 + 0.8 * torch.rand(P, 1, generator=g))).clamp(max=1.0)
        sc2.shs = sc.shs + 0.1 * torch.randn(sc.shs.shape, generator=g)
        if sc.opacity_b is not None:
            sc2.opacity_b = sc.opacity_b * 0.5
        if sc.color_b is not None:
            sc2.color_b = sc.color_b + 0.01 * torch.randn(sc.color_b.shape, generator=g)
        s2 = sc2.to(dev)
        o = oracle(sc2, cams)
        img, radii, ctx = R.raster_forward(cg, s2.xyz, s2.opacity, s2.scaling, s2.rotation, H=H, W=W, refresh_of=ctx0,
                                           **colour_kw(s2), **blend_kw(s2, b3=b3))
        assert torch.equal(img.cpu(), o.image), tag + " (refresh)"
        og = o.backward(dimg)
        og.pop("means2D", None)
        cur["referee"] = lambda: float64_grads(cams, sc2.xyz, sc2.opacity, sc2.scaling, sc2.rotation, sc2.shs, sc2.use_rgb, sc2.sh_degree, blend_kw(sc2), dimg, H, W)
        check_grads(tag, R.raster_backward(ctx, dimg.to(dev), want_means2D=False), og)
        o.close()
    elif variant == "shared_mask":
        img0, radii0, ctx0 = R.raster_forward(cg, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, **colour_kw(s), **blend_kw(s, b3=b3))
        cm = cams.clone(); cm[:, 37:40] = 0
        ones = torch.ones(P, 3)
        geo = {k: v for k, v in blend_kw(sc).items() if k in ("xyz_b", "opacity_b")}
        om = OracleRender(cm, sc.xyz, sc.opacity, sc.scaling, sc.rotation, H=H, W=W, colors_precomp=ones, **geo)
        img, radii, ctx = R.raster_forward(cm.to(dev), s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, geometry_of=ctx0,
                                           colors_precomp=ones.to(dev), **{k: gdev(v) for k, v in geo.items()})
        assert torch.equal(img.cpu(), om.image), tag + " (mask pass)"
        og = om.backward(dimg)
        og.pop("means2D", None)
        cur["referee"] = lambda: float64_grads(cm, sc.xyz, sc.opacity, sc.scaling, sc.rotation, ones.reshape(P, 1, 3), True, 0, geo, dimg, H, W)
        check_grads(tag + " (mask pass)", R.raster_backward(ctx, dimg.to(dev), want_means2D=False), og)
        o = oracle(sc, cams)                                   # and the first call's own backward, after the second call ran
        og = o.backward(dimg)
        og.pop("means2D", None)
        cur["referee"] = lambda: float64_grads(cams, sc.xyz, sc.opacity, sc.scaling, sc.rotation, sc.shs, sc.use_rgb, sc.sh_degree, blend_kw(sc), dimg, H, W)
        check_grads(tag + " (rgb pass)", R.raster_backward(ctx0, dimg.to(dev), want_means2D=False), og)
        om.close(); o.close()
    elif variant == "depth_bound":
        cache = R.DepthBoundCache(refresh_every=1, min_pixels=0)
        R.raster_forward(cg, s.xyz, s.opacity, s.scaling, s.rotation, H=H, W=W, depth_bound=cache, **colour_kw(s), **blend_kw(s, b3=b3))
        sc2 = sc.to("cpu")
        sc2.xyz = sc.xyz + rnd.choice([0.0, 1e-4, 1e-3]) * torch.randn(P, 3, generator=torch.Generator().manual_seed(it))
        s2 = sc2.to(dev)
        o = oracle(sc2, cams)
        img, radii, ctx = R.raster_forward(cg, s2.xyz, s2.opacity, s2.scaling, s2.rotation, H=H, W=W, depth_bound=cache,
                                           **colour_kw(s2), **blend_kw(s2, b3=b3))
        assert torch.equal(img.cpu(), o.image) and torch.equal(radii.cpu(), o.radii), tag + f" (bounded calls {cache.bounded_calls}, misses {cache.misses})"
        og = o.backward(dimg)
        og.pop("means2D", None)
        cur["referee"] = lambda: float64_grads(cams, sc2.xyz, sc2.opacity, sc2.scaling, sc2.rotation, sc2.shs, sc2.use_rgb, sc2.sh_degree, blend_kw(sc2), dimg, H, W)
        check_grads(tag, R.raster_backward(ctx, dimg.to(dev), want_means2D=False), og)
        o.close()
    else:                                                       # pose batch: NV different Gaussian sets, one launch sequence
        g = torch.Generator().manual_seed(11 + it)
        xyz = torch.cat([sc.xyz + 0.01 * v * torch.randn(P, 3, generator=g) for v in range(NV)])
        rep = lambda x: None if x is None else x.repeat(NV, *([1] * (x.dim() - 1)))
        per = dict(opacity=rep(sc.opacity) * (0.5 + 0.5 * torch.rand(NV * P, 1, generator=g)), scaling=rep(sc.scaling), rotation=rep(sc.rotation), shs=rep(sc.shs))
        bl = blend_kw(sc)
        wpg = "color_w" in bl and bl["color_w"].numel() != 48
        blp = {k: (v if k == "xyz_b" or (k == "color_w" and not wpg) else rep(v)) for k, v in bl.items()}
        kwp = dict(colors_precomp=per["shs"].squeeze(1)) if sc.use_rgb else dict(shs=per["shs"], sh_degree=sc.sh_degree)
        img, radii, ctx = R.raster_forward(cg, xyz.to(dev), per["opacity"].to(dev), per["scaling"].to(dev), per["rotation"].to(dev), H=H, W=W,
                                           per_view_gaussians=True, **{k: gdev(v) for k, v in kwp.items() if k != "sh_degree"},
                                           **({"sh_degree": sc.sh_degree} if not sc.use_rgb else {}), **{k: gdev(v) for k, v in blp.items()})
        grads = R.raster_backward(ctx, dimg.to(dev), want_means2D=False)
        shared_sum = {}
        for v in range(NV):
            sl = slice(v * P, (v + 1) * P)
            kwv = dict(colors_precomp=per["shs"][sl].squeeze(1)) if sc.use_rgb else dict(shs=per["shs"][sl], sh_degree=sc.sh_degree)
            blv = {k: (x if k == "xyz_b" or (k == "color_w" and not wpg) else x[sl]) for k, x in blp.items()}
            o = OracleRender(cams[v:v + 1], xyz[sl], per["opacity"][sl], per["scaling"][sl], per["rotation"][sl], H=H, W=W, **kwv, **blv)
            assert torch.equal(img[v].cpu(), o.image[0]) and torch.equal(radii[v].cpu(), o.radii[0]), tag + f" (view {v})"
            og = o.backward(dimg[v:v + 1])
            og.pop("means2D", None)
            cur["referee"] = lambda v=v, sl=sl, blv=blv: float64_grads(cams[v:v + 1], xyz[sl], per["opacity"][sl], per["scaling"][sl], per["rotation"][sl],
                                                                       per["shs"][sl], sc.use_rgb, sc.sh_degree, blv, dimg[v:v + 1], H, W)
            shared = {k for k in og if k == "xyz_b" or (k == "color_w" and not wpg)}     # one tensor for all views: summed below
            gv = {k: grads[k].reshape(NV, P, *grads[k].shape[1:])[v] for k in og if k not in shared}
            check_grads(tag + f" (view {v})", gv, {k: x for k, x in og.items() if k not in shared})
            for k in shared:
                shared_sum[k] = shared_sum.get(k, 0) + og[k].double()
            o.close()
        cur["referee"] = None
        check_grads(tag + " (shared blend terms, summed over the views)", grads, {k: x.float() for k, x in shared_sum.items()})
    return variant


t0 = time.time()
for it in (only or range(n_iter)):
    try:
        one(it)
    except AssertionError as e:
        bad.append(str(e)[:400])
        print("MISMATCH", str(e)[:400], flush=True)
    except Exception as e:                                         # a crash is a finding too: say where and go on
        bad.append(f"{cur['tag']}: {type(e).__name__}: {e}"[:400])
        print("ERROR", bad[-1], flush=True)
    if (it + 1) % 25 == 0:
        print(f"{it + 1} iterations, {len(bad)} findings, {time.time() - t0:.0f} s", flush=True)
torch.cuda.synchronize()
R.check_overflow()
print(f"feature fuzz: {n_iter} iterations (seed {seed}): {count}; {len(bad)} findings; {refereed[0]} gradient tensors went to the float64 referee, "
      f"the HIP path was the worse float32 program in {refereed[1]}")
print("worst gradient deviation from the (float32) oracle by scene distribution, rel-L2 / element-wise: " + "; ".join(f"{m} {w[0]:.1e} / {w[1]:.1e}" for m, w in worst_by.items()))
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
