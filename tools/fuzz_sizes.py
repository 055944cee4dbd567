"""Size-class fuzz: the HIP path against the C oracle on scenes that select the DIFFERENT code paths the launch logic picks by size.

tools/fuzz_features.py covers the feature matrix at P <= 2,500, where every radix pass runs with 4 keys per thread, every segment
fits the scan-free histogram form and the tile ids fit one or two narrow passes. What is chosen by size (gh_binning.hip, gh_api.hip):
    depth sort     4 / 8 keys per thread (total keys >= 2^19), blocks per view <= 128 (every block sums the histogram rows itself)
                   or more (row-scan kernel), three passes (GH_FLAG_DEPTH24 holds) or four (visible depths straddle a factor-4 boundary)
    tile partition 4 / 8 / 16 keys per thread (capacity 2^21, 2^25), one 1024-digit pass (9-10 tile bits of ONE view), two or three
                   8-bit passes (up to 17 tile bits: 32 views of 1024^2), row scans of more than 1024 blocks (two sweeps)
    render         lists of a few entries ... tens of thousands per tile, images up to 255 tiles wide
Each iteration draws P in 1 ... 400,000, 1-32 views, images from 16^2 to 4080 pixels wide, a camera distance that keeps or breaks the
24-bit depth assumption, and a scale distribution that sets the instances per Gaussian (D up to ~2e7); image and radii must be
bit-equal to the oracle's (run in its OpenMP mode, which is bit-identical to the checker mode: tests/test_oracle_cross.py), gradients
within 3e-4 rel-L2 when D <= 4e6 (a path check, not an accuracy bar: fuzz_features.py has those).   usage: fuzz_sizes.py [n_iterations] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import rasterizer as R
from guassianhand_amd.camera import pack_cameras_from_w2c
from guassianhand_amd.scenes import ring_cameras
from oracle import oracle_c
from oracle.oracle_c import OracleRender
from tests.helpers import rel_l2, limit_torch_threads_to_the_cpu_share

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--only=")]
FORCE_CLASS = next((a.split("=")[1] for a in sys.argv if a.startswith("--class=")), None)      # e.g. --class=tiny (P in {0, 1, 2, 63 ... 257})
rnd = random.Random(seed)
dev = torch.device("cuda:0")
limit_torch_threads_to_the_cpu_share()
oracle_c.set_parallel(True)
oracle_c.set_num_threads(torch.get_num_threads())
bad = []
seen = dict(items_depth=set(), self_hist=set(), depth24=set(), tile_bits=set(), items_tile=set(), max_D=0)


def one(it):
    rnd.seed(seed * 1000003 + it)
    g = torch.Generator().manual_seed(it)
    cls = rnd.choice(["tiny", "small", "mid", "mid", "large", "wide", "many_views", "dense"] * 3 + ["huge"])
    cls = FORCE_CLASS or cls
    if cls == "tiny":
        P, NV, H, W = rnd.choice([0, 1, 2, 63, 64, 65, 255, 257]), rnd.randint(1, 3), rnd.randint(16, 64), rnd.randint(16, 64)
    elif cls == "small":
        P, NV, H, W = rnd.randint(1000, 20000), rnd.randint(1, 4), rnd.randint(64, 400), rnd.randint(64, 400)
    elif cls == "mid":
        P, NV, H, W = rnd.randint(50000, 140000), rnd.randint(1, 8), rnd.choice([334, 512, 640]), rnd.choice([334, 512, 640])
    elif cls == "large":
        P, NV, H, W = rnd.randint(150000, 400000), rnd.randint(1, 3), rnd.choice([512, 1024]), rnd.choice([512, 1024])
    elif cls == "wide":
        P, NV, H, W = rnd.randint(2000, 30000), 1, rnd.choice([16, 64, 200]), rnd.choice([2048, 4080])
        if rnd.random() < 0.5:
            H, W = W, H
    elif cls == "many_views":
        P, NV, H, W = rnd.randint(2000, 20000), rnd.choice([9, 16, 32]), rnd.choice([256, 512, 1024]), rnd.choice([256, 512, 1024])
    elif cls == "huge":                                       # > 2^25 instances: the tile partition runs with 16 keys per thread
        P, NV, H, W = rnd.randint(100000, 200000), rnd.randint(4, 8), 512, rnd.choice([334, 512])
    else:                                                     # dense: many instances per Gaussian
        P, NV, H, W = rnd.randint(20000, 80000), rnd.randint(1, 4), rnd.choice([334, 512]), rnd.choice([334, 512])
    radius = rnd.choice([0.6, 1.0, 1.0, 1.7, 2.2, 6.0])       # [0.5, 2) is one 24-bit depth range: 1.7 / 2.2 with extent 0.5 straddle 2.0
    ext = rnd.choice([0.2, 0.5]) * min(1.0, radius)
    f = 0.9 * max(H, W) * radius / ext * rnd.choice([0.5, 1.0])
    xyz = (torch.rand(P, 3, generator=g) - 0.5) * ext
    base = ext / max(1.0, P ** (1 / 3)) * (4.0 if cls in ("dense", "huge") else rnd.choice([0.3, 1.0, 2.0]))
    # keep the expected instance count under ~1.5e7 (the oracle walks every one of them on the host's cores); "huge": ~4.5e7
    limit = 4.5e7 if cls == "huge" else 1.5e7
    sigma_px = base * f / radius
    est = P * NV * min((6.0 * sigma_px / 16.0 + 1.0) ** 2, ((W + 15) // 16) * ((H + 15) // 16))
    base *= max(0.02, (limit / est) ** 0.5) if (est > limit or cls == "huge") else 1.0
    scaling = base * torch.exp(0.5 * torch.randn(P, 3, generator=g))
    rot = torch.randn(P, 4, generator=g); rot = rot / rot.norm(dim=1, keepdim=True)
    opacity = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g)) * rnd.choice([0.3, 1.0])
    use_rgb = rnd.random() < 0.7
    col = torch.rand(P, 3, generator=g) if use_rgb else torch.cat([0.5 + 0.3 * torch.randn(P, 1, 3, generator=g), 0.2 * torch.randn(P, 15, 3, generator=g)], 1)
    w2c, K = ring_cameras(torch.zeros(3), NV, H, W, f, radius=radius)
    cams = pack_cameras_from_w2c(w2c, K, H, W, torch.tensor([0.05, 0.1, 0.15]))
    tag = f"it {it} {cls} P={P} NV={NV} {H}x{W} radius={radius} ext={ext:.2f} f={f:.0f} scale={base:.2e} {'rgb' if use_rgb else 'sh3'}"
    kw = dict(colors_precomp=col) if use_rgb else dict(shs=col, sh_degree=3)
    # the call variant: plain | the views as two halves on two streams | the fused attribute blend | static lists + a refresh with
    # other opacities and colours (compared on the refreshed values)
    variant = rnd.choice(["plain", "plain", "split", "blend", "static_refresh", "pose_batch"])
    if variant == "split" and NV < 2:
        variant = "plain"
    if variant == "pose_batch" and (NV < 2 or NV * P > 1_500_000 or P == 0):
        variant = "plain"
    if variant == "pose_batch":
        # NV different Gaussian sets in one launch sequence (GH_FLAG_PER_VIEW_GAUSSIANS): every per-Gaussian tensor holds NV * P rows;
        # compared view by view with the oracle's render of that view's rows
        xs = torch.cat([xyz + 0.02 * ext * v * torch.randn(P, 3, generator=g) for v in range(NV)])
        ops = torch.cat([(opacity * (0.5 + 0.5 * torch.rand(P, 1, generator=g))) for _ in range(NV)])
        rep = lambda x: x.repeat(NV, *([1] * (x.dim() - 1)))
        sc_all, rot_all, col_all = rep(scaling), rep(rot), rep(col)
        kw_all = dict(colors_precomp=col_all) if use_rgb else dict(shs=col_all, sh_degree=3)
        tag += " pose_batch"
        imgp, radp, ctxp = R.raster_forward(cams.to(dev), xs.to(dev), ops.to(dev), sc_all.to(dev), rot_all.to(dev), H=H, W=W, sync=True,
                                            per_view_gaussians=True, **{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw_all.items()})
        Dp = R.last_num_rendered()
        seen["max_D"] = max(seen["max_D"], Dp)
        seen["pose_batches"] = seen.get("pose_batches", 0) + 1
        print(f"{tag}: D={Dp} rows={NV * P}", flush=True)
        for v in range(NV):
            sl = slice(v * P, (v + 1) * P)
            kv = dict(colors_precomp=col_all[sl]) if use_rgb else dict(shs=col_all[sl], sh_degree=3)
            ov = OracleRender(cams[v:v + 1], xs[sl], ops[sl], sc_all[sl], rot_all[sl], H=H, W=W, **kv)
            assert torch.equal(imgp[v].cpu(), ov.image[0]) and torch.equal(radp[v].cpu(), ov.radii[0]), tag + f" (view {v})"
            ov.close()
        gp = R.raster_backward(ctxp, torch.randn(NV, 3, H, W, generator=g).to(dev), want_means2D=False)
        assert all(bool(torch.isfinite(x).all()) for x in gp.values()), tag + " (gradient not finite)"
        del imgp, radp, ctxp, gp
        R.clear_workspace_pool(); torch.cuda.empty_cache()
        return
    bl = {}
    if variant == "blend":
        bl = dict(xyz_b=0.01 * ext * torch.randn(3, generator=g), opacity_b=0.02 * torch.randn(P, 1, generator=g),
                  color_w=1 + 0.05 * torch.randn(48, generator=g), color_b=0.02 * torch.randn(P, 48, generator=g))
    tag += f" {variant}"
    todev = lambda d_: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d_.items()}
    geo = (cams.to(dev), xyz.to(dev), opacity.to(dev), scaling.to(dev), rot.to(dev))
    if variant == "static_refresh":
        _, _, ctx0 = R.raster_forward(*geo, H=H, W=W, sync=True, static_lists=True, **todev(kw))
        opacity = (opacity * (0.3 + 0.7 * torch.rand(P, 1, generator=g))).clamp(max=1.0)
        col = col + 0.1 * torch.randn(col.shape, generator=g)
        kw = dict(colors_precomp=col) if use_rgb else dict(shs=col, sh_degree=3)
    t0 = time.time()
    o = OracleRender(cams, xyz, opacity, scaling, rot, H=H, W=W, **kw, **bl)
    t_or = time.time() - t0
    if variant == "static_refresh":
        img, radii, ctx = R.raster_forward(geo[0], geo[1], opacity.to(dev), geo[3], geo[4], H=H, W=W, sync=True, refresh_of=ctx0, **todev(kw))
    else:
        img, radii, ctx = R.raster_forward(*geo, H=H, W=W, sync=True, split_streams=(variant == "split"), **todev(kw), **todev(bl))
    D = R.last_num_rendered()
    key = (P, NV, H, W, variant == "split")
    d24 = R._depth24.get(key) is True
    if d24 and variant != "static_refresh":
        # the first call of a shape runs four depth-sort passes and learns that three suffice: the three-pass path is the SECOND call
        img2, radii2, ctx2 = R.raster_forward(*geo, H=H, W=W, sync=True, split_streams=(variant == "split"), **todev(kw), **todev(bl))
        assert ctx2.dims.flags & 32, tag + " (GH_FLAG_DEPTH24 not used by the second call)"
        assert torch.equal(img2, img) and torch.equal(radii2, radii), tag + " (three-pass depth sort differs from four-pass)"
        del img2, radii2, ctx2
    tiles = NV * ((W + 15) // 16) * ((H + 15) // 16)
    tb = max(1, (tiles - 1).bit_length())
    cap = int(ctx.dims.max_instances)
    seen["items_depth"].add(8 if P * NV >= (1 << 19) else 4)
    seen["self_hist"].add((P + 256 * (8 if P * NV >= (1 << 19) else 4) - 1) // (256 * (8 if P * NV >= (1 << 19) else 4)) <= 128)
    seen["depth24"].add(bool(d24)); seen["tile_bits"].add(tb)
    seen["items_tile"].add(4 if cap <= (1 << 21) else (8 if cap <= (1 << 25) else 16))
    seen["max_D"] = max(seen["max_D"], D)
    print(f"{tag}: D={D} cap={cap} tile bits {tb} depth24={d24} oracle {t_or:.1f} s", flush=True)
    assert torch.equal(radii.cpu(), o.radii), tag + " (radii)"
    same = torch.equal(img.cpu(), o.image)
    if not same:
        d = (img.cpu() - o.image)
        raise AssertionError(tag + f" (image: {int((d != 0).sum())} values differ, {int(torch.isnan(d).sum())} NaN, max {float(d[~torch.isnan(d)].abs().max()):.3g})")
    if D <= 4_000_000:
        dimg = torch.randn(NV, 3, H, W, generator=g)
        og = o.backward(dimg)
        gr = R.raster_backward(ctx, dimg.to(dev), want_means2D=False)
        for k, b in og.items():
            if k == "means2D":
                continue
            if b.numel() == 0:                                  # (P = 0: empty gradients on both sides)
                assert gr[k].numel() == 0, tag + f" ({k}: a gradient for no Gaussians)"
                continue
            a = gr[k].cpu().reshape(b.shape)
            assert bool(torch.isfinite(a).all()), tag + f" ({k} not finite)"
            if float(b.abs().max()) > 0:
                l2 = rel_l2(a, b)
                assert l2 <= 3e-4, (tag, k, l2)
    else:
        # too many instances for the oracle's backward on the host: the gradients are checked for finiteness and — the views as two
        # halves on two streams promise bit-identical gradients — against the split (or unsplit) call of the same inputs
        dimg = torch.randn(NV, 3, H, W, generator=g).to(dev)
        gr = {k: v.clone() for k, v in R.raster_backward(ctx, dimg, want_means2D=False).items()}
        for k, v in gr.items():
            assert bool(torch.isfinite(v).all()), tag + f" ({k} not finite at D={D})"
        if NV >= 2 and variant in ("plain", "blend", "split"):
            img2, _, ctx2 = R.raster_forward(*geo, H=H, W=W, sync=True, split_streams=(variant != "split"), **todev(kw), **todev(bl))
            assert torch.equal(img2, img), tag + " (split vs unsplit image)"
            gr2 = R.raster_backward(ctx2, dimg, want_means2D=False)
            for k in gr:
                assert torch.equal(gr[k], gr2[k]), tag + f" (split vs unsplit gradient {k} at D={D})"
            seen["big_bwd"] = seen.get("big_bwd", 0) + 1
            del img2, ctx2, gr2
    o.close()
    ctx0 = None
    del img, radii, ctx
    R.clear_workspace_pool()
    torch.cuda.empty_cache()


t0 = time.time()
for it in (only or range(n_iter)):
    try:
        one(it)
    except AssertionError as e:
        bad.append(str(e)[:600]); print("MISMATCH", bad[-1], flush=True)
    except Exception as e:
        bad.append(f"it {it}: {type(e).__name__}: {e}"[:600]); print("ERROR", bad[-1], flush=True)
        R.clear_workspace_pool(); torch.cuda.empty_cache()
R.check_overflow()
print(f"size-class fuzz: {n_iter} iterations (seed {seed}) in {time.time() - t0:.0f} s; paths seen: depth-sort keys/thread {sorted(seen['items_depth'])}, "
      f"scan-free histogram {sorted(seen['self_hist'])}, depth24 {sorted(seen['depth24'])}, tile bits {sorted(seen['tile_bits'])}, "
      f"tile-partition keys/thread {sorted(seen['items_tile'])}, largest D {seen['max_D']}, {seen.get('big_bwd', 0)} backward passes above 4e6 instances "
      f"checked split against unsplit, {seen.get('pose_batches', 0)} pose batches; {len(bad)} findings")
for b in bad[:20]:
    print("  ", b)
sys.exit(1 if bad else 0)
