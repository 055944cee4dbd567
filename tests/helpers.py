"""Shared helpers for the parity tests (oracle = checker, HIP path = thing under test)."""
import math
import torch

from guassianhand_amd.scenes import make_scene


def scene_kwargs(sc, on_cpu=True):
    """(colour kwargs, blend kwargs) for OracleRender / raster_forward from a Scene."""
    kw = dict(colors_precomp=sc.shs.squeeze(1)) if sc.use_rgb else dict(shs=sc.shs, sh_degree=sc.sh_degree)
    bl = {k: getattr(sc, k) for k in ("xyz_b", "opacity_b", "color_w", "color_b") if getattr(sc, k) is not None}
    return kw, bl


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b, floor=1e-3):
    """max |a-b| / (|b| + floor*max|b|): the 'grad rtol' of BASELINE.json with a scale-aware floor."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(((a - b).abs() / (b.abs() + floor * b.abs().max() + 1e-30)).max())


def dimg_like(nv, H, W, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(nv, 3, H, W, generator=g)


def oracle_render_views(gs, w2cs, Ks, H, W, bg, color_w=None, xyz_b=None, color_b=None, opacity_b=None, *, use_rgb=True,
                        sh_degree=3, scaling_modifier=1.0, sync=True):
    """CPU stand-in for guassianhand_amd.renderer.render_views built on the dense autograd oracle (checker):
    same signature and outputs, differentiable, used to exercise the host logic of the fit loop without a GPU."""
    from guassianhand_amd.camera import pack_cameras_from_w2c
    from oracle import oracle_torch as OT
    cams = pack_cameras_from_w2c(w2cs, Ks, H, W, bg)
    means, op, cols, shs = OT.blend_attributes(gs.xyz, gs.opacity, gs.shs, use_rgb=use_rgb, color_w=color_w, xyz_b=xyz_b,
                                               color_b=color_b, opacity_b=opacity_b)
    rgbs, masks = [], []
    for c in cams:
        kw = dict(viewmatrix=c[:16].reshape(4, 4), projmatrix=c[16:32].reshape(4, 4), campos=c[32:35],
                  tanfovx=float(c[35]), tanfovy=float(c[36]), H=H, W=W, scale_modifier=scaling_modifier)
        ckw = dict(colors_precomp=cols) if use_rgb else dict(shs=shs, sh_degree=sh_degree)
        img, _ = OT.rasterize_dense(means, op, gs.scaling, gs.rotation, bg=c[37:40], **kw, **ckw)
        m, _ = OT.rasterize_dense(means, op, gs.scaling, gs.rotation, bg=torch.zeros(3), colors_precomp=torch.ones_like(means), **kw)
        rgbs.append(img.permute(1, 2, 0))
        masks.append(m.permute(1, 2, 0))
    return {"comp_rgb": torch.stack(rgbs), "comp_mask": torch.stack(masks), "comp_rgb_bg": bg, "3dgs": gs}


def tiny_fit_problem(P=160, n_views=4, hw=(32, 32), map_hw=(16, 32), seed=0, device="cpu"):
    """A small one-shot fit problem: frozen Gaussians + target images rendered with 'true' blend maps."""
    from guassianhand_amd.renderer import GaussianModel
    from guassianhand_amd.scenes import make_scene, ring_cameras
    sc = make_scene("random1k", n_views=n_views, P=P, seed=20240610 + seed)
    H, W = hw
    sc.w2c, sc.K = ring_cameras(torch.zeros(3), n_views, H, W, 2.5 * W)
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand(P, 2, generator=g) * 2 - 1
    gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling * 2.0, sc.shs)
    true = dict(color_w=1 + 0.1 * torch.randn(48, generator=g), color_b=0.1 * torch.randn(48, *map_hw, generator=g),
                opacity_b=0.05 * torch.randn(1, *map_hw, generator=g))
    mv = lambda t: t.to(device)
    return dict(gs=GaussianModel(*[mv(t) for t in gs]), uv=mv(uv), w2c=mv(sc.w2c), K=mv(sc.K), H=H, W=W, bg=mv(torch.zeros(3)),
                true={k: mv(v) for k, v in true.items()}, map_hw=map_hw)


def forward_single_view(gs, viewpoint_camera, background_color: torch.Tensor, ret_mask: bool = True,
                        color_w=None, xyz_b=None, color_b=None, opacity_b=None, *, use_rgb: bool = True,
                        sh_degree: int = 3, scaling_modifier: float = 1.0):
    """The call protocol of GS3DRenderer.forward_single_view (renderer_one_shot.py:259-382) restated for the tests and the
    host-cost tools: blend in torch, then the RGB call and the mask call through the drop-in GaussianRasterizer — what the
    reference's unmodified function does on the import shim."""
    import math
    from guassianhand_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    device = gs.xyz.device
    screenspace_points = torch.zeros_like(gs.xyz, dtype=gs.xyz.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    mk = lambda bg, deg: GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.height), image_width=int(viewpoint_camera.width), tanfovx=tanfovx,
        tanfovy=tanfovy, bg=bg, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform.float(), sh_degree=deg, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=mk(background_color, sh_degree))
    means3D = gs.xyz
    if xyz_b is not None:
        means3D = means3D + xyz_b
    opacity = gs.opacity
    if opacity_b is not None:
        opacity = opacity + opacity_b.view(-1, 1)
    shs, colors_precomp = None, None
    if use_rgb:
        colors_precomp = gs.shs.squeeze(1)
        if color_w is not None:
            colors_precomp = colors_precomp * color_w.view(-1, 16, 3)[:, 0, :] + color_w.view(-1, 16, 3)[:, 1, :] - 1
        if color_b is not None:
            colors_precomp = colors_precomp + color_b.view(-1, 16, 3)[:, 0, :]
    else:
        shs = gs.shs
        if color_w is not None:
            shs = shs * color_w.view(-1, 16, 3)
        if color_b is not None:
            shs = shs * color_w.view(-1, 16, 3) + color_b.view(-1, 16, 3)
    rendered_image, radii = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs,
                                       colors_precomp=colors_precomp, opacities=opacity, scales=gs.scaling,
                                       rotations=gs.rotation, cov3D_precomp=None)
    ret = {"comp_rgb": rendered_image.permute(1, 2, 0), "comp_rgb_bg": background_color}
    if ret_mask:
        mask_bg = torch.zeros(3, dtype=torch.float32, device=device)
        rasterizer = GaussianRasterizer(raster_settings=mk(mask_bg, 0))
        rendered_mask, radii = rasterizer(means3D=means3D, means2D=screenspace_points,
                                          colors_precomp=torch.ones_like(means3D), opacities=opacity,
                                          scales=gs.scaling, rotations=gs.rotation, cov3D_precomp=None)
        ret["comp_mask"] = rendered_mask.permute(1, 2, 0)
    return ret


def limit_torch_threads_to_the_cpu_share():
    """torch's CPU thread pool defaults to the HOST's core count; inside a container with a CPU quota (the GPU box: 256 host CPUs,
    a share of 16) that oversubscribes and dense CPU work — the float64 Oracle A — runs orders of magnitude slower."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(round(int(q[0]) / int(q[1])))))
    except (OSError, ValueError, IndexError):
        pass
    torch.set_num_threads(max(1, n))
    return n


def float64_grads(cams_, xyz, opacity, scaling, rotation, shs, use_rgb, sh_degree, blend, dimg_, H, W):
    """The same gradients from Oracle A (dense float64 autograd, oracle/oracle_torch.py): the referee between two float32 programs."""
    from oracle import oracle_torch as OT
    d = torch.float64
    leaves = {n: x.to(d).clone().requires_grad_(True) for n, x in dict(xyz=xyz, opacity=opacity.reshape(-1, 1), scaling=scaling, rotation=rotation, shs=shs).items()}
    bl = {k: v.to(d).clone().requires_grad_(True) for k, v in blend.items()}
    cams_ = cams_.to(d)
    tot = 0
    for v in range(cams_.shape[0]):
        c = cams_[v]
        means, opac, cols, sh = OT.blend_attributes(leaves["xyz"], leaves["opacity"], leaves["shs"], use_rgb=use_rgb,
                                                    **{k: (x.reshape(-1, 1) if k == "opacity_b" else x) for k, x in bl.items()})
        kw = dict(colors_precomp=cols) if use_rgb else dict(shs=sh, sh_degree=sh_degree)
        img, _ = OT.rasterize_dense(means, opac, leaves["scaling"], leaves["rotation"], viewmatrix=c[:16].reshape(4, 4),
                                    projmatrix=c[16:32].reshape(4, 4), campos=c[32:35], tanfovx=float(c[35]), tanfovy=float(c[36]),
                                    bg=c[37:40], H=H, W=W, **kw)
        tot = tot + (img * dimg_[v].to(d)).sum()
    tot.backward()
    out = dict(means3D=leaves["xyz"].grad, opacities=leaves["opacity"].grad, scales=leaves["scaling"].grad, rotations=leaves["rotation"].grad)
    out["colors_precomp" if use_rgb else "shs"] = leaves["shs"].grad
    out.update({k: v.grad for k, v in bl.items()})
    return {k: (torch.zeros_like(leaves["xyz"][:0]) if v is None else v) for k, v in out.items()}


# ---- stand-ins for the sub-modules forward_single_batch calls (renderer_one_shot.py:448-512) -------------------------------
# Shared by tests/golden/make_batch_fixture.py (which hands them to the REFERENCE's own forward_single_batch) and by
# tests/test_gpu_single_batch.py (which hands them to the composed path): the networks themselves are out of scope, what is
# pinned is the composition around them. Weights come from a fixed seed; the validity score is read straight from feature
# column 0, so that the two thresholds decide on bit-identical values on every device.
class BatchStandIns:
    C = 12                               # feature channels

    def __init__(self, device="cpu", use_rgb=True, seed=11):
        g = torch.Generator().manual_seed(seed)
        mk = lambda *s: (0.4 * torch.randn(*s, generator=g)).to(device)
        self.W = {"xyz": mk(self.C, 3), "scaling": 0.5 * mk(self.C, 3), "rotation": mk(self.C, 4), "opacity": 3.0 * mk(self.C, 1),
                  "shs": mk(self.C, 3 if use_rgb else 48)}
        self.Wr = mk(self.C, 3)
        self.use_rgb = use_rgb
        self.threshold_low, self.threshold_high = 0.1, 0.9

    def gs_valid(self, feat, pts):
        return feat[:, 0:1]

    def vert_pos_refinement(self, feat, pts):
        return pts + 0.003 * torch.tanh(feat @ self.Wr)

    def forward_gs(self, feat, pts):
        from guassianhand_amd.renderer import gs_activations
        raw = {k: feat @ w for k, w in self.W.items()}
        raw["scaling"] = raw["scaling"] - 5.0
        return gs_activations(raw, pts, use_rgb=self.use_rgb)

    @staticmethod
    def get_uvd(pts, vert3d_uv0, face_uv, face_uv_xy):
        """(uv in [0,1] x [0,0.5], distance, intermediates) like livehand.input_encoder.get_uvd's return triple."""
        uv = torch.stack([torch.sigmoid(25.0 * pts[:, 0]), 0.5 * torch.sigmoid(25.0 * pts[:, 1])], 1)
        return uv.detach(), pts[:, 2].detach(), None

    def namespace(self, device, sh_degree=3, scaling_modifier=1.0, radius_texture=1.0):
        """The attributes forward_single_batch / forward_single_view read of the renderer object."""
        from types import SimpleNamespace
        return SimpleNamespace(gs_valid=self.gs_valid, vert_pos_refinement=self.vert_pos_refinement, forward_gs=self.forward_gs,
                               threshold_low=self.threshold_low, threshold_high=self.threshold_high, device=torch.device(device),
                               cfg=SimpleNamespace(scaling_modifier=scaling_modifier, sh_degree=sh_degree, radius_texture=radius_texture),
                               gs_net=SimpleNamespace(cfg=SimpleNamespace(use_rgb=self.use_rgb)), get_uvd=self.get_uvd)


def batch_inputs(N=400, n_views=2, H=64, W=48, map_hw=(16, 32), seed=5):
    g = torch.Generator().manual_seed(seed)
    feat = torch.rand(N, BatchStandIns.C, generator=g)
    feat[:, 0] = torch.rand(N, generator=g)
    feat[:, 0][(feat[:, 0] - 0.1).abs() < 1e-4] = 0.2          # keep the score off the thresholds
    feat[:, 0][(feat[:, 0] - 0.9).abs() < 1e-4] = 0.8
    pts = 0.05 * torch.randn(N, 3, generator=g)
    K = torch.eye(4).repeat(n_views, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 180.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    w2c = torch.eye(4).repeat(n_views, 1, 1)
    for v in range(n_views):
        a = 0.35 * v
        w2c[v, :3, :3] = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]])
        w2c[v, :3, 3] = torch.tensor([0.01 * v, -0.01, 1.0])
    return dict(feat=feat, pts=pts, w2cs=w2c, Ks=K, H=H, W=W, bg=torch.tensor([0.1, 0.2, 0.3]),
                color_w=1 + 0.05 * torch.randn(48, generator=g), xyz_b=0.004 * torch.randn(3, generator=g),
                color_b=0.05 * torch.randn(48, *map_hw, generator=g), opacity_b=0.05 * torch.randn(1, *map_hw, generator=g))


def edit_batch_inputs(N=400, n_views=2, seed=9):
    """batch_inputs for the EDIT renderer's forward_single_batch (renderer_one_shot_edit.py:440-520): maps 2048 texels wide (render_edit
    slices them at column 1024), and a handful of points whose U lands between the map's columns 1023 and 1024 — the seam of the
    two hands' colour weights, where the bilinear lookup mixes the left and the right constants."""
    d = batch_inputs(N=N, n_views=n_views, map_hw=(8, 2048), seed=seed)
    fr = torch.tensor([0.0, 0.125, 0.25, 0.5, 0.75, 0.999, 1.0, -0.3])
    u = (1023.0 + fr) / 2047.0                                   # BatchStandIns.get_uvd: u = sigmoid(25 x)
    d["pts"][:fr.numel(), 0] = torch.log(u / (1 - u)) / 25.0
    d["feat"][:fr.numel(), 0] = 0.5                                # kept (> threshold_low), not duplicated
    return d
