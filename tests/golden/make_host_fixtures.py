#!/usr/bin/env python
"""Generate tests/golden/host_fixtures.npz by running the REFERENCE's own host-side Python.

Runs only in the build container (needs /root/reference); the resulting .npz is data (inputs +
expected outputs) and is committed; nothing of the reference travels. It imports
tgs.models.renderer_one_shot under stub modules for the reference's missing third-party deps and
a *recording fake* `diff_gaussian_rasterization`, then captures

  1. camera fixture   — Camera.from_w2c + forward_single_view settings (renderer_one_shot.py:61-112, :276-294)
  2. blend fixture    — the tensors that arrive at the rasteriser for both passes (:298-334, :353-379),
                        RGB mode and SH mode, color_w as (48,) and as (P,48)
  3. call protocol    — keyword names, None-ness, shapes, dtypes, number of calls (:338-346, :372-379)
  4. GSLayer fixture  — activations incl. trunc_exp and the 1.2/32 restricted offset (:191-214)

Usage: python tests/golden/make_host_fixtures.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_fixtures.npz")


class _Dummy:
    """Stands in for any class / function / constant of a stubbed third-party module."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, k):
        return _Dummy()

    def __getitem__(self, k):
        return self

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    __path__ = []   # looks like a package so `import a.b.c` works

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy()


class _StubFinder:
    """Meta-path finder that fabricates empty modules for the reference's missing third-party deps."""
    ROOTS = ("plyfile", "livehand", "trimesh", "jaxtyping", "omegaconf", "pytorch3d", "cv2", "torchvision", "skimage",
             "utils", "lpips", "smplx", "imageio", "torch_scatter")

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


def install_stubs(calls):
    sys.meta_path.insert(0, _StubFinder())

    def stub(name, **attrs):
        m = _StubModule(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    from typing import NamedTuple

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    class GaussianRasterizer(torch.nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, **kw):
            calls.append((self.raster_settings, kw))
            H, W = self.raster_settings.image_height, self.raster_settings.image_width
            return torch.zeros(3, H, W), torch.zeros(kw["means3D"].shape[0], dtype=torch.int32)

    stub("diff_gaussian_rasterization", GaussianRasterizationSettings=GaussianRasterizationSettings,
         GaussianRasterizer=GaussianRasterizer)


def main():
    calls = []
    install_stubs(calls)
    sys.path.insert(0, REF)
    import tgs.models.renderer_one_shot as ref  # noqa: E402

    out = {}
    g = torch.Generator().manual_seed(7)

    # ---- 1. camera fixtures ------------------------------------------------------------------
    cam_cases = [
        dict(f=(1500.0, 1500.0), c=(167.0, 256.0), s=0.0, H=512, W=334),
        dict(f=(1300.0, 1250.0), c=(150.5, 270.25), s=0.0, H=512, W=334),     # off-centre principal point
        dict(f=(650.0, 640.0), c=(120.0, 130.0), s=2.5, H=256, W=256),        # non-zero skew
        dict(f=(2600.0, 2600.0), c=(512.0, 512.0), s=0.0, H=1024, W=1024),
    ]
    for i, cc in enumerate(cam_cases):
        K = torch.eye(4)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1] = cc["f"][0], cc["f"][1], cc["c"][0], cc["c"][1], cc["s"]
        A = torch.randn(3, 3, generator=g)
        Q, _ = torch.linalg.qr(A)
        if torch.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        w2c = torch.eye(4)
        w2c[:3, :3] = Q
        w2c[:3, 3] = torch.tensor([0.05, -0.03, 1.0]) + 0.1 * torch.randn(3, generator=g)
        cam = ref.Camera.from_w2c(w2c=w2c, intrinsic=K, height=cc["H"], width=cc["W"], znear=0.71, zfar=1.42)
        import math
        out[f"cam{i}_K"] = K.numpy()
        out[f"cam{i}_w2c"] = w2c.numpy()
        out[f"cam{i}_HW"] = np.array([cc["H"], cc["W"]])
        out[f"cam{i}_viewmatrix"] = cam.world_view_transform.numpy()
        out[f"cam{i}_projmatrix"] = cam.full_proj_transform.float().numpy()
        out[f"cam{i}_campos"] = cam.camera_center.numpy()
        out[f"cam{i}_tanfov"] = np.array([math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)], dtype=np.float64)
        out[f"cam{i}_znear_zfar"] = np.array([cam.znear, cam.zfar])
    out["n_cams"] = np.array(len(cam_cases))

    # ---- 2+3. blend + call-protocol fixtures via the reference's forward_single_view ------------
    P = 37
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 1300.0
    K[0, 2], K[1, 2] = 167.0, 256.0
    w2c = torch.eye(4)
    w2c[2, 3] = 1.0
    cam = ref.Camera.from_w2c(w2c=w2c, intrinsic=K, height=512, width=334, znear=0.71, zfar=1.42)
    modes = [("rgb_w48", True, False), ("rgb_wP48", True, True), ("sh_w48", False, False), ("sh_wP48", False, True),
             ("rgb_noblend", True, None), ("sh_wonly", False, "wonly")]
    for name, use_rgb, wmode in modes:
        gs = ref.GaussianModel(
            xyz=0.1 * torch.randn(P, 3, generator=g), opacity=torch.rand(P, 1, generator=g),
            rotation=torch.nn.functional.normalize(torch.randn(P, 4, generator=g)),
            scaling=torch.exp(-5 + 0.3 * torch.randn(P, 3, generator=g)),
            shs=torch.rand(P, 1, 3, generator=g) if use_rgb else 0.3 * torch.randn(P, 16, 3, generator=g))
        if wmode is None:
            color_w = color_b = opacity_b = xyz_b = None
        else:
            color_w = 1 + 0.05 * torch.randn(P, 48, generator=g) if wmode is True else 1 + 0.05 * torch.randn(48, generator=g)
            color_b = None if wmode == "wonly" else 0.02 * torch.randn(P, 48, generator=g)
            opacity_b = 0.02 * torch.randn(P, 1, generator=g)
            xyz_b = 0.01 * torch.randn(3, generator=g)
        self_ns = types.SimpleNamespace(
            device=torch.device("cpu"),
            cfg=types.SimpleNamespace(scaling_modifier=1.0, sh_degree=3),
            gs_net=types.SimpleNamespace(cfg=types.SimpleNamespace(use_rgb=use_rgb)))
        calls.clear()
        bg = torch.tensor([0.0, 0.0, 0.0])
        ret = ref.GS3DRenderer.forward_single_view(self_ns, gs, cam, bg, color_w=color_w, xyz_b=xyz_b, color_b=color_b,
                                                   opacity_b=opacity_b)
        for k in ("xyz", "opacity", "rotation", "scaling", "shs"):
            out[f"{name}_gs_{k}"] = getattr(gs, k).numpy()
        for k, v in (("color_w", color_w), ("color_b", color_b), ("opacity_b", opacity_b), ("xyz_b", xyz_b)):
            if v is not None:
                out[f"{name}_{k}"] = v.numpy()
        out[f"{name}_ncalls"] = np.array(len(calls))
        out[f"{name}_ret_keys"] = np.array(sorted(ret.keys()))
        out[f"{name}_ret_rgb_shape"] = np.array(ret["comp_rgb"].shape)
        for ci, (rs, kw) in enumerate(calls):
            out[f"{name}_call{ci}_kwnames"] = np.array(sorted(kw.keys()))
            out[f"{name}_call{ci}_none"] = np.array(sorted(k for k, v in kw.items() if v is None))
            out[f"{name}_call{ci}_sh_degree"] = np.array(rs.sh_degree)
            out[f"{name}_call{ci}_bg"] = rs.bg.numpy()
            out[f"{name}_call{ci}_settings_fields"] = np.array(rs._fields)
            out[f"{name}_call{ci}_hw"] = np.array([rs.image_height, rs.image_width])
            for k, v in kw.items():
                if v is not None:
                    out[f"{name}_call{ci}_{k}"] = v.detach().numpy()
                    out[f"{name}_call{ci}_{k}_dtype"] = np.array(str(v.dtype))
    out["blend_modes"] = np.array([m[0] for m in modes])

    # ---- 4. GSLayer activations -------------------------------------------------------------------
    from tgs.utils.ops import trunc_exp
    x = torch.linspace(-8, 20, 29).requires_grad_(True)
    y = trunc_exp(x)
    y.sum().backward()
    out["trunc_exp_x"] = x.detach().numpy()
    out["trunc_exp_y"] = y.detach().numpy()
    out["trunc_exp_grad"] = x.grad.numpy()
    v = torch.randn(11, 3, generator=g)
    pts = torch.randn(11, 3, generator=g)
    out["offset_v"] = v.numpy()
    out["offset_pts"] = pts.numpy()
    out["offset_out"] = ((torch.sigmoid(v) - 0.5) * (1.2 / 32) + pts).numpy()   # renderer_one_shot.py:208-211

    # ---- 4b. GSLayer.forward itself (renderer_one_shot.py:191-214), called unbound on a stand-in `self` that carries what the
    # method reads: cfg.feature_channels / use_rgb / restrict_offset / xyz_offset / clip_scaling and the linear heads.
    from types import SimpleNamespace
    N, Cin = 13, 16
    x_feat = torch.randn(N, Cin, generator=g)
    pts2 = torch.randn(N, 3, generator=g)
    out["gslayer_x"], out["gslayer_pts"] = x_feat.numpy(), pts2.numpy()
    for tag, use_rgb, restrict, clip in (("a", True, True, None), ("b", False, False, 0.02)):
        chans = {"xyz": 3, "scaling": 3, "rotation": 4, "opacity": 1, "shs": 3 if use_rgb else 48}
        layers = torch.nn.ModuleList()
        for k, oc in chans.items():
            lin = torch.nn.Linear(Cin, oc)
            with torch.no_grad():
                lin.weight.copy_(0.3 * torch.randn(oc, Cin, generator=g)); lin.bias.copy_(0.3 * torch.randn(oc, generator=g))
                if k == "scaling":
                    lin.bias.add_(-5.0)                     # around the reference's init_scaling (:165)
            layers.append(lin)
            out[f"gslayer_{tag}_{k}_raw"] = lin(x_feat).detach().numpy()      # the head outputs the activations act on
        self_ns = SimpleNamespace(cfg=SimpleNamespace(feature_channels=chans, use_rgb=use_rgb, restrict_offset=restrict, xyz_offset=True,
                                                      clip_scaling=clip), out_layers=layers)
        gm = ref.GSLayer.forward(self_ns, x_feat, pts2)
        for k in ("xyz", "opacity", "rotation", "scaling", "shs"):
            out[f"gslayer_{tag}_{k}"] = getattr(gm, k).detach().numpy()
        out[f"gslayer_{tag}_cfg"] = np.array([int(use_rgb), int(restrict), -1.0 if clip is None else clip])

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
