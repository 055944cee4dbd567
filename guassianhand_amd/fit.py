"""One-shot fit loop — counterpart of the reference's one-shot optimisation stage.

What it mirrors (all in /root/reference):
  * learnable one-shot parameters: `color_w` (48,), `color_b` (48,1024,2048), `xyz_b` (3,), `opacity_b`
    (1,1024,2048) — infer_one_shot.py:159-163; only color_w / color_b / opacity_b (and network-side map_bias /
    identity codes, out of scope here) are trainable (:340-343);
  * per-Gaussian lookup of the maps at the Gaussians' UV coordinates with bilinear `grid_sample`,
    align_corners=True — renderer_one_shot.py:420-446, :489-492;
  * loss: 10 * L1(rgb) + 1.0 * MSE(clip(mean_c(mask), -0.001, 1), gt_mask) — utils.py:180-252, :282-291 with the
    shipped lambdas (config/one_shot.json:121-132; the 0.1*VGG term needs torchvision weights and is out of scope)
    — plus the regularisers 100*mean|color_b| + mean(opacity_b^2) — infer_one_shot.py:514-519;
  * optimiser: Adam(lr 0.01) + MultiStepLR(milestones [2,5,10,20,35,50,75], gamma 0.5) per epoch — :345-349,
    config/one_shot.json:29.

MI355X form: all of a rank's cameras go through ONE fused launch sequence (RGB + alpha in one pass, blend fused
into the projection kernel); with N ranks the cameras are sharded round-robin and the only collective is one
all-reduce(sum) of the gradient block AT THE RASTERISER BOUNDARY (per-Gaussian blend values + color_w) before it
is back-propagated into the 403 MB maps — instead of PL-DDP all-reducing the maps themselves (:638).

Active-texel mode (default on a ROCm device): the UVs are constant during the fit, so only the <= 4P texels under the
Gaussians' bilinear footprints ever receive an image gradient; every other texel of the zero-initialised maps has
regulariser gradient 100*sign(0)/n = 0 resp. 2*0/n = 0 and Adam leaves it at exactly 0. The fit therefore keeps the
active texels compacted (U,C) (76 MB instead of 403 MB + 806 MB of Adam state), looks them up with gh_uv_gather_*,
and applies regulariser + Adam in ONE fused pass (gh_adam_reg_step). `.color_b` / `.opacity_b` still return the dense
reference-layout maps; tests/test_gpu_fit.py checks the mode against the dense torch.optim.Adam path step by step.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as ghdist
from .renderer import GaussianModel
from .uvmap import (ActiveTexels, AdamReg, adam_group_step, reg_total, to_reference_layout, uv_gather, uv_gather2,
                    uv_gather_backward, uv_gather_backward2, uv_sample)

MILESTONES = (2, 5, 10, 20, 35, 50, 75)


def sample_map(tex: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    """(C,Hm,Wm) map sampled at uv (P,2) in [-1,1] -> (P,C); renderer_one_shot.py:420-446 (query_triplane_texture
    with radius_texture 1.0: positions pass through scale_tensor((-1,1)->(-1,1)) unchanged)."""
    out = F.grid_sample(tex[None], uv[None, :, None, :], align_corners=True, mode="bilinear")   # (1,C,P,1)
    return out[0, :, :, 0].transpose(0, 1)


def fit_loss(comp_rgb, comp_mask, gt_rgb, gt_mask, bbox_mask=None, lambda_l1: float = 10.0, lambda_mloss: float = 1.0):
    """Image part of the reference loss for a stack of views. comp_rgb (Nv,H,W,3), comp_mask (Nv,H,W,3),
    gt_rgb (Nv,H,W,3), gt_mask (Nv,H,W). Returns the SUM over views of the per-view means (callers divide by
    the global number of views so that sharded ranks add up to the single-process loss)."""
    rgb = comp_rgb
    if bbox_mask is not None:                       # infer_one_shot.py:507-510: pixels outside the bbox are zeroed
        rgb = rgb * (bbox_mask[..., None] != 0)
    l1 = lambda_l1 * (rgb - gt_rgb).abs().mean(dim=(1, 2, 3))                       # utils.py:290-294
    alpha = comp_mask.float().mean(-1)                                              # infer_one_shot.py:497
    ml = lambda_mloss * ((alpha.clip(-0.001, 1.0) - gt_mask) ** 2).mean(dim=(1, 2))  # utils.py:249-252
    return (l1 + ml).sum()


class OneShotFit(nn.Module):
    def __init__(self, gs: GaussianModel, uv: torch.Tensor, *, use_rgb: bool = True, sh_degree: int = 3,
                 map_hw: Sequence[int] = (1024, 2048), lr: float = 0.01, render_fn: Optional[Callable] = None,
                 active_texels: Optional[bool] = None, static_geometry: Optional[bool] = None, occlusion_bound: bool = False):
        super().__init__()
        self.gs = GaussianModel(*[t.detach() for t in gs])          # frozen network outputs
        self.register_buffer("uv", uv.detach().float())
        self.use_rgb, self.sh_degree = use_rgb, sh_degree
        dev = gs.xyz.device
        Hm, Wm = map_hw
        self.color_w = nn.Parameter(torch.ones(48, device=dev))                      # infer_one_shot.py:159
        self.xyz_b = nn.Parameter(torch.zeros(3, device=dev), requires_grad=False)   # :161 (not in the trainable set)
        self.map_hw = (Hm, Wm)
        self.lr0, self.epoch = lr, 0
        self.active = uv.is_cuda if active_texels is None else bool(active_texels)
        self._default_render = render_fn is None
        self._cams, self._cams_src, self._mine_src, self._one = None, None, None, None
        # Static geometry (default with the default renderer on a device): `gs` is frozen here (detached network outputs), xyz_b
        # is not trained (:161) and a fit renders the same cameras every step, so projection, both sorts and the record gather
        # are done ONCE; every later step only refreshes opacities / colours in the per-instance records (gh_forward_refresh).
        # Any other camera tensor, image size or a changed `_version` of a geometry tensor rebuilds automatically; after an
        # in-place change through `.data` call invalidate_geometry(). (The reference's map_bias / identity codes move the
        # Gaussians themselves, infer_one_shot.py:340-343: a fit that trains them passes static_geometry=False.)
        # occlusion_bound=True is the policy for the OTHER kind of fit — Gaussians that move a little every step through
        # update_gaussians() (network-side trainables): static lists cannot serve it, the previous step's verified per-tile occlusion
        # depth can (rasterizer.DepthBoundCache: the instances behind it are not listed; exact by verification).
        self._geom_cache = None
        if occlusion_bound:
            if static_geometry:
                raise ValueError("OneShotFit: static_geometry (frozen Gaussians) or occlusion_bound (moving Gaussians), not both")
            from .rasterizer import DepthBoundCache
            self._geom_cache = DepthBoundCache()      # (ignored by calls of fewer than four 512x334 views' worth of pixels)
        elif (static_geometry if static_geometry is not None else (render_fn is None and gs.xyz.is_cuda)):
            from .rasterizer import GeometryCache
            self._geom_cache = GeometryCache()
        self.keep_boundary_grads, self.boundary_grads, self.last_reg = False, None, None
        self._side = None
        if render_fn is None:
            from .renderer import render_views
            render_fn = render_views
        self.render_fn = render_fn
        if self.active:
            # active-texel mode: compact (U,C) storage of the texels the Gaussians can reach, fused regulariser + Adam
            self.texels = ActiveTexels(self.uv, Hm, Wm)
            U = self.texels.U
            # RGB mode reads color_b.view(-1,16,3)[:,0,:] only (renderer_one_shot.py:328): the other 45 channels never get
            # an image gradient and stay 0 like the inactive texels, so only 3 channels are stored (GH_FLAG_BLEND_COLOR_B_RGB)
            self.cb_channels = 3 if use_rgb else 48
            self.color_b_tex = torch.zeros(U, self.cb_channels, device=dev)
            self.opacity_b_tex = torch.zeros(U, 1, device=dev)
            self._adam = {
                "color_w": AdamReg(self.color_w.data, lr),
                "color_b": AdamReg(self.color_b_tex, lr, reg_l1=100.0 / (48 * Hm * Wm)),        # 100*mean|color_b|
                "opacity_b": AdamReg(self.opacity_b_tex, lr, reg_l2=1.0 / (Hm * Wm)),          # mean(opacity_b^2)
            }
            return
        # the maps are stored CHANNEL-LAST (Hm,Wm,C) for the device lookup (uvmap.py); `.color_b` / `.opacity_b` are
        # views in the reference's (C,Hm,Wm) layout (infer_one_shot.py:160,163) for loading / exporting state
        self.color_b_map = nn.Parameter(torch.zeros(Hm, Wm, 48, device=dev))
        self.opacity_b_map = nn.Parameter(torch.zeros(Hm, Wm, 1, device=dev))
        self.opt = torch.optim.Adam([self.color_w, self.color_b_map, self.opacity_b_map], lr=lr)
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=list(MILESTONES), gamma=0.5)

    @property
    def color_b(self) -> torch.Tensor:
        """(48,Hm,Wm), the layout of the reference parameter (infer_one_shot.py:160)."""
        if not self.active:
            return to_reference_layout(self.color_b_map)
        d = self.texels.dense(self.color_b_tex)
        if d.shape[-1] != 48:
            d = torch.cat([d, d.new_zeros(*d.shape[:2], 48 - d.shape[-1])], -1)
        return to_reference_layout(d)

    @property
    def opacity_b(self) -> torch.Tensor:
        """(1,Hm,Wm) (infer_one_shot.py:163)."""
        return to_reference_layout(self.texels.dense(self.opacity_b_tex) if self.active else self.opacity_b_map)

    def load_maps(self, color_b: torch.Tensor, opacity_b: torch.Tensor) -> None:
        """Load reference-layout maps. Active-texel mode requires them to be zero outside the active texels (true for
        the reference's zero initialisation and for anything this class exported); otherwise use active_texels=False."""
        cb, ob = color_b.permute(1, 2, 0).contiguous(), opacity_b.permute(1, 2, 0).contiguous()
        if not self.active:
            with torch.no_grad():
                self.color_b_map.copy_(cb); self.opacity_b_map.copy_(ob)
            return
        ct, ot = self.texels.compact(cb), self.texels.compact(ob)
        if float((self.texels.dense(ct) - cb).abs().max()) != 0.0 or float((self.texels.dense(ot) - ob).abs().max()) != 0.0 \
                or float(ct[:, self.cb_channels:].abs().max() if self.cb_channels < 48 else 0.0) != 0.0:
            raise ValueError("maps are non-zero outside the active texels / channels: construct OneShotFit(active_texels=False)")
        self.color_b_tex.copy_(ct[:, :self.cb_channels]); self.opacity_b_tex.copy_(ot)

    # -- pieces ----------------------------------------------------------------------------------------
    def _packed_cameras(self, w2cs, Ks, H, W, bg) -> torch.Tensor:
        """Packed camera records of the caller's (w2cs, Ks, bg). The cache holds STRONG references to the tensors it was
        built from and is hit only by those very objects, unmodified (`is` + `_version`): a caller that allocates fresh
        camera tensors every step (as the reference's training_step does with its batch) gets them repacked — an address
        recycled by the caching allocator can never alias a stale entry."""
        src = self._cams_src
        if src is not None and src[0] is w2cs and src[1] is Ks and src[2] is bg and src[3] == (H, W) and \
                src[4] == (w2cs._version, Ks._version, bg._version):
            return self._cams
        from .camera import pack_cameras_from_w2c
        self._cams = pack_cameras_from_w2c(w2cs, Ks, H, W, bg)
        self._cams_src = (w2cs, Ks, bg, (H, W), (w2cs._version, Ks._version, bg._version))
        return self._cams

    def blend_values(self) -> Dict[str, torch.Tensor]:
        """Per-Gaussian blend values: the device UV lookup of renderer_one_shot.py:489-492 (gh_uv_sample_* /
        gh_uv_gather_*)."""
        if self.active:
            cb, ob = uv_gather2(self.color_b_tex, self.opacity_b_tex, self.texels)       # both maps in one launch
            return dict(color_w=self.color_w, color_b=cb, opacity_b=ob, xyz_b=self.xyz_b)
        return dict(color_w=self.color_w, color_b=uv_sample(self.color_b_map, self.uv),
                    opacity_b=uv_sample(self.opacity_b_map, self.uv), xyz_b=self.xyz_b)

    def regulariser(self) -> torch.Tensor:
        if self.active:
            Hm, Wm = self.map_hw
            return 100.0 * self.color_b_tex.abs().sum() / (48 * Hm * Wm) + self.opacity_b_tex.pow(2.0).sum() / (Hm * Wm)
        return 100.0 * self.color_b_map.abs().mean() + self.opacity_b_map.pow(2.0).mean()   # infer_one_shot.py:514-518

    def update_gaussians(self, gs: GaussianModel) -> None:
        """New Gaussians for the next steps (same count and UVs): the path for a caller whose NETWORK-side trainables move the
        Gaussians between steps (the reference's map_bias / identity codes, infer_one_shot.py:340-343, :517-519 — out of this
        repository's scope, they live in the reference's networks). The blend maps, their Adam state and the active-texel index
        are unaffected (they are indexed by the constant UVs); the static tile lists are dropped, so the next step is a full
        forward. A fit that calls this every step should be constructed with static_geometry=False."""
        if gs.xyz.shape != self.gs.xyz.shape:
            raise ValueError("update_gaussians: the number of Gaussians is fixed (the UV lookup is indexed by it)")
        self.gs = GaussianModel(*[t.detach() for t in gs])
        from .rasterizer import DepthBoundCache
        if not isinstance(self._geom_cache, DepthBoundCache):      # (an occlusion bound survives a small move: that is its purpose)
            self.invalidate_geometry()

    skipped_steps = 0      # sync-free steps a miss of the speculative occlusion bound turned into no-ops (counted by step())

    def invalidate_geometry(self) -> None:
        """Forget the static tile lists (after modifying a geometry tensor in place through `.data`)."""
        if self._geom_cache is not None:
            self._geom_cache.clear()

    def render(self, w2cs, Ks, H, W, bg, blend: Dict[str, torch.Tensor], sync: bool = True):
        kw = {}
        if self._default_render:                                      # the camera records of a fit never change: pack them once
            kw["cams"] = self._packed_cameras(w2cs, Ks, H, W, bg)
            if self._geom_cache is not None:
                kw["geometry_cache"] = self._geom_cache
        return self.render_fn(self.gs, w2cs, Ks, H, W, bg, color_w=blend["color_w"], xyz_b=blend["xyz_b"],
                              color_b=blend["color_b"], opacity_b=blend["opacity_b"], use_rgb=self.use_rgb,
                              sh_degree=self.sh_degree, sync=sync, **kw)

    def _is_depth_bound_cache(self) -> bool:
        from . import rasterizer as R
        return isinstance(self._geom_cache, R.DepthBoundCache)

    # -- one optimisation step over all cameras (sharded over ranks) ------------------------------------
    def step(self, w2cs, Ks, H: int, W: int, bg, gt_rgb, gt_mask, bbox_mask=None, sync: bool = True) -> torch.Tensor:
        """w2cs/Ks/gt_* hold ALL Nv cameras on every rank; each rank renders views v % world == rank."""
        rank = torch.distributed.get_rank() if ghdist.dist.is_initialized() else 0
        world = torch.distributed.get_world_size() if ghdist.dist.is_initialized() else 1
        n_total = w2cs.shape[0]
        mine = ghdist.shard_views(n_total, rank, world)
        if not self.active:
            self.opt.zero_grad(set_to_none=True)

        if not sync and self.color_w.is_cuda and self._is_depth_bound_cache():
            # Moving geometry through a DepthBoundCache, sync-free: a miss of the speculative bound is an EXPECTED event
            # there — the device-side guard turned that step into a no-op (NaN loss, untouched parameters) and the cache dropped the
            # bound. Look at the read-backs that have arrived, count the skipped step instead of leaving it unseen until somebody
            # calls check_overflow(), and go on: this step renders without a bound. Any other verdict (capacity, stale lists) raises.
            from . import rasterizer as R
            try:
                R.check_overflow(block=False)
            except R.GhDepthBoundMiss:
                self.skipped_steps += 1
        blend = self.blend_values()                                   # graph A: maps -> per-Gaussian values
        names = ["color_w", "color_b", "opacity_b"] if self.active else \
            [k for k in ("color_w", "color_b", "opacity_b") if blend[k].requires_grad]
        leaves = {k: (blend[k].detach().requires_grad_(True) if k in names else blend[k]) for k in blend}
        if not mine:                                                  # more ranks than cameras: this rank only joins the reduction
            loss_img = torch.zeros((), device=self.color_w.device)
            grads = {k: torch.zeros_like(leaves[k]) for k in names}
        else:                                                      # graph B: rasteriser + image loss
            allv = len(mine) == n_total
            sel = (lambda t: t) if len(mine) == n_total else (lambda t: t[mine])
            out = None
            if self._default_render and self.color_w.is_cuda and allv:
                # render + image loss as ONE autograd node (loss.rendered_fit_loss): dL/dloss is applied inside the render
                # backward instead of in two elementwise passes over the images
                from .loss import rendered_fit_loss
                loss_img, _, _ = rendered_fit_loss(self._packed_cameras(w2cs, Ks, H, W, bg), self.gs.xyz, self.gs.opacity, self.gs.scaling, self.gs.rotation,
                                                   self.gs.shs, gt_rgb, gt_mask, None if bbox_mask is None else bbox_mask.float(),
                                                   scale=1.0 / n_total, H=H, W=W, use_rgb=self.use_rgb, sh_degree=self.sh_degree,
                                                   xyz_b=leaves["xyz_b"], opacity_b=leaves["opacity_b"], color_w=leaves["color_w"],
                                                   color_b=leaves["color_b"], sync=sync, geometry_cache=self._geom_cache,
                                                   defer_loss=True)      # (the value is looked at behind the backward only: its final sum
                                                                         #  rides in the render backward, GH_FLAG_DEFER_LOSS_SUM)
            else:
                if allv:
                    w_sel, k_sel = w2cs, Ks
                else:                                 # this rank's cameras: sliced once, so that the packed records and the
                    mk = (w2cs, Ks, w2cs._version, Ks._version, tuple(mine))       # static tile lists are found again next step
                    ms = self._mine_src
                    if ms is None or ms[0][0] is not w2cs or ms[0][1] is not Ks or ms[0][2:] != mk[2:]:
                        self._mine_src = ms = (mk, w2cs[mine].contiguous(), Ks[mine].contiguous())
                    w_sel, k_sel = ms[1], ms[2]
                out = self.render(w_sel, k_sel, H, W, bg, leaves, sync=sync)
            if out is None:
                pass
            elif "image_chw" in out and out["image_chw"].is_cuda:    # fused loss + gradients on the rasteriser's layouts
                from . import rasterizer as R
                from .loss import fit_image_loss
                loss_img = fit_image_loss(out["image_chw"], out["alpha"], sel(gt_rgb), sel(gt_mask),
                                          None if bbox_mask is None else sel(bbox_mask).float(), scale=1.0 / n_total,
                                          guard=R.last_guard() if self._default_render else None)
            else:
                loss_img = fit_loss(out["comp_rgb"], out["comp_mask"], sel(gt_rgb), sel(gt_mask),
                                    None if bbox_mask is None else sel(bbox_mask)) / n_total
            if self._one is None or self._one.device != loss_img.device:
                self._one = torch.ones((), dtype=torch.float32, device=loss_img.device)
            g = torch.autograd.grad(loss_img, [leaves[k] for k in names], grad_outputs=self._one, allow_unused=True)
            grads = {k: (gi if gi is not None else torch.zeros_like(leaves[k])) for k, gi in zip(names, g)}
        # Which collective runs must be the same decision on every rank: the in-place block path needs every rank to have
        # rendered (its buffer is the one the backward kernels wrote), so it is taken only when every rank owns a camera
        # (n_total >= world); with more ranks than cameras all ranks pack the same [loss | overflow | grads] buffer instead.
        use_block = world > 1 and self._default_render and self.color_w.is_cuda and self.active and n_total >= world
        # Device-side overflow guard of THIS rank's render (sync-free / graph replay): its loss kernel has emitted NaN and zero
        # gradients. In a sharded fit the flag is summed with the gradients, so that every rank skips the same Adam step.
        local_guard = None
        if self._default_render and self.color_w.is_cuda and mine:
            from . import rasterizer as R
            local_guard = R.last_guard()
        ovf = None
        if world > 1 and self.color_w.is_cuda:
            ovf = torch.zeros((), device=self.color_w.device) if local_guard is None else \
                (local_guard.view(torch.int32)[1] & 15).ne(0).float()      # the error bits (GH_COUNTER_ERROR_MASK); bit 4 is information
        blk = None
        if use_block:
            from . import rasterizer as R
            blk = R.last_grad_block()
            if not all(grads[k].data_ptr() == blk[0].data_ptr() + 4 * a for k, _, a, _ in blk[2] if k in grads):
                raise RuntimeError("sharded fit: the blend gradients are not views of the rasteriser's gradient block")
        if world == 1:
            loss_tot, red = loss_img.detach(), grads
        elif blk is not None:
            # The backward kernels wrote the blend gradients into ONE contiguous block; its leading floats are reserved for
            # the loss (float 0) and the overflow flag (float 1): all-reduce that prefix in place on a side stream, right
            # behind the backward (no packing pass).
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.color_w.device)
            blk[0][1:2].copy_(ovf.reshape(1))
            work, buf = ghdist.allreduce_block(blk[0], blk[3], loss_img.detach(), stream=self._side)
            if work is not None:
                work.wait()
            torch.cuda.current_stream().wait_stream(self._side)
            loss_tot, ovf = buf[0], buf[1]
            red = {k: blk[0][a:a + n].view(grads[k].shape) for k, shp, a, n in blk[2] if k in grads}
        else:
            small = dict(grads)
            wide = self.use_rgb and "color_b" in grads and grads["color_b"].shape[1] == 48   # RGB mode touches color_b[:, 0:3] only (:328)
            if wide:
                small["color_b"] = grads["color_b"][:, :3].contiguous()
            if ovf is not None:
                small["~overflow"] = ovf.reshape(1)
            loss_tot, red = ghdist.allreduce_grads(small, loss_img.detach(), sorted(small))
            red = dict(red)
            if ovf is not None:
                ovf = red.pop("~overflow")[0]
            if wide:
                full = torch.zeros_like(grads["color_b"])
                full[:, :3] = red["color_b"]
                red["color_b"] = full
        if self.keep_boundary_grads:                                  # tests: the reduced gradient block at the rasteriser boundary
            self.boundary_grads = {k: v.detach().clone() for k, v in red.items()}
        if self.active:                                               # maps: scatter, then regulariser + Adam in one pass
            uv_gather_backward2(red["color_b"], red["opacity_b"], self.texels, self._adam["color_b"].grad, self._adam["opacity_b"].grad)
            lr = self.lr0 * 0.5 ** sum(1 for m in MILESTONES if m <= self.epoch)
            guard = local_guard
            if ovf is not None:                                       # sharded: the SUM of the ranks' overflow flags, as GhCounters
                guard = torch.zeros(4, dtype=torch.int32, device=self.color_w.device)
                guard[1] = ovf.ne(0).to(torch.int32)
            gw = red["color_w"]                                       # (the kernels' own buffer when it can be used as it stands)
            gw = gw if (gw.is_contiguous() and gw.dtype is torch.float32 and gw.numel() == 48) else None
            if gw is None:
                self._adam["color_w"].grad.copy_(red["color_w"])
            for a in self._adam.values():
                a.lr = lr
            adam_group_step(list(self._adam.values()), guard, [gw if k == "color_w" else None for k in self._adam])   # one launch
            Hm, Wm = self.map_hw
            # loss = image loss + 100 * mean|color_b| + mean(opacity_b^2) of the values this step started from (:514-519)
            tot = reg_total(self._adam["color_b"], 0, 100.0 / (48 * Hm * Wm), self._adam["opacity_b"], 1, 1.0 / (Hm * Wm), loss_tot)
            self.last_reg = tot[1]
            return tot[0]
        reg = self.regulariser()                                      # identical on every rank: never reduced
        self.last_reg = reg.detach()
        torch.autograd.backward([blend[k] for k in names] + [reg], [red[k] for k in names] + [torch.ones_like(reg)])
        self.opt.step()
        return loss_tot + reg.detach()

    def end_epoch(self) -> None:
        self.epoch += 1
        if not self.active:
            self.sched.step()

    def captured(self, w2cs, Ks, H: int, W: int, bg, gt_rgb, gt_mask, bbox_mask=None) -> "CapturedFitStep":
        """The step on these (fixed) cameras and targets as a replayable HIP graph: see CapturedFitStep."""
        return CapturedFitStep(self, (w2cs, Ks, H, W, bg, gt_rgb, gt_mask, bbox_mask))


class CapturedFitStep:
    """One fit step (lookup -> render -> loss -> backward -> scatter -> regulariser + Adam) captured in a HIP graph and
    replayed: ~45 kernel launches become one graph launch, so the step runs at the speed of the GPU however slow the host is
    (the fit of infer_one_shot.py repeats the same cameras and targets every step). Requirements: active-texel mode, the default
    renderer, one process (collectives stay outside graphs), tensors that stay in place. Construction runs TWO regular steps
    (one sizes the instance capacity, one warms the capture stream up); every `replay()` is one more step and returns its
    loss (a device tensor that the next replay overwrites). The learning rate is a kernel argument: when the epoch count moves
    it across a milestone, the next replay re-captures (no extra step). Nothing on the host can see an instance-capacity
    overflow inside a replay; the device-side guard turns such a step into a no-op with a NaN loss, and `check()` (a host
    read-back) raises GhOverflowError with the capacity raised — construct a new CapturedFitStep then."""

    def __init__(self, fit: OneShotFit, args):
        if not (fit.active and fit._default_render and fit.color_w.is_cuda):
            raise ValueError("CapturedFitStep needs the active-texel mode and the default renderer on a ROCm device")
        if ghdist.dist.is_initialized() and torch.distributed.get_world_size() > 1:
            raise ValueError("CapturedFitStep is single-process: the gradient all-reduce of a sharded fit stays outside graphs")
        self.fit, self.args = fit, args
        self.graph, self.loss, self.lr = None, None, None
        fit.step(*args, sync=True)                                   # sizes the instance capacity (a regular step)
        from . import rasterizer as R
        R.check_overflow()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fit.step(*args, sync=False)                              # warm-up of the capture path (a regular step)
        torch.cuda.current_stream().wait_stream(side)
        self._capture()

    def _lr(self) -> float:
        return self.fit.lr0 * 0.5 ** sum(1 for m in MILESTONES if m <= self.fit.epoch)

    def _capture(self) -> None:
        from . import rasterizer as R
        R.set_graph_mode(True)
        try:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = self.fit.step(*self.args, sync=False)    # (capturing enqueues nothing: this is not a step)
            self.counters = R.graph_counters()
        finally:
            R.set_graph_mode(False)
        self.lr = self._lr()
        # The captured refresh reads the static tile lists out of the geometry owner's workspace by ADDRESS. Hold the owner
        # here: were it only the cache's (cleared by any overflow in the process, invalidate_geometry(), update_gaussians()),
        # its workspace would go back to the pool and the next forward of that size would overwrite the lists under the graph.
        gc = self.fit._geom_cache
        self._geom = getattr(gc, "ctx", None)                     # (a DepthBoundCache holds no lists and is not used inside graphs)

    def _stale(self) -> bool:
        gc = self.fit._geom_cache
        return gc is not None and hasattr(gc, "ctx") and gc.ctx is not self._geom

    def replay(self) -> torch.Tensor:
        if self._stale():
            # the cache was cleared or rebuilt since the capture: the graph would render the OLD lists (kept alive above, so
            # not a fault — but not what the fit holds now). This step runs as a regular one (it rebuilds the lists), then capture again.
            # (Checked BEFORE the learning rate: a capture over an empty cache records the BUILD — a full forward — and every replay
            # of that graph would rebuild the lists instead of refreshing them; found by tools/fuzz_fit.py.)
            from . import rasterizer as R
            loss = self.fit.step(*self.args, sync=True).detach().clone()   # THIS replay's step, run eagerly (at the current learning rate)
            R.check_overflow()
            self._capture()
            return loss
        if self._lr() != self.lr:
            self._capture()
        self.graph.replay()
        return self.loss

    def check(self) -> None:
        """Host read-back of the captured render's counters: raises rasterizer.GhOverflowError if a replay overflowed."""
        from . import rasterizer as R
        for counters, cap, key, full in self.counters:
            c4 = counters.tolist()
            # stale static lists (an opacity above their bound) / GH_FLAG_DEPTH24 not holding / an instance overflow: the rasteriser
            # learns what the word says (capacity, verdicts, caches cleared) and raises the matching error
            R.report_counter_word(key, c4[1], c4[0] & 0xFFFFFFFF, cap, c4[2] & 0xFFFFFFFF, dev=counters.device,
                                  where=(" [inside the captured fit step: the next replay() rebuilds and re-captures]" if (c4[1] & 2) else
                                         " [inside the captured fit step: construct a new CapturedFitStep]" if (c4[1] & 8) else
                                         " inside the captured fit step"), learn24=full)
