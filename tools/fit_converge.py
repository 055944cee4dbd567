"""The one-shot fit at BASELINE configs[3]'s full size on one GPU: 8 ring cameras, P = 98,562 two-hand Gaussians, 1024x2048
blend maps learned from target images rendered with known maps; 300 steps replayed from a captured HIP graph
(fit.CapturedFitStep), the learning-rate milestones crossed. Prints the loss curve and the image error against the targets."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from guassianhand_amd import fit as F, rasterizer as R
from guassianhand_amd.renderer import GaussianModel
from guassianhand_amd.scenes import make_scene
dev = torch.device("cuda:0")
sc = make_scene("two_hands", n_views=8, blend=False).to(dev)
g = torch.Generator().manual_seed(4)
uv = (torch.rand(sc.P, 2, generator=g) * 2 - 1).to(dev)
gs = GaussianModel(sc.xyz, sc.opacity, sc.rotation, sc.scaling, sc.shs)
map_hw = (1024, 2048)
# smooth "true" maps (low-resolution noise, upsampled) so that neighbouring Gaussians agree on them
lo = lambda c, s_: torch.nn.functional.interpolate(s_ * torch.randn(1, c, 16, 32, generator=g), size=map_hw, mode="bilinear", align_corners=True)[0]
true = F.OneShotFit(gs, uv, map_hw=map_hw)
true.load_maps(torch.zeros(48, *map_hw, device=dev), torch.zeros(1, *map_hw, device=dev))
with torch.no_grad():
    true.color_w.copy_((1 + 0.1 * torch.randn(48, generator=g)).to(dev))
    cb = torch.zeros(48, *map_hw); cb[:3] = lo(3, 0.15)
    tex = true.texels
    true.color_b_tex.copy_(tex.compact(cb.permute(1, 2, 0).contiguous().to(dev))[:, :3])
    true.opacity_b_tex.copy_(tex.compact(lo(1, 0.05).permute(1, 2, 0).contiguous().to(dev)))
    out = true.render(sc.w2c, sc.K, sc.H, sc.W, sc.bg, true.blend_values())
    gt_rgb, gt_mask = out["comp_rgb"].clone(), out["comp_mask"].mean(-1).clone()
f = F.OneShotFit(gs, uv, map_hw=map_hw)
args = (sc.w2c, sc.K, sc.H, sc.W, sc.bg, gt_rgb, gt_mask)
with torch.no_grad():
    o0 = f.render(*args[:5], f.blend_values())
    e0 = float((o0["comp_rgb"] - gt_rgb).abs().mean())
cap = f.captured(*args)                      # two regular steps
torch.cuda.synchronize(); t0 = time.perf_counter()
losses = []
steps_per_epoch = 25
for i in range(2, 300):
    if i % steps_per_epoch == 0:
        f.end_epoch()
    l = cap.replay()
    if i % 25 == 0 or i == 299:
        losses.append((i, float(l)))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
cap.check()
with torch.no_grad():
    o1 = f.render(*args[:5], f.blend_values())
    e1 = float((o1["comp_rgb"] - gt_rgb).abs().mean())
print(f"P = {sc.P}, 8 views {sc.H}x{sc.W}, maps 48x{map_hw[0]}x{map_hw[1]}, U = {f.texels.U} active texels")
print("loss:", "  ".join(f"{i}: {l:.5f}" for i, l in losses))
print(f"mean |rgb - target|: {e0:.5f} before, {e1:.5f} after 300 steps; 298 replayed steps in {dt * 1e3:.0f} ms ({dt / 298 * 1e3:.3f} ms per step incl. the host read-backs of this print loop)")
assert e1 < 0.35 * e0 and all(torch.isfinite(p).all() for p in (f.color_w, f.color_b_tex, f.opacity_b_tex))
