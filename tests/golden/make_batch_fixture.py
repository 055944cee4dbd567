#!/usr/bin/env python
"""Generate tests/golden/batch_fixture.npz by running the REFERENCE's own GS3DRenderer.forward_single_batch
(tgs/models/renderer_one_shot.py:448-512) on CPU, under the stub finder of make_host_fixtures.py.

What is pinned is the COMPOSITION: prune (> threshold_low) and duplicate-and-refine (> threshold_high) by boolean-mask indexing
(:468-474), the cat order of points and features (:476-477), forward_gs on the concatenated set (:478), the UV lookup of
color_b / opacity_b at the UVs of the concatenated set (:481-492), the per-view loop with its two rasteriser calls (:494-503) and
the stacked dict (:505-510). The sub-modules the method calls (gs_valid, vert_pos_refinement, forward_gs, get_uvd) are networks /
a third-party function outside the scope: tests/helpers.py::BatchStandIns supplies deterministic stand-ins, the same objects the
GPU test hands to the composed path. query_triplane_texture and forward_single_view are the reference's own methods, bound to
the stand-in object. The recording fake rasteriser captures every call; the .npz holds inputs and captured data only.

Runs only in the build container (needs /root/reference). Usage: python tests/golden/make_batch_fixture.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
OUT = os.path.join(HERE, "batch_fixture.npz")


def main():
    import make_host_fixtures as mh
    calls = []
    mh.install_stubs(calls)
    sys.path.insert(0, mh.REF)
    import tgs.models.renderer_one_shot as ref
    from helpers import BatchStandIns, batch_inputs

    out = {}
    for tag, use_rgb in (("rgb", True), ("sh", False)):
        st = BatchStandIns("cpu", use_rgb=use_rgb)
        inp = batch_inputs()
        ns = st.namespace("cpu")
        # the reference's own methods on the stand-in object; get_uvd is a module global of the reference (:19)
        ns.query_triplane_texture = types.MethodType(ref.GS3DRenderer.query_triplane_texture, ns)
        ns.forward_single_view = types.MethodType(ref.GS3DRenderer.forward_single_view, ns)
        ref.get_uvd = st.get_uvd
        calls.clear()
        res = ref.GS3DRenderer.forward_single_batch(
            ns, inp["feat"], inp["pts"], inp["w2cs"], inp["Ks"], inp["H"], inp["W"], 0.71, 1.42, inp["bg"],
            color_w=inp["color_w"], xyz_b=inp["xyz_b"], color_b=inp["color_b"], opacity_b=inp["opacity_b"],
            vert3d_uv=[None], face_uv=None, face_uv_xy=None)
        out[f"{tag}_keys"] = np.array(sorted(res.keys()))
        for k, v in res.items():
            if isinstance(v, torch.Tensor):
                out[f"{tag}_shape_{k}"] = np.array(v.shape)
        out[f"{tag}_comp_rgb_bg"] = res["comp_rgb_bg"].numpy()
        for k in ("xyz", "opacity", "rotation", "scaling", "shs"):
            out[f"{tag}_3dgs_{k}"] = getattr(res["3dgs"], k).detach().numpy()
        out[f"{tag}_ncalls"] = np.array(len(calls))
        for ci, (rs, kw) in enumerate(calls):
            out[f"{tag}_call{ci}_cam"] = np.concatenate([rs.viewmatrix.reshape(-1).numpy(), rs.projmatrix.reshape(-1).numpy(),
                                                         rs.campos.numpy(), [rs.tanfovx, rs.tanfovy], rs.bg.numpy()]).astype(np.float32)
            out[f"{tag}_call{ci}_sh_degree"] = np.array(rs.sh_degree)
            out[f"{tag}_call{ci}_hw"] = np.array([rs.image_height, rs.image_width])
            for k, v in kw.items():
                if v is not None and k != "means2D":
                    out[f"{tag}_call{ci}_{k}"] = v.detach().numpy()
        # selection counts, for the record
        s = inp["feat"][:, 0]
        out[f"{tag}_counts"] = np.array([int((s > 0.1).sum()), int((s > 0.9).sum())])
    # the plugin seam: the subclass factory over the REAL reference class
    from guassianhand_amd.renderer import forward_single_batch, fused_renderer_cls
    cls = fused_renderer_cls(ref.GS3DRenderer)
    ok = issubclass(cls, ref.GS3DRenderer) and cls.forward_single_batch is forward_single_batch and cls.forward is ref.GS3DRenderer.forward
    import inspect
    same_sig = list(inspect.signature(forward_single_batch).parameters) == list(inspect.signature(ref.GS3DRenderer.forward_single_batch).parameters)
    out["seam_ok"] = np.array([int(ok), int(same_sig)])
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes; seam {out['seam_ok']}; counts {out['rgb_counts']}")


if __name__ == "__main__":
    main()
