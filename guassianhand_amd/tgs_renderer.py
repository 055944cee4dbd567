"""`renderer_cls: guassianhand_amd.tgs_renderer.GS3DRenderer` — the one line a maintainer changes in
config/config_one_shot.yaml:175 (the string is resolved by tgs.find, tgs/__init__.py:4-9: import_module + getattr) — and
`renderer_cls: guassianhand_amd.tgs_renderer.GS3DRendererEdit` for the three configs that bind the edit / avatar-drive renderer
(config_one_shot_edit.yaml:179, config_one_shot_avatar_drive.yaml:179, config_one_shot_edit_drive.yaml:180:
tgs.models.renderer_one_shot_edit.GS3DRenderer, whose forward_single_batch takes `render_edit` and per-Gaussian colour weights,
renderer_one_shot_edit.py:440-520).

The classes are built on first access from the reference's own classes (renderer.fused_renderer_cls / fused_renderer_cls_edit), so
importing this module needs nothing of the reference."""
_cache = {}


def __getattr__(name):
    if name == "GS3DRenderer":
        if name not in _cache:
            from tgs.models.renderer_one_shot import GS3DRenderer as base
            from .renderer import fused_renderer_cls
            _cache[name] = fused_renderer_cls(base)
        return _cache[name]
    if name == "GS3DRendererEdit":
        if name not in _cache:
            from tgs.models.renderer_one_shot_edit import GS3DRenderer as base
            from .renderer import fused_renderer_cls_edit
            _cache[name] = fused_renderer_cls_edit(base)
        return _cache[name]
    raise AttributeError(name)
