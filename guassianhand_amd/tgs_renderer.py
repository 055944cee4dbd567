"""`renderer_cls: guassianhand_amd.tgs_renderer.GS3DRenderer` — the one line a maintainer changes in
config/config_one_shot.yaml:175 (the string is resolved by tgs.find, tgs/__init__.py:4-9: import_module + getattr).

The class is built on first access from the reference's own GS3DRenderer (renderer.fused_renderer_cls), so importing this
module needs nothing of the reference. (renderer_one_shot_edit.py's forward_single_batch has another signature — `render_edit`,
colour weights looked up from a map it builds per call, :488-500 — and keeps running on the import shim unchanged.)"""
_cache = {}


def __getattr__(name):
    if name == "GS3DRenderer":
        if name not in _cache:
            from tgs.models.renderer_one_shot import GS3DRenderer as base
            from .renderer import fused_renderer_cls
            _cache[name] = fused_renderer_cls(base)
        return _cache[name]
    raise AttributeError(name)
