"""C-ABI surface: the library loads and exports every symbol include/gh_raster.h declares; layout and
argument validation behave as the header says. No GPU compute is launched here."""
import ctypes as C
import os
import re

import pytest

from guassianhand_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gh_raster.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|size_t)\s+(gh_\w+)\s*\(", txt, flags=re.M)))


def test_header_declares_expected_symbols():
    assert header_symbols() == sorted(_abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(gh_lib_path):
    L = C.CDLL(gh_lib_path)
    for sym in header_symbols():
        assert hasattr(L, sym), sym
    _abi.declare(L)
    assert L.gh_version() == (0 << 16) | 8


def test_struct_sizes_match_header():
    # GhDims: abi tag (v0.8) + 6 int32 + float + uint32 (+ 4 bytes of padding) + int64
    assert C.sizeof(_abi.GhDims) == 48 and _abi.GhDims.abi.offset == 0 and _abi.GhDims.max_instances.offset == 40
    assert C.sizeof(_abi.GhInputs) == 13 * 8          # v0.6: + cov3D_precomp
    assert C.sizeof(_abi.GhGrads) == 16 * 8           # v0.6: + dL_dcov3D; v0.8: + deferred_loss
    assert C.sizeof(_abi.GhOutputs) == 9 * 8          # 4 pointers + (float, uint32) + v0.7: the fused L1's 3 pointers + fit_loss
    assert C.sizeof(_abi.GhFitLoss) == 8 * 8          # 3 pointers, 3 floats (+ padding), 3 pointers
    assert C.sizeof(_abi.GhCounters) == 16
    assert C.sizeof(_abi.GhLayout) == len(_abi.LAYOUT_FIELDS) * 8


def test_workspace_layout(gh_lib_path):
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    d = _abi.GhDims(98562, 8, 512, 334, 0, 0, 1.0, 0, 5_000_000)
    lay = _abi.GhLayout()
    assert L.gh_workspace_layout(C.byref(d), C.byref(lay)) == 0
    offs = [getattr(lay, f) for f in _abi.LAYOUT_FIELDS[1:]]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert lay.total_bytes == L.gh_workspace_bytes(C.byref(d))
    assert lay.keys_b - lay.keys_a >= 4 * 5_000_000 and lay.clamped - lay.geom >= 64 * 8 * 98562 and lay.depth_keys_b - lay.depth_keys_a >= 4 * 8 * 98562
    assert lay.inst_grad + 4 * 36 * 5_000_000 <= lay.inst_flag      # 4 quadrant sub-records of 9 floats per instance
    # monotone in capacity
    d2 = _abi.GhDims(98562, 8, 512, 334, 0, 0, 1.0, 0, 6_000_000)
    assert L.gh_workspace_bytes(C.byref(d2)) > lay.total_bytes


@pytest.mark.parametrize("dims,code", [
    ((10, 0, 64, 64, 0, 0, 1.0, 0, 100), _abi.GH_ERR_INVALID_ARG),      # n_views < 1
    ((10, 1, 64, 64, 4, 16, 1.0, 0, 100), _abi.GH_ERR_UNSUPPORTED),     # sh_degree > 3
    ((10, 1, 64, 64, 0, 5, 1.0, 0, 100), _abi.GH_ERR_UNSUPPORTED),      # M not in {0,1,4,9,16}
    ((10, 1, 64, 16 * 256, 0, 0, 1.0, 0, 100), _abi.GH_ERR_UNSUPPORTED),  # > 255 tiles wide
    ((10, 1, 64, 64, 0, 0, 1.0, 0, -1), _abi.GH_ERR_INVALID_ARG),
])
def test_invalid_dims_rejected(gh_lib_path, dims, code):
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    d = _abi.GhDims(*dims)
    lay = _abi.GhLayout()
    assert L.gh_workspace_layout(C.byref(d), C.byref(lay)) == code
    assert L.gh_workspace_bytes(C.byref(d)) == 0


def test_forward_rejects_bad_arguments_before_touching_the_gpu(gh_lib_path):
    """Validation errors are returned as status codes (nothing throws across the boundary, no launch)."""
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    d = _abi.GhDims(10, 1, 64, 64, 0, 0, 1.0, 0, 100)
    inp = _abi.GhInputs()        # all NULL
    out = _abi.GhOutputs()
    assert L.gh_forward(C.byref(d), C.byref(inp), C.byref(out), None, 0, None) == _abi.GH_ERR_INVALID_ARG
    assert L.gh_forward(None, C.byref(inp), C.byref(out), None, 0, None) == _abi.GH_ERR_INVALID_ARG
    gr = _abi.GhGrads()
    assert L.gh_backward(C.byref(d), C.byref(inp), C.byref(gr), None, 0, None) == _abi.GH_ERR_INVALID_ARG
    # both / neither colour source
    one = C.c_void_p(8)
    inp2 = _abi.GhInputs(one, one, one, one, one, one, one, None, None, None, None)
    assert L.gh_forward(C.byref(d), C.byref(inp2), C.byref(_abi.GhOutputs(one, one, None)), one, 1 << 30, None) == _abi.GH_ERR_INVALID_ARG
    # workspace too small is reported, with valid-looking pointers, before any launch
    inp3 = _abi.GhInputs(one, one, one, one, one, None, one, None, None, None, None)
    assert L.gh_forward(C.byref(d), C.byref(inp3), C.byref(_abi.GhOutputs(one, one, None)), one, 16, None) == _abi.GH_ERR_WORKSPACE_SMALL


def test_occlusion_bound_arguments_are_validated_before_any_launch(gh_lib_path):
    """GhInputs.tile_depth_bound / GhOutputs.tile_depth_seen (v0.5): the report must not alias the bound, and lists that outlive
    the call (GH_FLAG_STATIC_LISTS) cannot be truncated by a per-call bound."""
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    one, two = C.c_void_p(256), C.c_void_p(512)
    d = _abi.GhDims(10, 1, 64, 64, 0, 0, 1.0, 0, 100)
    inp = _abi.GhInputs(one, one, one, one, one, None, one, None, None, None, None, two)
    out = _abi.GhOutputs(one, one, None, two, 1.002, 8)
    assert L.gh_forward(C.byref(d), C.byref(inp), C.byref(out), one, 1 << 30, None) == _abi.GH_ERR_INVALID_ARG
    ds = _abi.GhDims(10, 1, 64, 64, 0, 0, 1.0, _abi.GH_FLAG_STATIC_LISTS, 100)
    out2 = _abi.GhOutputs(one, one, None, None, 1.0, 0)
    assert L.gh_forward(C.byref(ds), C.byref(inp), C.byref(out2), one, 1 << 30, None) == _abi.GH_ERR_UNSUPPORTED
    assert _abi.GH_FLAG_DEPTH24 == 32


def test_fused_loss_arguments_are_validated_before_any_launch(gh_lib_path):
    """GhOutputs.l1_* (v0.7): all three pointers or none; not with the mask channel, an occlusion report / bound or two streams;
    gh_forward only."""
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    one, two = C.c_void_p(1 << 20), C.c_void_p(2 << 20)          # (fake device addresses, a megabyte apart: disjoint image-sized ranges)
    d = _abi.GhDims(10, 2, 32, 32, 0, 0, 1.0, 0, 1000)
    inp = _abi.GhInputs(one, one, one, one, one, None, one)
    call = lambda out, dims=d, i=inp: L.gh_forward(C.byref(dims), C.byref(i), C.byref(out), one, 16, None)
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, one, None, one)) == _abi.GH_ERR_INVALID_ARG
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, one, one, None)) == _abi.GH_ERR_INVALID_ARG
    three = C.c_void_p(3 << 20)
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, two, three, one)) == _abi.GH_ERR_WORKSPACE_SMALL   # three distinct arrays
    for img, tgt, dl in ((one, one, two), (one, two, one), (one, two, two)):                                     # any two the same
        assert call(_abi.GhOutputs(img, one, None, None, 1.0, 0, tgt, dl, three)) == _abi.GH_ERR_INVALID_ARG
    # partially overlapping ranges are as wrong as equal pointers (the epilogue stores the gradient while other waves read the target)
    nb = 2 * 3 * 32 * 32 * 4
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, two, C.c_void_p((2 << 20) + nb - 4), three)) == _abi.GH_ERR_INVALID_ARG
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, two, C.c_void_p((2 << 20) + nb), three)) == _abi.GH_ERR_WORKSPACE_SMALL
    assert call(_abi.GhOutputs(one, one, one, None, 1.0, 0, two, three, one)) == _abi.GH_ERR_UNSUPPORTED
    assert call(_abi.GhOutputs(one, one, None, two, 1.0, 0, two, three, one)) == _abi.GH_ERR_UNSUPPORTED
    inp_b = _abi.GhInputs(one, one, one, one, one, None, one, None, None, None, None, two)
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, two, three, one), i=inp_b) == _abi.GH_ERR_UNSUPPORTED
    d2 = _abi.GhDims(10, 2, 32, 32, 0, 0, 1.0, _abi.GH_FLAG_SPLIT_STREAMS, 1000)
    assert call(_abi.GhOutputs(one, one, None, None, 1.0, 0, two, three, one), d2) == _abi.GH_ERR_UNSUPPORTED
    out = _abi.GhOutputs(one, None, None, None, 1.0, 0, one, one, one)
    assert L.gh_forward_shared(C.byref(d), C.byref(inp), C.byref(out), two, one, 16, None) == _abi.GH_ERR_UNSUPPORTED
    d3 = _abi.GhDims(10, 2, 32, 32, 0, 0, 1.0, _abi.GH_FLAG_STATIC_LISTS, 1000)
    out3 = _abi.GhOutputs(one, None, None, None, 1.0, 0, two, three, one)
    assert L.gh_forward_refresh(C.byref(d3), C.byref(inp), C.byref(out3), two, one, 16, None) == _abi.GH_ERR_WORKSPACE_SMALL   # a refresh fuses it too
    # the fit's image loss (GhOutputs.fit_loss): needs the mask channel; every pointer but bbox; not together with l1_target
    four, five = C.c_void_p(4 << 20), C.c_void_p(5 << 20)
    fit = _abi.GhFitLoss(two, three, None, 10.0, 1.0, 1.0, four, five, one)
    mk = lambda alpha, f, l1=(None, None, None): _abi.GhOutputs(one, one, alpha, None, 1.0, 0, *l1, C.pointer(f))
    assert call(mk(C.c_void_p(6 << 20), fit)) == _abi.GH_ERR_WORKSPACE_SMALL
    assert L.gh_forward_refresh(C.byref(d3), C.byref(inp), C.byref(mk(C.c_void_p(6 << 20), fit)), two, one, 16, None) == _abi.GH_ERR_WORKSPACE_SMALL
    assert call(mk(None, fit)) == _abi.GH_ERR_INVALID_ARG                                       # no mask channel
    assert call(mk(C.c_void_p(6 << 20), fit, (two, three, one))) == _abi.GH_ERR_INVALID_ARG          # both losses
    assert call(mk(C.c_void_p(6 << 20), _abi.GhFitLoss(two, None, None, 10.0, 1.0, 1.0, four, five, one))) == _abi.GH_ERR_INVALID_ARG
    assert call(mk(C.c_void_p(6 << 20), _abi.GhFitLoss(two, three, None, 10.0, 1.0, 1.0, one, five, one))) == _abi.GH_ERR_INVALID_ARG   # dL aliases the image
    assert call(mk(C.c_void_p(6 << 20), fit), d2) == _abi.GH_ERR_UNSUPPORTED                        # two streams
    assert L.gh_forward_shared(C.byref(d), C.byref(inp), C.byref(mk(C.c_void_p(6 << 20), fit)), two, one, 16, None) == _abi.GH_ERR_UNSUPPORTED


def test_abi_tag_is_checked_before_anything_else(gh_lib_path):
    """v0.8: GhDims starts with the tag of the header the caller was compiled against. A host built against an older header
    (whose first field is P) or against another minor version gets GH_ERR_ABI from every entry point that takes a GhDims — before
    the library reads a struct that may be shorter than its own."""
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    h = open(os.path.join(ROOT, "include", "gh_raster.h")).read()
    major, minor = int(re.search(r"#define GH_VERSION_MAJOR (\d+)", h).group(1)), int(re.search(r"#define GH_VERSION_MINOR (\d+)", h).group(1))
    assert (major, minor) == (_abi.GH_VERSION_MAJOR, _abi.GH_VERSION_MINOR) and _abi.GH_ABI_TAG == 0x47480000 | (major << 8) | minor
    good = _abi.GhDims(10, 1, 64, 64, 0, 0, 1.0, 0, 100)
    assert good.abi == _abi.GH_ABI_TAG and L.gh_workspace_bytes(C.byref(good)) > 0
    one = C.c_void_p(1 << 20)
    inp, out, gr = _abi.GhInputs(one, one, one, one, one, None, one), _abi.GhOutputs(one, one, None), _abi.GhGrads(one)
    for tag in (10, _abi.GH_ABI_TAG - 1, _abi.GH_ABI_TAG + 1, 0):          # an old host's P in that place; the neighbouring minors; nothing
        bad = _abi.GhDims(10, 1, 64, 64, 0, 0, 1.0, 0, 100)
        bad.abi = tag
        lay = _abi.GhLayout()
        assert L.gh_workspace_layout(C.byref(bad), C.byref(lay)) == _abi.GH_ERR_ABI and L.gh_workspace_bytes(C.byref(bad)) == 0
        assert L.gh_forward(C.byref(bad), C.byref(inp), C.byref(out), one, 1 << 30, None) == _abi.GH_ERR_ABI
        assert L.gh_backward(C.byref(bad), C.byref(inp), C.byref(gr), one, 1 << 30, None) == _abi.GH_ERR_ABI
        assert L.gh_forward_stages(C.byref(bad), C.byref(inp), C.byref(out), one, 1 << 30, None, 7) == _abi.GH_ERR_ABI
        assert L.gh_backward_stages(C.byref(bad), C.byref(inp), C.byref(gr), one, 1 << 30, None, 3) == _abi.GH_ERR_ABI
        for fn, o in ((L.gh_forward_shared, out), (L.gh_forward_refresh, out), (L.gh_backward_shared, gr), (L.gh_backward_refresh, gr)):
            assert fn(C.byref(bad), C.byref(inp), C.byref(o), one, C.c_void_p(2 << 20), 1 << 30, None) == _abi.GH_ERR_ABI
    assert _abi.status_name(_abi.GH_ERR_ABI) == "GH_ERR_ABI"


def test_loader_refuses_a_library_of_another_version(gh_lib_path, monkeypatch):
    """_lib.lib() compares gh_version() with the version this Python mirror was written against (GH_RASTER_LIB may point at a build of
    another commit): a mismatch is an error at load time, not a struct read with the wrong layout later."""
    from guassianhand_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_abi, "GH_VERSION_MINOR", _abi.GH_VERSION_MINOR + 1)
    with pytest.raises(_lib.GhLibraryError, match="C-ABI v0"):
        _lib.lib()
    monkeypatch.undo()
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.lib() is not None


def test_deferred_loss_sum_arguments(gh_lib_path):
    """GhGrads.deferred_loss (v0.8): only the entry points whose forward can fuse a loss take it."""
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    one, two, three = C.c_void_p(1 << 20), C.c_void_p(2 << 20), C.c_void_p(3 << 20)
    inp = _abi.GhInputs(one, one, one, one, one, None, one)
    gr = _abi.GhGrads(one)
    gr.deferred_loss = three
    d = _abi.GhDims(10, 2, 32, 32, 0, 0, 1.0, _abi.GH_FLAG_DEFER_LOSS_SUM, 1000)
    assert L.gh_backward(C.byref(d), C.byref(inp), C.byref(gr), one, 16, None) == _abi.GH_ERR_WORKSPACE_SMALL          # accepted
    ds = _abi.GhDims(10, 2, 32, 32, 0, 0, 1.0, _abi.GH_FLAG_SPLIT_STREAMS, 1000)
    assert L.gh_backward(C.byref(ds), C.byref(inp), C.byref(gr), one, 16, None) == _abi.GH_ERR_UNSUPPORTED
    assert L.gh_backward_shared(C.byref(d), C.byref(inp), C.byref(gr), two, one, 16, None) == _abi.GH_ERR_UNSUPPORTED
    dr = _abi.GhDims(10, 2, 32, 32, 0, 0, 1.0, _abi.GH_FLAG_STATIC_LISTS | _abi.GH_FLAG_DEFER_LOSS_SUM, 1000)
    assert L.gh_backward_refresh(C.byref(dr), C.byref(inp), C.byref(gr), two, one, 16, None) == _abi.GH_ERR_WORKSPACE_SMALL   # accepted
    assert _abi.GH_FLAG_DEFER_LOSS_SUM == 64


def test_product_has_no_cpu_fallback():
    """CPU tensors must raise (a silent CPU path would void the parity claims)."""
    import torch
    from guassianhand_amd.rasterizer import raster_forward
    z = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        raster_forward(torch.zeros(1, 40), z, torch.zeros(4), z, torch.zeros(4, 4), H=16, W=16, colors_precomp=z)


def test_product_never_imports_oracle():
    """No file of the product package may reference the oracle (test infrastructure only)."""
    pkg = os.path.join(ROOT, "guassianhand_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gh_oracle" not in txt, f


def test_python_mirror_follows_the_header_field_by_field():
    """GhLayout's field order and the GH_FLAG_* values of the ctypes mirror are read off include/gh_raster.h."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = open(os.path.join(root, "include", "gh_raster.h")).read()
    body = h[h.index("typedef struct GhLayout"):h.index("} GhLayout;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"size_t\s+([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert tuple(fields) == tuple(_abi.LAYOUT_FIELDS)
    flags = dict(re.findall(r"#define (GH_FLAG_[A-Z_]+) (\d+)u", h))
    for name, val in flags.items():
        if name != "GH_FLAG_NONE":
            assert getattr(_abi, name) == int(val), name


def test_workspace_layout_properties_over_random_dims(gh_lib_path):
    """gh_workspace_layout over 2,000 random dimension sets (all flag combinations, degenerate sizes): accepted dims give a layout whose
    offsets ascend in struct order in 256-byte steps and end inside total_bytes; the size grows with the instance capacity; rejected
    dims give 0 bytes; nothing crashes (pure host arithmetic, no GPU)."""
    import random
    L = C.CDLL(gh_lib_path)
    _abi.declare(L)
    rnd = random.Random(7)
    flag_bits = (_abi.GH_FLAG_BLEND_W_PER_GAUSSIAN, _abi.GH_FLAG_BLEND_COLOR_B_RGB, _abi.GH_FLAG_PER_VIEW_GAUSSIANS, _abi.GH_FLAG_SPLIT_STREAMS,
                 _abi.GH_FLAG_STATIC_LISTS, _abi.GH_FLAG_DEPTH24, _abi.GH_FLAG_DEFER_LOSS_SUM, _abi.GH_FLAG_FRESH_ORDER)
    ok = 0
    for _ in range(2000):
        P = rnd.choice([0, 1, 2, 255, 256, 257, 1000, 98562, rnd.randint(0, 300000)])
        NV = rnd.choice([1, 2, 3, 8, 32, rnd.randint(1, 64)])
        H, W = (rnd.choice([1, 15, 16, 17, 334, 512, 1024, rnd.randint(1, 4080)]) for _ in range(2))
        M = rnd.choice([0, 0, 1, 4, 9, 16])
        flags = 0
        for b in flag_bits:
            flags |= b if rnd.random() < 0.3 else 0
        cap = rnd.choice([0, 1, 63, 64, 65, 2047, 2048, 2049, 10 ** 6, rnd.randint(0, 2 * 10 ** 7)])
        d = _abi.GhDims(P, NV, H, W, rnd.randint(0, 3), M, 1.0, flags, cap)
        lay = _abi.GhLayout()
        rc = L.gh_workspace_layout(C.byref(d), C.byref(lay))
        nbytes = L.gh_workspace_bytes(C.byref(d))
        if rc != 0:
            assert rc in (_abi.GH_ERR_INVALID_ARG, _abi.GH_ERR_UNSUPPORTED) and nbytes == 0
            continue
        ok += 1
        offs = [getattr(lay, f) for f in _abi.LAYOUT_FIELDS[1:]]
        assert offs == sorted(offs), (P, NV, H, W, M, flags, cap)
        assert all(o % 256 == 0 for o in offs) and offs[-1] <= lay.total_bytes == nbytes
        assert lay.keys_b - lay.keys_a >= 4 * cap and lay.vals_b - lay.vals_a >= 4 * cap
        d2 = _abi.GhDims(P, NV, H, W, d.sh_degree, M, 1.0, flags, cap + 4096)
        assert L.gh_workspace_bytes(C.byref(d2)) >= nbytes
    assert ok > 900, ok
