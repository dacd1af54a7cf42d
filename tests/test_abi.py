"""The C-ABI library loads without a GPU and exports every symbol include/imagdressing_hip.h declares;
the ctypes structs mirror the header field-for-field."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "imagdressing_hip.h")


@pytest.fixture(scope="module")
def lib():
    from imagdressing_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(imd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"


def test_binding_covers_header(lib):
    from imagdressing_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared_functions()
    assert lib.imd_abi_version() == _lib.ABI_VERSION == 9


def header_struct_fields(name):
    text = open(HEADER).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.split(",")
        for i, part in enumerate(parts):
            nm = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?\s*$", part.strip())[0]
            fields.append(nm)
    return fields


@pytest.mark.parametrize("cname,pyname", [("imd_heads_dest", "HeadsDest"), ("imd_conv_gemm_params", "ConvGemmParams"),
                                          ("imd_attn_params", "AttnParams"), ("imd_groupnorm_params", "GroupNormParams"),
                                          ("imd_layernorm_params", "LayerNormParams"), ("imd_ddim_params", "DdimParams"),
                                          ("imd_ff_params", "FfParams")])
def test_struct_layout_matches_header(cname, pyname):
    from imagdressing_amd import _lib
    py = [f[0] for f in getattr(_lib, pyname)._fields_]
    assert py == header_struct_fields(cname)


def test_integration_md_ctypes_snippet_matches_header():
    """INTEGRATION.md section 2 shows a maintainer the ctypes mirror of imd_attn_params: the fenced snippet is parsed and its
    _fields_ must be the header's, name for name, in order (a stale snippet hands the library a truncated struct)."""
    import ast
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    snippet = [b for b in blocks if "class AttnParams" in b]
    assert len(snippet) == 1
    tree = ast.parse(snippet[0])
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AttnParams"][0]
    assign = [n for n in cls.body if isinstance(n, ast.Assign) and n.targets[0].id == "_fields_"][0]
    names = [elt.elts[0].value for elt in assign.value.elts]
    assert names == header_struct_fields("imd_attn_params")
    types = [ast.unparse(elt.elts[1]) for elt in assign.value.elts]
    from imagdressing_amd import _lib
    mine = [f[1] for f in _lib.AttnParams._fields_]
    assert [getattr(ctypes, ty.split(".")[-1]) for ty in types] == mine          # (c_uint32 is an alias of c_uint)
    assert "struct_bytes = C.sizeof(p)" in snippet[0] and "imd_abi_version() == %d" % _lib.ABI_VERSION in snippet[0]


def test_foreign_struct_size_is_refused(lib):
    """ABI v8: a parameter block whose struct_bytes is not the library's sizeof is refused before any field is read (no GPU needed:
    the check precedes the launch)."""
    from imagdressing_amd import _lib
    for cls, fn, extra in ((_lib.AttnParams, lib.imd_attention, (None,)), (_lib.ConvGemmParams, lib.imd_conv_gemm, (0, None)),
                           (_lib.GroupNormParams, lib.imd_groupnorm, (None,)), (_lib.LayerNormParams, lib.imd_layernorm, (None,)),
                           (_lib.DdimParams, lib.imd_ddim_cfg_step, (None,)), (_lib.FfParams, lib.imd_ff_geglu, (None,))):
        p = cls()
        assert p.struct_bytes == ctypes.sizeof(cls)
        p.struct_bytes = ctypes.sizeof(cls) - 8                 # what a binding of an older, shorter header would pass
        assert fn(ctypes.byref(p), *extra) != 0
        assert b"parameter block is" in lib.imd_last_error() and b"ABI v9" in lib.imd_last_error()
        p.struct_bytes = 0                                      # a v7 caller: first word is the low half of a pointer or zero
        assert fn(ctypes.byref(p), *extra) != 0 and b"parameter block is" in lib.imd_last_error()
    q = _lib.ConvGemmParams()
    q.struct_bytes = 4
    assert lib.imd_conv_patch_supported(ctypes.byref(q)) == 0 and lib.imd_gemm_dma_supported(ctypes.byref(q)) == 0


def test_pure_queries_work_without_gpu(lib):
    a, b = ctypes.c_int(), ctypes.c_int()
    assert lib.imd_attn_padded_dims(40, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (48, 64)
    assert lib.imd_attn_padded_dims(160, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (160, 160)
    assert lib.imd_attn_padded_dims(33, ctypes.byref(a), ctypes.byref(b)) != 0
    assert b"unsupported head dim" in lib.imd_last_error()
    assert lib.imd_groupnorm_workspace_floats(8, 4096, 320, 32) == 8 * 64 * 32 * 2 + 2 * 8 * 320       # 64-pixel chunk partials (512-block target) + coefficients
    assert lib.imd_groupnorm_workspace_floats(8, 64, 1280, 32) == 8 * 16 * 32 * 2 + 2 * 8 * 1280         # 4-pixel chunks at 8x8
    assert lib.imd_conv_gemm_auto_cfg(32768, 320) in (0, 1, 2, 4)
    # tile config 24 (whole 8-wide maps, K-sliced only): a pure predicate over the parameter block
    from imagdressing_amd import _lib
    q = _lib.ConvGemmParams()
    q.M, q.N, q.K, q.Cin, q.taps, q.stride = 8 * 64, 1280, 9 * 1280, 1280, 9, 1
    q.Hin = q.Win = q.Hout = q.Wout = 8
    q.x_pix_stride, q.split_k = 1280, 6
    assert lib.imd_conv_img_supported(ctypes.byref(q)) == 1
    for field, bad in (("split_k", 1), ("Wout", 16), ("Cin", 1296), ("N", 1248), ("stride", 2), ("Hout", 13), ("out_f32", 1)):
        keep = getattr(q, field)
        setattr(q, field, bad)
        assert lib.imd_conv_img_supported(ctypes.byref(q)) == 0, field
        setattr(q, field, keep)
    assert lib.imd_conv_img_supported(ctypes.byref(q)) == 1


def test_no_cpu_path():
    """CPU tensors are refused loudly: there is no fallback to hide a missing GPU / library."""
    import torch
    from imagdressing_amd import ops
    from imagdressing_amd._lib import ImdError
    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(ImdError):
        ops.linear(x, x)
    with pytest.raises(ImdError):
        ops.layer_norm(x, torch.ones(8), torch.zeros(8))
    from imagdressing_amd.adapter.attention_processor import AttnProcessor2_0

    class A:
        heads = 8
    with pytest.raises(ImdError):
        AttnProcessor2_0()(type("attn", (), {"to_q": type("w", (), {"weight": x})()})(), torch.zeros(1, 4, 64))
