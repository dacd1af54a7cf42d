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
    assert lib.imd_abi_version() == 7


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


def test_pure_queries_work_without_gpu(lib):
    a, b = ctypes.c_int(), ctypes.c_int()
    assert lib.imd_attn_padded_dims(40, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (48, 64)
    assert lib.imd_attn_padded_dims(160, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (160, 160)
    assert lib.imd_attn_padded_dims(33, ctypes.byref(a), ctypes.byref(b)) != 0
    assert b"unsupported head dim" in lib.imd_last_error()
    assert lib.imd_groupnorm_workspace_floats(8, 4096, 320, 32) == 8 * 64 * 32 * 2 + 2 * 8 * 320       # 64-pixel chunk partials (512-block target) + coefficients
    assert lib.imd_groupnorm_workspace_floats(8, 64, 1280, 32) == 8 * 16 * 32 * 2 + 2 * 8 * 1280         # 4-pixel chunks at 8x8
    assert lib.imd_conv_gemm_auto_cfg(32768, 320) in (0, 1, 2, 4)


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
