"""The C-ABI library loads (no GPU needed) and exports exactly what include/*.h declares."""
import ctypes
import os
import re

import helpers as H


def _declared():
    txt = open(os.path.join(H.ROOT, "include", "dflash_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dfl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from dflash_amd import _lib
    names = _declared()
    assert len(names) >= 16
    assert sorted(_lib.SIGNATURES) == names, "binding table and header disagree"
    handle = _lib.lib()
    for n in names:
        assert isinstance(getattr(handle, n), ctypes._CFuncPtr)
    assert handle.dfl_version() == 1
    assert handle.dfl_argmax_ws_bytes() > 0 and handle.dfl_attn_ws_bytes(32, 8) > 0


def test_binding_arity_matches_header():
    """Every ctypes signature has as many arguments as the header's prototype (a missing one
    shifts every later argument silently: found the hard way with dfl_accept_commit_batch)."""
    from dflash_amd import _lib
    txt = open(os.path.join(H.ROOT, "include", "dflash_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", txt, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} parameters, binding {len(args)}"


def test_batch_entry_points_validate_without_gpu():
    from dflash_amd import _lib
    h = _lib.lib()
    assert h.dfl_batch_tiles(1) == 2 and h.dfl_batch_tiles(2) == 2 and h.dfl_batch_tiles(3) == 4
    assert h.dfl_batch_ksplit(4096) == 2 and h.dfl_batch_ksplit(12288) == 6 and h.dfl_batch_ksplit(512) == 1
    assert h.dfl_gemm_batch_ws_bytes(4096, 4096) > 2 * 256 * 4 * 1024
    assert h.dfl_gemm_f32_batch(None, None, 2, 16, 32, None, None, None) == -22
    assert h.dfl_accept_commit_batch(None, 16, None, 16, 1, None, 1, 1, None, None, None, 0, None, None, 0, None) == -22


def test_argument_validation_needs_no_gpu():
    from dflash_amd import _lib
    h = _lib.lib()
    assert h.dfl_pack_weight(None, None, 16, 32, None) == -22
    assert b"null" in h.dfl_last_error()
    assert h.dfl_gemm_f32(1, None, None, 3, 16, 32, 1, 1, None, None) == -22
    assert h.dfl_accept_commit(1, 1, 0, 1, 4, 1, None, 0, None, None) == -22


def test_product_has_no_cpu_path():
    import pytest
    import torch
    from dflash_amd import sample
    with pytest.raises(RuntimeError):
        sample(torch.zeros(1, 2, 8), 0.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(H.ROOT, "dflash_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", ""), fn
