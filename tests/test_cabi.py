"""CPU: the C-ABI library loads and exports every symbol include/arfe_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "arfe_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arfe_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for must in ("arfe_fpn_gather_fwd", "arfe_fpn_gatemap_fwd", "arfe_fpn_scatter_fwd", "arfe_fpn_scatter_bwd",
                 "arfe_fpn_input_bwd", "arfe_fpn_gate_wgrad", "arfe_nonlocal_fwd", "arfe_nonlocal_bwd",
                 "arfe_rff_make_rois", "arfe_map_roi_levels", "arfe_roialign_fwd", "arfe_roialign_bwd", "arfe_roialign_bwd_prezero", "arfe_roialign_bwd_apply",
                 "arfe_conv3x3_fwd", "arfe_conv3x3_bwd_data", "arfe_conv3x3_bwd_weight", "arfe_rff_fuse_fwd",
                 "arfe_rff_fuse_bwd", "arfe_linear_fwd", "arfe_linear_bwd"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from arfe_amd.build import build_library
    lib_path = build_library()
    lib = ctypes.CDLL(lib_path)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/arfe_hip.h but not exported"


def test_binding_table_matches_header():
    from arfe_amd import hip
    assert sorted(hip.SIGNATURES) == declared_symbols()
    l = hip.lib()
    assert l.arfe_abi_version() == hip.ABI_VERSION


def test_argument_errors_do_not_touch_the_gpu():
    from arfe_amd import hip
    l = hip.lib()
    rc = l.arfe_fpn_gather_fwd(None, None, None, None, None)
    assert rc == -1 and b"pyramid" in l.arfe_last_error()
    p = hip.Pyramid()
    p.num_levels, p.refine_level, p.batch, p.channels = 9, 0, 1, 1
    assert l.arfe_fpn_gather_fwd(ctypes.byref(p), None, None, None, None) == -1
    assert b"num_levels" in l.arfe_last_error()
    assert l.arfe_map_roi_levels(4, None, 6, 56.0, None, None) == -1      # num_levels > 5
    assert l.arfe_linear_fwd(-1, 4, 4, None, None, None, 0, None, None, 0, None) == -1
    assert l.arfe_linear_workspace_bytes(1024, 12544, 1024) == (256 << 20) + 64 and l.arfe_linear_workspace_bytes(-1, 4, 4) == -1
    assert l.arfe_conv3x3_packed_bytes(256) == 5 * 9 * 256 * 256 * 4      # fwd + dgrad layouts: packed words, bf16 (hi | lo) planes, half planes
    assert l.arfe_set_operand_precision(2) == -1 and l.arfe_set_operand_precision(1) == 0 and l.arfe_get_operand_precision() == 1
    assert l.arfe_set_operand_precision(0) == 0 and l.arfe_get_operand_precision() == 0
    assert l.arfe_cast_multi(1, None, None, None, 0, None) == -1 and l.arfe_cast_multi(0, None, None, None, 0, None) == 0
    assert l.arfe_conv3x3_workspace_bytes(256) == 32 * 9 * 256 * 256 * 4   # 32 row-slice slabs of the stage-once weight gradient
    assert l.arfe_conv3x3_workspace_bytes(20) == 9 * 20 * 20 * 4
    assert l.arfe_nonlocal_workspace_bytes(2, 256, 4200) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from arfe_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hip.ArfeHipError, match="no CPU/PyTorch fallback"):
        hip.lib()
