"""The C-ABI library loads without a GPU and exports exactly what include/demf_hip.h
declares; the ctypes table in demf_amd/_ffi.py agrees with the header."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from demf_amd import _ffi


def _header_decls():
    text = open(os.path.join(ROOT, "include", "demf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(demf_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        decls[m.group(2)] = n
    return decls


def test_library_exports_every_declared_symbol():
    decls = _header_decls()
    assert len(decls) >= 20
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in demf_hip.h but not exported"


def test_ffi_table_matches_header():
    decls = _header_decls()
    lib = _ffi.load()
    assert lib.demf_version() == 2
    assert lib.demf_last_error() is not None
    for name, argtypes in _ffi.SIGNATURES.items():
        assert name in decls, f"{name} bound in _ffi.py but not declared in the header"
        assert len(argtypes) == decls[name], f"{name}: arity differs from the header"
    bound = set(_ffi.SIGNATURES) | {"demf_version", "demf_last_error"}
    assert bound == set(decls)


PUBLIC = {"demf_version", "demf_last_error", "demf_fps_f32", "demf_ball_query_f32", "demf_group_points_fwd",
          "demf_group_points_bwd", "demf_gather_points_fwd", "demf_gather_points_bwd", "demf_three_nn_f32",
          "demf_three_interpolate_fwd", "demf_three_interpolate_bwd", "demf_msda_fwd_f32", "demf_msda_bwd_f32"}


def test_header_marks_everything_but_the_upstream_shaped_operators_internal():
    """include/demf_hip.h: the entry points with an upstream counterpart (mmdet3d.ops / mmcv.ops) are the public ABI;
    every other declaration carries DEMF_INTERNAL."""
    text = open(os.path.join(ROOT, "include", "demf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    marked = set(re.findall(r"DEMF_INTERNAL[ \t]+int[ \t]+(demf_\w+)\s*\(", text))
    every = set(_header_decls())
    assert PUBLIC <= every
    assert marked == every - PUBLIC, sorted((every - PUBLIC) ^ marked)


def test_bad_arguments_are_reported_not_launched():
    # argument validation happens before any device work, so this runs without a GPU
    with pytest.raises(RuntimeError, match="bad sizes"):
        _ffi.call("demf_fps_f32", 1, 0, 4, None, None, None, None)
    with pytest.raises(RuntimeError, match="null pointer"):
        _ffi.call("demf_ball_query_f32", 1, 8, 2, 0.0, 1.0, 4, None, None, None, None)
    assert b"null pointer" in _ffi.load().demf_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from demf_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.furthest_point_sample(torch.zeros(1, 8, 3), 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.ball_query(0.0, 1.0, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))
