"""The built librda_hip.so loads without a GPU and exports every symbol include/rda_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rda_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rda_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported():
    from rda_planner_amd import _lib
    so = _lib.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 18, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rda_hip.h but not exported"


def test_struct_layouts_match_header():
    from rda_planner_amd._capi import Cfg, Info
    assert ctypes.sizeof(Cfg) == 8 * 4 + 16 * 8
    assert ctypes.sizeof(Info) == 2 * 8 + 3 * 4 + 4          # padded to 8


def test_product_path_has_no_oracle_dependency():
    """rda_planner_amd must never import / link the oracle"""
    pkg = os.path.join(ROOT, "rda_planner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "librda_oracle" not in text, f


def test_no_device_is_a_loud_error():
    """without a GPU the product refuses to construct a solver instead of falling back"""
    import pytest
    from rda_planner_amd import _lib
    lib = ctypes.CDLL(_lib.build())
    lib.rda_device_count.restype = ctypes.c_int
    if lib.rda_device_count() > 0:
        pytest.skip("a GPU is present")
    from rda_planner_amd import scenarios as sc
    from rda_planner_amd.rda_solver import RDA_solver
    with pytest.raises(RuntimeError):
        RDA_solver(5, sc.rectangle_robot(), time_print=False)


def test_host_side_helpers_build_and_load():
    """the C caller bench.py times (tools/closed_loop_host.c) and the flatten accelerator of the Python API (csrc/flatten_ext.c) are plain gcc
    builds: they compile, load and expose what their ctypes / import users expect (no GPU needed)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop_host as clh
    lib = ctypes.CDLL(clh.build())
    assert hasattr(lib, "closed_loop_run")
    assert ctypes.sizeof(clh.Api) == 5 * 8 and ctypes.sizeof(clh.Scene) == 4 * 4 + 5 * 8        # struct closed_loop_api / closed_loop_scene
    src = open(os.path.join(ROOT, "tools", "closed_loop_host.c")).read()
    assert '#include "../include/rda_hip.h"' in src and "oracle" not in src                     # a caller of the C-ABI and of nothing else
    from rda_planner_amd import _lib
    assert _lib.build_flatten_ext(force=True) is not None
    import importlib
    mod = importlib.import_module("rda_planner_amd._flatten")
    assert callable(mod.flatten)


def test_integration_md_names_every_entry_point_of_the_header():
    """INTEGRATION.md is the map from the C-ABI to the reference interfaces it replaces: no function of include/rda_hip.h may be missing"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = sorted(set(re.findall(r"\b(rda_[a-z0-9_]+)\s*\(", open(os.path.join(root, "include", "rda_hip.h")).read())))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert [n for n in names if n not in doc] == []
