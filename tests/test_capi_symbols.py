"""The built librda_hip.so loads without a GPU and exports every symbol include/rda_hip.h declares."""
import ctypes
import os
import re
import pytest

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


def _header_struct_fields(name):
    """[(field, 'int' | 'double', length)] of `typedef struct <name> { ... } <name>;` in include/rda_hip.h, in declaration order"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "rda_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        m = re.match(r"\s*(int32_t|double)\s+(.*)", decl.strip(), re.S)
        if not m:
            continue
        for item in m.group(2).split(","):
            mm = re.match(r"\s*(\w+)\s*(?:\[(\d+)\])?\s*$", item)
            out.append((mm.group(1), "int" if m.group(1) == "int32_t" else "double", int(mm.group(2) or 1)))
    return out


@pytest.mark.parametrize("cname,pyname", [("rda_opts", "Opts"), ("rda_cfg", "Cfg"), ("rda_info", "Info")])
def test_ctypes_structs_mirror_the_header_field_by_field(cname, pyname):
    """the ctypes mirrors of the C-ABI structs (rda_planner_amd/_capi.py) against include/rda_hip.h: names, order, types, array lengths -
    a field added on one side only shifts everything behind it silently"""
    import ctypes as C
    from rda_planner_amd import _capi
    want = _header_struct_fields(cname)
    got = []
    for fname, ftype in getattr(_capi, pyname)._fields_:
        length = getattr(ftype, "_length_", 1)
        base = getattr(ftype, "_type_", ftype) if length > 1 else ftype
        got.append((fname, "int" if base is C.c_int else "double", length))
        assert base in (C.c_int, C.c_double), (fname, base)
    assert got == want
