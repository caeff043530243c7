"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU and exports
every symbol include/m3pc_hip.h declares; the ctypes struct layouts match the header."""
import ctypes
import os
import re

import pytest

from m3pc_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    path = build.build_library()
    return capi.load_library(path)


def _declared():
    src = open(os.path.join(ROOT, "include", "m3pc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(m3pc_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/m3pc_hip.h but not exported"
    assert set(names) == set(capi.EXPORTS)


def test_the_library_exports_its_c_abi_and_nothing_else(lib):
    """-fvisibility=hidden + csrc/exports.map: no C++ internal (launchers, kernel handles) in the dynamic symbol table; the
    defined dynamic symbols of the product library are exactly the entry points of include/m3pc_hip.h."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert not [n for n in names if n.startswith("_Z")], [n for n in names if n.startswith("_Z")][:5]
    assert names == sorted(capi.EXPORTS)


def test_abi_version_and_error_string(lib):
    assert lib.m3pc_abi_version() == capi.ABI_VERSION
    assert isinstance(lib.m3pc_last_error(), bytes)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Dims) == 12 * 4 and capi.Dims.max_goal_batch.offset == 44  # (ABI v4: max_goal_batch appended)
    assert ctypes.sizeof(capi.PlanArgs) == 6 * 4 + 3 * 8 + 2 * 4 + 8 + 2 * 4
    assert capi.PlanArgs.flags.offset == 64 and capi.PlanArgs.window.offset == 68
    assert capi.PlanArgs.lmbda.offset == 24 and capi.PlanArgs.rtg.offset == 40
    assert capi.PlanArgs.slot.offset == 48 and capi.PlanArgs.returns_f64.offset == 52 and capi.PlanArgs.returns.offset == 56
    assert ctypes.sizeof(capi.NamedTensor) == 32
    hdr = open(os.path.join(ROOT, "include", "m3pc_hip.h")).read()
    assert int(re.search(r"#define M3PC_SLOTS (\d+)", hdr).group(1)) == capi.SLOTS
    assert int(re.search(r"#define M3PC_ABI_VERSION (\d+)", hdr).group(1)) == capi.ABI_VERSION


def test_product_library_has_no_debug_hooks_and_reads_no_environment(lib):
    """The kernel-level hooks (include/m3pc_hip_debug.h) and the environment A/B switches live in the lab build only."""
    src = open(os.path.join(ROOT, "include", "m3pc_hip_debug.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    dbg = sorted(set(re.findall(r"\b(m3pc_debug_[a-z_0-9]+)\s*\(", src)))
    assert len(dbg) >= 8
    for n in dbg:
        assert not hasattr(lib, n), f"{n} is exported by the product library"
    blob = open(build.LIB, "rb").read()
    assert b"getenv" not in blob and b"M3PC_NO_" not in blob
    lab = capi.load_library(build.build_library(lab=True))
    for n in dbg:
        assert hasattr(lab, n), f"{n} declared in m3pc_hip_debug.h but missing from the lab build"


def test_argument_validation_without_gpu(lib):
    # null arguments are rejected before any HIP call is made
    assert lib.m3pc_create(None, 0, None) == -1
    assert b"null" in lib.m3pc_last_error()
    assert lib.m3pc_destroy(None) == 0


def test_no_cpu_fallback_in_package():
    """The product path must not import the oracle or silently run on the CPU."""
    pkg = os.path.join(ROOT, "m3pc_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in text and "from oracle" not in text and "mtm_oracle" not in text, fn
