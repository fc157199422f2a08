"""CPU-side checks of the drop-in boundary: the in-tree library loads, exports exactly what
include/cloudmerge.h declares, agrees with the ctypes structs on layout, and fails loudly
without a GPU (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from cloud_merger_amd import build as cm_build
from cloud_merger_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cloudmerge.h")


@pytest.fixture(scope="module")
def lib():
    cm_build.build()
    return capi.load()


def declared_symbols():
    text = open(HEADER).read()
    return re.findall(r"^CM_API\s+[\w\s\*]+?\b(cm_\w+)\s*\(", text, flags=re.M)


def test_header_and_binding_list_agree():
    assert sorted(declared_symbols()) == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_no_oracle_or_cpu_path_in_product():
    nm = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True)
    exported = [l.split()[-1] for l in nm.stdout.splitlines() if len(l.split()) >= 3]       # every defined dynamic symbol: T, W, V, B ...
    assert sorted(exported) == sorted(capi.SYMBOLS), "only the C-ABI is exported (no weak template symbols either)"
    ldd = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "cm_oracle" not in ldd and "libamdhip64" in ldd
    for f in ("cm_api.cpp", "cm_kernels.hip", "cm_kernels.h", "cm_device.h"):
        src = open(os.path.join(ROOT, "cloud_merger_amd", "csrc", f)).read()
        assert "oracle" not in src.lower().replace("oracle/", "")


def test_version_and_status_strings(lib):
    assert lib.cm_version() == 100
    assert capi.status_string(capi.OK) == "CM_OK"
    assert capi.status_string(capi.GRID_OVERFLOW) == "CM_GRID_OVERFLOW"
    assert capi.status_string(capi.NO_DEVICE) == "CM_NO_DEVICE"


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "cloudmerge.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(cm_limits),sizeof(cm_params),sizeof(cm_result),sizeof(cm_stage_times),"
                   "sizeof(cm_zone),sizeof(cm_ground_params),sizeof(cm_ground_plane),sizeof(cm_frame_stats));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(capi.Limits), C.sizeof(capi.Params), C.sizeof(capi.Result), C.sizeof(capi.StageTimes),
                     C.sizeof(capi.Zone), C.sizeof(capi.GroundParams), C.sizeof(capi.GroundPlane), C.sizeof(capi.FrameStats)]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_create_fails_loudly_without_gpu(lib):
    with pytest.raises(capi.CloudMergeError) as e:
        capi.CloudMerger(max_points_total=1000, max_sensors=2)
    assert e.value.status in (capi.NO_DEVICE, capi.HIP_ERROR)


def test_bad_arguments_are_status_codes(lib):
    ctx = C.c_void_p()
    assert lib.cm_create(C.byref(ctx), 0, None) == capi.BAD_ARG
    lim = capi.Limits(0, 0, 10)
    assert lib.cm_create(C.byref(ctx), 0, C.byref(lim)) == capi.BAD_ARG
    lim = capi.Limits(2, 0, 1 << 31)
    assert lib.cm_create(C.byref(ctx), 0, C.byref(lim)) == capi.BAD_ARG
    assert lib.cm_destroy(None) == capi.BAD_ARG
    assert lib.cm_merge_voxelize(None, None, None) == capi.BAD_ARG


def test_constants_mirror_the_header():
    """Status codes, context flags and path flags of the ctypes binding against the #defines / enum of cloudmerge.h."""
    text = open(HEADER).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(CM_\w+)\s+(0x[0-9a-fA-F]+|\d+)u?\b", text)}
    enums = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(CM_[A-Z_]+)\s*=\s*(-?\d+)\s*,", text)}
    for name in ("PROFILE", "LATEST_WINS", "OCCUPANCY"):
        assert getattr(capi, "FLAG_" + name) == defines["CM_FLAG_" + name]
    for name in ("LDS_RANK", "BUCKET", "PREDICTED", "REDONE", "PACKED", "SPLIT"):
        assert getattr(capi, "PATH_" + name) == defines["CM_PATH_" + name]
    for name in ("OK", "EMPTY_INPUT", "GRID_OVERFLOW", "NOT_READY", "SKIPPED", "BAD_ARG", "CAPACITY"):
        assert getattr(capi, name) == enums["CM_" + name]
    assert capi.MAX_SENSORS == defines["CM_MAX_SENSORS"] and capi.NO_FIELD == defines["CM_NO_FIELD"]


def test_scripts_compile():
    """The measurement and differential-run scripts under scripts/ are not imported by the suite (they need a GPU and
    minutes of time): at least keep them syntactically alive."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "scripts", "*.py")))
    assert len(files) >= 8
    for f in files:
        compile(open(f).read(), f, "exec")
