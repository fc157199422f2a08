"""Builds libcloudmerge_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.

    python -m cloud_merger_amd.build [--force] [--save-temps]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcloudmerge_hip.so")
# The test build: the same sources with -DCM_TEST_HOOKS (a hook that makes a global pass mis-rank two records, so that the
# finish's order check can be tested). Loaded only by the one test that asks for it (CM_LIB_VARIANT=testhooks); the
# shipped library holds no such code.
HOOKS_LIB_PATH = os.path.join(LIB_DIR, "libcloudmerge_hip_testhooks.so")
SOURCES = ["cm_kernels.hip", "cm_kernels_v2.hip", "cm_kernels_v3.hip", "cm_kernels_v4.hip", "cm_kernels_ground.hip", "cm_api.cpp"]
HEADERS = ["cm_device.h", "cm_kernels.h", "cm_common.hpp", "cloudmerge.map", os.path.join("..", "..", "include", "cloudmerge.h")]

# -ffp-contract=off / -fno-fast-math: occupancy must match the reference bit for bit, so no FMA
# contraction on the device or in the host-side quaternion/grid arithmetic (SURVEY.md §7 hard part 1).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-fvisibility=hidden",
         "-Wl,-rpath,/opt/rocm/lib"]
# Only the C-ABI leaves the library: a linker version script keeps the weak libstdc++ template instantiations (std::string,
# std::vector helpers) local, which -fvisibility=hidden does not reach.
VERSION_SCRIPT = os.path.join(CSRC, "cloudmerge.map")


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, save_temps=False, verbose=False, test_hooks=False):
    out = HOOKS_LIB_PATH if test_hooks else LIB_PATH
    if not force and not needs_build(out):
        return out
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + ["-Wl,--version-script=" + VERSION_SCRIPT, "-I", CSRC, "-o", out]
    if test_hooks:
        cmd += ["-DCM_TEST_HOOKS"]
    if os.environ.get("CM_PHASE_TIMING") == "1":       # experiment build: scripts/phase_times.py (never the shipped one)
        cmd += ["-DCM_PHASE_TIMING"]
    if save_temps:
        tmp = os.path.join(HERE, "build_tmp")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building " + os.path.basename(out))
    if verbose or save_temps:
        sys.stderr.write(r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True,
                test_hooks="--test-hooks" in sys.argv))
