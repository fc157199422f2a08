"""The reference-side binding and the roscpp adapter (cloud_merger_amd/host/ros1_node.cpp) are the only
code here that needs ROS, which this image does not have. They are syntax-checked against include/cloudmerge.h with
tests/ros_stub/ (declarations of the few ROS names they use) in place of the ROS headers — a compile check of OUR use of
OUR header, nothing of the reference is built — and INTEGRATION.md must print the checked snippet verbatim."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cloud_merger_amd", "host")
STUB = os.path.join(ROOT, "tests", "ros_stub")


def _syntax_only(path, *defs):
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", STUB, "-I", os.path.join(ROOT, "include"),
                        "-I", HOST, *defs, path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_integration_snippet_compiles_against_the_header():
    _syntax_only(os.path.join(HOST, "integration_snippet.cpp"))


def test_integration_md_prints_the_checked_snippet():
    src = open(os.path.join(HOST, "integration_snippet.cpp")).read()
    snippet = src.split("// [snippet-begin]\n")[1].split("// [snippet-end]")[0]
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert snippet.strip() in md, "INTEGRATION.md §A and cloud_merger_amd/host/integration_snippet.cpp have drifted apart"


def test_ros1_adapter_compiles_against_the_header():
    _syntax_only(os.path.join(HOST, "ros1_node.cpp"), "-DCLOUDMERGE_WITH_ROS")
