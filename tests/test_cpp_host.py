"""The C++ host mirror (include/oxmpl/oxmpl.hpp) compiled into the C++ twin of the reference's
oxmpl/tests/rrt_rvss_tests.rs.  CPU: it must build, link against the C ABI, and refuse to plan
without a GPU (exit 77).  GPU: the reference's assertions must hold (exit 0)."""
import os
import subprocess

import pytest

from oxmpl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="test_rrt_rvss"):
    capi.build_library()
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "oxmpl_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L", libdir, "-loxmpl_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_builds_and_refuses_without_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    if r.returncode == 77:
        assert "refused as designed" in r.stdout


@pytest.mark.gpu
def test_cpp_twin_of_reference_rrt_test(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RRT planner test passed!" in r.stdout


def test_cpp_prm_mirror_builds_and_refuses_without_gpu(tmp_path):
    exe = _build(tmp_path, "test_prm_rvss")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    if r.returncode == 77:
        assert "refused as designed" in r.stdout


@pytest.mark.gpu
def test_cpp_twin_of_reference_prm_test(tmp_path):
    exe = _build(tmp_path, "test_prm_rvss")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PRM planner test passed!" in r.stdout
