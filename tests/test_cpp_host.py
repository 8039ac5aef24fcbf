"""C++ host mirror (include/vdbhip.hpp = DynamicIndex over the C ABI): compiles with g++ against libvdbhip.so.
CPU: construction fails loudly without a device.  GPU: the reference's own DB scenario through the mirror."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "test_dynamic_index.cpp")
LIBDIR = os.path.join(ROOT, "lab_1806_vec_db_amd")


def _build(tmp_path):
    exe = str(tmp_path / "test_dynamic_index")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, "-L" + LIBDIR, "-lvdbhip",
           "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


def test_cpp_host_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    r = subprocess.run([exe, "--no-gpu"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        assert "no gpu" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_cpp_host_dynamic_index_scenario(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0 and "cpp host ok" in r.stdout, r.stdout + r.stderr
