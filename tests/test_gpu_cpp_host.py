"""The C++ host side above the C-ABI (lidar-slam_amd/host/lsa_plugins.hpp: the mirror of the reference's
RegistrationInterface / CloudFilterInterface plugin classes) driven like FrontEnd::Update drives the reference's plugins,
checked against the oracle's C entry points.  Compiled with plain g++ (no HIP headers needed above the boundary)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "lidar-slam_amd", "host"),
           "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"),
           "-L" + os.path.join(ROOT, "lidar-slam_amd"), "-llidar_slam_amd", "-L" + os.path.join(ROOT, "oracle"), "-llsa_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "lidar-slam_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib",
           "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_cpp_host_mirror_compiles_without_gpu(tmp_path):
    """not gpu: the adaptor layer compiles and links against the C-ABI with a plain C++ compiler."""
    import oracle_py
    oracle_py.build(ref=False)
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_host_mirror_matches_oracle(tmp_path):
    import oracle_py
    oracle_py.build(ref=False)
    exe = _build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ALL OK" in p.stdout
