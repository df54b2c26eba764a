"""The C++ host mirror (tests/cpp/friedrich.hpp) replays the reference's doctests / src/main.rs on the GPU and is
checked against the oracle's golden values."""
import json
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_mirror_readme_example(tmp_path):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v1.json")))["readme_default"]
    exe = str(tmp_path / "test_friedrich_hpp")
    lib_dir = os.path.join(ROOT, "friedrich_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "test_friedrich_hpp.cpp"),
           "-o", exe, "-L" + lib_dir, "-lfriedrich_amd", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir,
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe, str(ref["iterations"]), repr(ref["predict_1"]), repr(ref["variance_1"]), repr(ref["noise"]),
                          repr(ref["prior"])], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
