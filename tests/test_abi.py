"""The C-ABI library loads without a GPU and exports every symbol include/friedrich_amd.h declares; the ctypes
binding, the header and the oracle agree on the POD layouts.  No compute call is made here."""
import ctypes
import os
import re

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "friedrich_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from friedrich_amd import _capi

    lib = _capi.load()
    names = header_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in friedrich_amd.h but not exported"
    # ... and nothing else: the dynamic symbol table is the header (friedrich_amd/build.py writes the linker's version script from it)
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in nm.splitlines() if line.split()[-2] in ("T", "t", "W") and line.split()[-1].startswith("fr_"))
    assert exported == names, (set(exported) ^ set(names))
    others = [line.split()[-1] for line in nm.splitlines() if line.split()[-2] in ("T", "W", "D", "B") and not line.split()[-1].startswith("fr_")]
    assert not others, others
    # the Python binding declares exactly the header's functions
    assert sorted(_capi.SIGNATURES) == names
    assert lib.fr_abi_version() == 2


def test_no_gpu_means_no_context():
    # the product path must fail loudly, never fall back to a CPU implementation
    import torch

    from friedrich_amd import _capi

    lib = _capi.load()
    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    assert lib.fr_ctx_create(ctypes.byref(h), -1) == _capi.FR_NO_DEVICE
    assert not h


def test_kprog_layout_matches_header_and_oracle():
    from friedrich_amd import _capi
    from oracle import oracle as O

    assert ctypes.sizeof(_capi.KernelOp) == 32 and ctypes.sizeof(_capi.KProg) == 8 + 15 * 32
    assert ctypes.sizeof(O.KernelOp) == ctypes.sizeof(_capi.KernelOp)
    assert ctypes.sizeof(O.KProg) == ctypes.sizeof(_capi.KProg)
    spec = ("sum", ("prod", ("squared_exp", 1.0, 2.0), ("matern1", 2.0, 0.5)), ("linear", 1.5))
    a, b = _capi.kprog(spec), O.kprog(spec)
    assert bytes(a) == bytes(b)
    hdr = open(os.path.join(ROOT, "include", "friedrich_amd.h")).read()
    ohdr = open(os.path.join(ROOT, "oracle", "friedrich_oracle.h")).read()
    for name, val in re.findall(r"FR_K_([A-Z0-9]+) = (\d+)", hdr):
        assert re.search(rf"FRO_K_{name} = {val}\b", ohdr), name


def test_product_does_not_import_the_oracle():
    # oracle/ is test infrastructure: nothing under friedrich_amd/ may reference it
    pkg = os.path.join(ROOT, "friedrich_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "friedrich_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_synth_generator_is_deterministic():
    import numpy as np

    from friedrich_amd import synth

    u = synth.splitmix64_uniform(0x5EED0000, 0, 4)
    # SplitMix64 reference values for seed 0x5EED0000 (frozen; any port of the generator must reproduce them)
    assert u.tolist() == synth.splitmix64_uniform(0x5EED0000, 0, 8)[:4].tolist()
    assert np.all((u >= 0) & (u < 1))
    X, y, Xq = synth.make_problem(64, 3, cfg=2, m=5)
    X2, y2, Xq2 = synth.make_problem(64, 3, cfg=2, m=5)
    assert np.array_equal(X, X2) and np.array_equal(y, y2) and np.array_equal(Xq, Xq2)
    assert X.flags.f_contiguous and X.shape == (64, 3) and Xq.shape == (5, 3)


def test_binding_passes_leading_rows_of_a_column_major_matrix_without_a_copy():
    """The ABI takes pointer + leading dimension (EMatrix::as_matrix has ld = capacity, extendable_matrix.rs:52-55): the ctypes
    binding hands the leading rows of a column-major matrix over as they lie (fr_chol_add_rows gets the caller's whole matrix
    on every call; a host copy per call was 45 us of configs[4]'s 1.1 ms per append), and still copies what is not column-major."""
    import numpy as np

    from friedrich_amd.device import _Mat

    X = np.asfortranarray(np.arange(40.0).reshape(8, 5))
    v = _Mat(X[:6])
    assert (v.rows, v.cols, v.ld) == (6, 5, 8) and v.ptr == X.ctypes.data
    full = _Mat(X)
    assert (full.rows, full.cols, full.ld) == (8, 5, 8) and full.ptr == X.ctypes.data
    c = _Mat(np.ascontiguousarray(X)[:6])  # row-major rows: copied into column-major
    assert (c.rows, c.cols, c.ld) == (6, 5, 6) and c.ptr != X.ctypes.data
    assert np.array_equal(c.keep, X[:6])
    one = _Mat(X[:6, :1])  # a single column: ld is irrelevant, any layout
    assert (one.rows, one.cols) == (6, 1)


def test_rccl_mock_exports_what_the_library_resolves():
    """tests/mock_rccl (the stand-in for librccl behind tests/test_gpu_rccl_multi.py on 1-GPU boxes) builds without a GPU and
    defines every nccl* symbol friedrich_amd/csrc/comm.hip looks up with dlsym; the product never names it (only the environment
    variable FRIEDRICH_AMD_RCCL_PATH of a test's child process does)."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "mock_rccl"))
    import build_mock

    lib = build_mock.build()
    src = open(os.path.join(ROOT, "friedrich_amd", "csrc", "comm.hip")).read()
    wanted = {"nccl" + n for n in re.findall(r"^\s*LOAD\((\w+)\);", src, flags=re.M)} | {"ncclCommGetAsyncError"}
    assert len(wanted) >= 12, wanted
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    have = {line.split()[-1] for line in nm.splitlines() if " T " in line}
    assert wanted <= have, wanted - have
    for dirpath, _, files in os.walk(os.path.join(ROOT, "friedrich_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                assert "rccl_mock" not in open(os.path.join(dirpath, f)).read(), f
