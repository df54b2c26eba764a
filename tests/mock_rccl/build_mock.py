"""Build tests/mock_rccl/librccl_mock.so (TEST-ONLY stand-in for librccl: thread-ranks sharing one device; see rccl_mock.hip).
Linked like the product library: no DT_NEEDED on libamdhip64 -- it binds to the HIP runtime of the process that loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_mock.hip")
OBJ = os.path.join(HERE, "rccl_mock.o")
LIB = os.path.join(HERE, "librccl_mock.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = os.environ.get("FR_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    for cmd in ([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I/opt/rocm/include", "-c", SRC, "-o", OBJ],
                [CLANGXX, "-shared", "-fPIC", "-o", LIB, OBJ, "-lpthread"]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building the RCCL mock failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
