"""Build libfriedrich_amd.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m friedrich_amd.build [--force]

Every translation unit under csrc/ is compiled with --offload-arch=gfx950 and linked into
friedrich_amd/lib/libfriedrich_amd.so.  No torch, no cmake: hipcc + libamdhip64 only (librccl is dlopen'ed).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libfriedrich_amd.so")
SOURCES = ["ctx.hip", "gram.hip", "gemm_f64.hip", "potf2.hip", "util.hip", "chol.hip", "gp.hip", "grad.hip", "comm.hip", "trsv.hip", "trsm_narrow.hip", "prior.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = os.environ.get("FR_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _version_script():
    """Linker version script: the dynamic symbol table holds exactly the functions include/friedrich_amd.h declares."""
    import re

    hdr = open(os.path.join(ROOT, "include", "friedrich_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", hdr)))
    path = os.path.join(OBJ, "exports.map")
    text = "{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local: *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "friedrich_amd.h"))
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    objs = []
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            jobs.append([HIPCC, "-x", "hip"] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    vs = _version_script()
    if force or jobs or _newer(LIB, objs + [vs]):
        # Linked WITHOUT a DT_NEEDED on libamdhip64: the HIP runtime is whichever one the host process already
        # has (torch bundles its own copy; two HSA runtimes in one process cannot both open the GPU).
        # Python: friedrich_amd._capi.load() preloads it; C/C++/Rust hosts link -lamdhip64 themselves.
        run([CLANGXX, "-shared", "-fPIC", "-Wl,--version-script=" + vs, "-o", LIB] + objs + ["-ldl", "-lpthread"])
    # bench.py's measured FP64 matrix-core ceiling: the instruction loop of scripts/mfma_f64_peak.hip as a
    # small executable next to the library, so that it travels to the GPU box like the built .so
    probe_src = os.path.join(ROOT, "scripts", "mfma_f64_peak.hip")
    probe = os.path.join(HERE, "lib", "mfma_f64_peak")
    if os.path.exists(probe_src) and (force or _newer(probe, [probe_src])):
        run([HIPCC, "--offload-arch=gfx950", "-O3", "-o", probe, probe_src])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
