"""The throughput kernels must not spill: a register spill in the trailing-update kernel shows up as doubled WRITE_SIZE and a
few per cent of a fit, and nothing else notices (round 4: a run-time loop around the tile function's one call site cost
syrk_lower_f64_kernel 17 registers and 228 scratch instructions until it became a template parameter).  Compiles gemm_f64.hip
and gram.hip for gfx950 with the build's flags and reads the compiler's own resource remarks -- no GPU needed."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def resources(src):
    cmd = [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "friedrich_amd", "csrc"), "-I/opt/rocm/include", "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(ROOT, "friedrich_amd", "csrc", src), "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur:
            res[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_throughput_kernels_do_not_spill():
    r = resources("gemm_f64.hip")
    checked = 0
    for name, v in r.items():
        if "syrk_lower_f64_kernel" in name or "15gemm_f64_kernel" in name or "19gemm_f64_m32_kernel" in name:
            assert v["ScratchSize"] == 0, (name, v)
            assert v["Occupancy"] >= 2, (name, v)
            checked += 1
    assert checked == 9, sorted(r)
    # the resident variants of round 5 (a loop around the tile function): the loop costs per-lane address arithmetic that is hoisted
    # out of it and reloaded at the start of a tile -- 100 B of scratch per lane, none of it inside the K-loop (ISA inspected);
    # more than that means the accumulators or the prefetch registers have started to spill.  Two workgroups per CU as well.
    persist = {n: v for n, v in r.items() if "persist" in n}
    assert len(persist) == 2, sorted(r)
    for name, v in persist.items():
        assert v["ScratchSize"] <= 128 and v["Occupancy"] >= 2, (name, v)
    rs16 = [v for n, v in r.items() if "rows_solve16" in n]
    assert rs16 and rs16[0]["ScratchSize"] == 0, rs16  # (the LDS-resident panel-row solve keeps 64 fragment registers in flight)
    # the resident panel chain (round 6): the flat diagonal-block body inside a loop next to the slab / row-group roles -- with the body's
    # per-lane addresses hoisted out of that loop it spilled 32 registers (124 B per lane) and the diagonal block took 38 us instead of 28
    p = resources("potf2.hip")
    chain = [v for n, v in p.items() if "panel_chain_kernel" in n]
    assert chain and chain[0]["ScratchSize"] == 0 and chain[0]["Occupancy"] >= 2, chain
    flat = [v for n, v in p.items() if "potf2_flat_kernel" in n]
    assert flat and flat[0]["ScratchSize"] == 0, flat
    g = resources("gram.hip")
    leaf = [v for n, v in g.items() if "gram_kernel" in n and "ILi1ELi2" in n]  # the squared-exponential leaf of the bench
    assert leaf and all(v["ScratchSize"] == 0 for v in leaf), g
