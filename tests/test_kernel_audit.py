"""CPU: what the compiler made of the kernels, read from the code objects inside the built library.

Round 3 found a 4 KB LDS array nobody wrote in every `simple_divisional` sweep: LLVM's VectorCombine had pinned four
parameter-block fields in a private array, AMDGPUPromoteAlloca moved it to LDS indexed by the flat thread id, and the flat
thread id cost every wave a read of the dispatch packet in host memory (28 us per single-image launch instead of 10;
DESIGN.md 3.1).  Nothing in the sources shows such a thing, so the build is audited: LDS and scratch of every sweep
instantiation must be what the source asks for."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "geocalib_amd", "lib", "libgeocalib_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_metadata(tmp_path):
    shutil.copy(SO, tmp_path / "lib.so")
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    kernels = {}
    for f in sorted(os.listdir(tmp_path)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], cwd=tmp_path, check=True, capture_output=True,
                               text=True).stdout
        for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+)\s*\n(?:.*\n)*?\s*\.name:\s*(\S+)\s*\n(?:.*\n)*?\s*"
                             r"\.private_segment_fixed_size:\s*(\d+)\s*\n(?:.*\n)*?\s*\.vgpr_count:\s*(\d+)", notes):
            kernels[m.group(2)] = {"lds": int(m.group(1)), "scratch": int(m.group(3)), "vgpr": int(m.group(4))}
    return kernels


@pytest.mark.skipif(not (os.path.exists(SO) and os.path.exists(f"{LLVM}/llvm-readelf")), reason="library or LLVM tools missing")
def test_no_sweep_kernel_carries_an_lds_array_or_scratch_it_was_not_given(tmp_path):
    k = kernel_metadata(tmp_path)
    sweeps = {n: v for n, v in k.items() if "sweep_kernelILi" in n}
    fused = {n: v for n, v in k.items() if "fused_step_kernelILi" in n}
    assert len(sweeps) >= 48 and len(fused) >= 32, (len(sweeps), len(fused))
    # the sweep's only LDS is the 4-wave reduction buffer (4 x NACC floats: 256 B; radial 384 B)
    assert all(v["lds"] <= 384 for v in sweeps.values()), {n: v for n, v in sweeps.items() if v["lds"] > 384}
    # the one-launch-per-step kernels add the update prologue's buffers (stripes of the record reduction, parameter block)
    assert all(v["lds"] <= 2816 for v in fused.values()), {n: v for n, v in fused.items() if v["lds"] > 2816}
    # scratch: none, except general-focal (LOGF = 0: a non-default conf, or the ONE final sweep of a solve with fx != fy)
    # instantiations held to 168 VGPRs: two of simple_divisional (8 / 16 B) and radial's scratch-plane reader (8 B)
    # -- pinned by their exact template arguments <MODEL, HAS_UP, HAS_UPC, HAS_LATC, LOGF = 0, VEC = 4, SLAT, MIRROR = 0>, with the bytes each
    # may spill (a substring match plus a count would let any other instantiation start spilling unnoticed)
    spilling = {re.search(r"sweep_kernelI(\w+?)EEv|fused_step_kernelI(\w+?)EEv", n).group(0): v["scratch"]
                for n, v in {**sweeps, **fused}.items() if v["scratch"] and "ELb1EEEv" not in n}   # (MIRROR = 1, the row pairs: below)
    allowed = {"sweep_kernelILi3ELb1ELb0ELb1ELb0ELi4ELi0ELb0EEEv": 8,      # simple_divisional, no up confidence, general focal
               "sweep_kernelILi3ELb1ELb1ELb1ELb0ELi4ELi0ELb0EEEv": 16,     # simple_divisional, five planes, general focal
               "sweep_kernelILi2ELb1ELb1ELb1ELb0ELi4ELi2ELb0EEEv": 8}      # radial, scratch-plane reader, general focal
    assert all(spilling[n] <= allowed.get(n, 0) for n in spilling), (spilling, allowed)
    # the BASELINE instantiations keep their occupancy: pinhole 80 VGPRs (6 waves / SIMD), simple_radial <= 128 (4 waves)
    main = {m: next(v for n, v in sweeps.items() if f"sweep_kernelILi{m}ELb1ELb1ELb1ELb1ELi4ELi0ELb0E" in n) for m in range(4)}
    assert main[0]["vgpr"] <= 80 and main[1]["vgpr"] <= 128 and main[2]["vgpr"] <= 168 and main[3]["vgpr"] <= 168, main
    # ... and so do the scratch-plane instantiations (SLAT = 1: the first sweep of a solve stores sin(latitude), SLAT = 2: the
    # later sweeps load it) that the distortion models run by default -- same waves per SIMD as the plain sweep, no scratch
    for slat in (1, 2):
        for logf in (0, 1):
            inst = {m: next(v for n, v in sweeps.items() if f"sweep_kernelILi{m}ELb1ELb1ELb1ELb{logf}ELi4ELi{slat}ELb0E" in n) for m in range(4)}
            assert inst[0]["vgpr"] <= 96 and inst[1]["vgpr"] <= 128 and inst[2]["vgpr"] <= 168 and inst[3]["vgpr"] <= 168, (slat, logf, inst)
            assert all(v["scratch"] == 0 for m, v in inst.items() if logf == 1), (slat, logf, inst)
    # the row-pair walkers (MIRROR = 1: radial / simple_divisional, five planes, float4; every SLAT, both focal forms) hold two
    # rows' loads.  simple_divisional: two waves per SIMD (256 VGPRs), never a byte of scratch (held to 168 it spills 276 B and
    # runs 41 % slower, profiles/r06_variant_row_pairs.log).  radial, log-focal (every loop sweep of the default conf): held to
    # 168 VGPRs = three waves, which is what makes its row pairs pay (-3.9 % against -1.3 % at two waves); 8-32 B of scratch, one
    # 8-byte reload per iteration of the hot loop.  radial, general focal: two waves, no scratch (216-248 B at 168)
    pairs = {n: v for n, v in sweeps.items() if re.search(r"sweep_kernelILi[23]ELb1ELb1ELb1ELb[01]ELi4ELi[012]ELb1EEEv", n)}
    assert len(pairs) == 12 and all(v["lds"] <= 384 for v in pairs.values()), pairs
    assert all(v["vgpr"] <= 256 and v["scratch"] == 0 for n, v in pairs.items() if "ILi3E" in n), pairs
    assert all(v["vgpr"] <= 168 and v["scratch"] <= 32 for n, v in pairs.items() if "ILi2ELb1ELb1ELb1ELb1E" in n), pairs
    assert all(v["vgpr"] <= 256 and v["scratch"] == 0 for n, v in pairs.items() if "ILi2ELb1ELb1ELb1ELb0E" in n), pairs
    assert not any(re.search(r"sweep_kernelILi[01]E\w*ELb1EEEv", n) for n in sweeps), "pinhole / simple_radial have no row-pair walker"
