"""Build container: is the DEVICE code of csrc/gclm_pass.hip at git revision REV, instruction for instruction, the code of the
working tree?  (Round 6 pruned the closed A/B switches from the hot file; identical device assembly means identical bits and
identical speed by construction -- stronger than the ISA statistics and than any measured A/B.)

usage: python scripts/asm_equal.py REV [FILE=gclm_pass.hip]
Compiles FILE of both trees with the Makefile's flags to gfx950 assembly (-S --cuda-device-only), drops comments, debug /
file directives and the compilation-unit id symbol (a hash of the source text), and compares the rest line by line."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast-honor-pragmas", "-S", "--cuda-device-only"]
PASS = ["-fno-slp-vectorize", "-mllvm", "-disable-vector-combine"]


def assembly(tree, name):
    out = tempfile.mktemp(suffix=".s")
    flags = FLAGS + (PASS if name == "gclm_pass.hip" else [])
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-o", out, os.path.join(tree, "geocalib_amd", "csrc", name)], check=True,
                   capture_output=True)
    lines = []
    for ln in open(out):
        ln = ln.rstrip() if ln.lstrip().startswith(".") else ln.split(";")[0].rstrip()
        if not ln.strip() or re.match(r"\s*\.(file|loc|ident|cfi|section\s+\.debug)", ln) or "__hip_cuid_" in ln:
            continue
        lines.append(ln)
    os.unlink(out)
    return lines


def main():
    rev = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "gclm_pass.hip"
    with tempfile.TemporaryDirectory() as old:
        for path in ("geocalib_amd/csrc", "include"):
            tar = subprocess.run(["git", "-C", ROOT, "archive", rev, path], check=True, capture_output=True).stdout
            subprocess.run(["tar", "-x", "-C", old], input=tar, check=True)
        a, b = assembly(old, name), assembly(ROOT, name)
        n_old = sum(1 for _ in open(os.path.join(old, "geocalib_amd", "csrc", name)))
    n_new = sum(1 for _ in open(os.path.join(ROOT, "geocalib_amd", "csrc", name)))
    kernels = sum(1 for ln in a if ln.strip().startswith(".amdhsa_kernel"))
    same = a == b
    print(f"{name}: {rev} ({n_old} source lines) vs working tree ({n_new} source lines): {len(a)} / {len(b)} assembly lines, "
          f"{kernels} kernels, {'IDENTICAL' if same else 'DIFFERENT'}")
    if not same:
        import difflib
        for ln in list(difflib.unified_diff(a, b, lineterm="", n=0))[:40]:
            print("   ", ln)
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
