"""Instruction statistics of the sweep kernel's main loop (run in the build container).
usage: python scripts/isa_stats.py [mangled-name-substring ...]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "geocalib_amd", "csrc", "gclm_pass.hip")
asm = "/tmp/gclm_pass.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", *os.environ.get("EXTRA","").split(), "-S",
                "--cuda-device-only", "-o", asm, src], check=True, capture_output=True)
text = open(asm).read()
wanted = sys.argv[1:] or ["ILi0ELb1ELb1ELb1ELi4E", "ILi1ELb1ELb1ELb1ELi4E"]
for w in wanted:
    m = re.search(r"^(_ZN4gclm\S*sweep_kernel%s\S*):\s*;.*?\n(.*?)\.amdhsa_kernel" % w, text, re.S | re.M)
    if not m:
        print("not found", w); continue
    body = m.group(2).split("\n")
    labels = {mm.group(1): i for i, l in enumerate(body) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = [(labels[mm.group(1)], i) for i, l in enumerate(body)
             if (mm := re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)) and labels.get(mm.group(1), 1 << 30) < i]
    vg = re.search(r"\.vgpr_count:\s+(\d+)", text[m.end():m.end() + 8000])
    nv = re.search(r"; NumVgprs: (\d+)", text[m.start():m.end() + 4000])
    a, b = max(loops, key=lambda t: t[1] - t[0])
    ops = [l.strip().split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    trans = sum(v for k, v in c.items() if re.match(r"v_(rsq|sqrt|rcp|exp|log|sin|cos)", k))
    mov = sum(v for k, v in c.items() if "mov" in k)
    print(f"{w}: loop {len(ops)} instr, VALU {valu} (packed {pk}, trans {trans}, mov {mov}) -> {valu/4:.1f} VALU/px; NumVgprs {nv.group(1) if nv else '?'}")
    print("   ", c.most_common(14))
