"""Instruction statistics of the sweep kernel's main loop (run in the build container: hipcc cross-compiles).

usage: python scripts/isa_stats.py [--json OUT] [--dump SYMBOL_SUBSTRING] [--slat 0|1|2] [--mirror 0|1] [MODEL ...]
       MODEL in {0 pinhole, 1 simple_radial, 2 radial, 3 simple_divisional}; default: all four.
Looks at the instantiation sweep_kernel<MODEL, HAS_UP=1, HAS_UPC=1, HAS_LATC=1, LOGF=1, VEC=4> (the loop sweep of the
default conf with both confidences: what bench.py runs) by its mangled template arguments, independent of how many
template parameters precede / follow them; --slat picks the SLAT instantiation (0: sin(latitude) computed per sweep, 1: computed
and stored into the scratch plane = the first sweep of a solve, 2: loaded from it = every later sweep; gclm_pass.hip: row_math).
--mirror 1 picks the row-pair walker (radial / simple_divisional: one loop iteration = two rows = 8 pixels per lane).
EXTRA="-D..." adds compile flags (A/B switches of gclm_pass.hip)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "geocalib_amd", "csrc", "gclm_pass.hip")
asm = "/tmp/gclm_pass.s"
NAMES = {0: "pinhole", 1: "simple_radial", 2: "radial", 3: "simple_divisional"}


def main():
    args = sys.argv[1:]
    out_json = dump = None
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    if "--dump" in args:
        i = args.index("--dump"); dump = args[i + 1]; del args[i:i + 2]
    slat = 0
    if "--slat" in args:
        i = args.index("--slat"); slat = int(args[i + 1]); del args[i:i + 2]
    mirror = 0
    if "--mirror" in args:
        i = args.index("--mirror"); mirror = int(args[i + 1]); del args[i:i + 2]
    models = [int(a) for a in args] or [0, 1, 2, 3]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast-honor-pragmas",
                    "-fno-slp-vectorize", *os.environ.get("PASS_BASE_EXTRA", "-mllvm -disable-vector-combine").split(), *os.environ.get("EXTRA", "").split(), "-S", "--cuda-device-only", "-o", asm, src],
                   check=True, capture_output=True)
    text = open(asm).read()
    results = {}
    for m_id in models:
        # sweep_kernel<MODEL, true, true, true, true(LOGF), 4(VEC)>
        pat = r"^(_ZN4gclm\S*sweep_kernelILi%dELb1ELb1ELb1ELb1ELi4ELi%dELb%dE\S*):\s*;.*?\n(.*?)\.amdhsa_kernel" % (m_id, slat, mirror)
        m = re.search(pat, text, re.S | re.M)
        if not m:
            print(f"model {m_id}: instantiation not found (template signature changed?)")
            sys.exit(1)
        body = m.group(2).split("\n")
        labels = {mm.group(1): i for i, l in enumerate(body) if (mm := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = [(labels[mm.group(1)], i) for i, l in enumerate(body)
                 if (mm := re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)) and labels.get(mm.group(1), 1 << 30) < i]
        tail = text[m.end():m.end() + 8000]
        nv = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", tail)
        sc = re.search(r"; ScratchSize: (\d+)", text[m.start():m.end() + 6000])
        # simple_divisional has TWO row loops (always-guarded for k ~ 0 = the first sweep, guard-free for the rest): the
        # common one is the loop with fewer v_cndmask; every other model has one big loop
        bigloops = [t for t in loops if t[1] - t[0] > 100] or loops
        a, b = min(bigloops, key=lambda t: sum("v_cndmask" in l for l in body[t[0]:t[1] + 1]))
        if mirror:      # three row loops (unshared first iteration is straight-line; unshared loop; shared loop): the hot one is the shortest
            a, b = min(bigloops, key=lambda t: sum(1 for l in body[t[0]:t[1] + 1] if l.startswith("\tv_")))
        other_loops = [sum(1 for l in body[t[0]:t[1] + 1] if l.startswith("\tv_")) for t in bigloops if t != (a, b)]
        # The loop holds side blocks that a wave only enters in rare cases: the latitude fold (|lat| > pi/2: marked by
        # v_rndne) and, for simple_divisional, the guarded copy of the body (the one of its two big blocks with MORE
        # v_cndmask).  Split the loop into basic blocks and count the COMMON path = everything but those.
        blocks, cur = [], []
        for l in body[a:b + 1]:
            if re.match(r"^(\.LBB\d+_\d+):", l) and cur:
                blocks.append(cur); cur = []
            if l.startswith("\t") and not l.strip().startswith((".", ";")):
                cur.append(l.strip().split()[0])
                if l.strip().startswith(("s_cbranch", "s_branch")):
                    blocks.append(cur); cur = []
        if cur:
            blocks.append(cur)
        nvalu = lambda ops_: sum(1 for o in ops_ if o.startswith("v_"))  # noqa: E731
        rare = [i for i, blk in enumerate(blocks) if any(o.startswith("v_rndne") for o in blk)]
        rare += [i for i, blk in enumerate(blocks) if i not in rare and 20 < nvalu(blk) < 200
                 and sum(o.startswith("v_cndmask") for o in blk) >= 10]          # the guard patches of simple_divisional
        ops_all = [o for blk in blocks for o in blk]
        ops = [o for i, blk in enumerate(blocks) if i not in rare for o in blk]
        rare_valu = {i: nvalu(blocks[i]) for i in rare}
        c = collections.Counter(ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
        trans = sum(v for k, v in c.items() if re.match(r"v_(rsq|sqrt|rcp|exp|log|sin|cos)", k))
        mov = sum(v for k, v in c.items() if "mov" in k)
        loads = sum(v for k, v in c.items() if k.startswith("global_load"))
        mfma = sum(v for k, v in c.items() if "mfma" in k)
        results[NAMES[m_id]] = {"loop_instructions": len(ops), "loop_instructions_with_rare_blocks": len(ops_all),
                                "rare_blocks_valu": sorted(rare_valu.values()), "other_loops_valu": other_loops, "cndmask": c.get("v_cndmask_b32_e32", 0) + c.get("v_cndmask_b32_e64", 0),
                                "valu": valu, "valu_packed": pk, "valu_transcendental": trans,
                                "valu_mov": mov, "global_loads": loads, "mfma": mfma, "pixels_per_iteration": 8 if mirror else 4,
                                "vgprs": int(nv.group(1)) if nv else None, "scratch_bytes": int(sc.group(1)) if sc else None,
                                "top": c.most_common(12)}
        print(f"{NAMES[m_id]}: loop {len(ops)} instr, VALU {valu} (packed {pk}, trans {trans}, mov {mov}), "
              f"{loads} global loads, {mfma} mfma -> {valu / (8 if mirror else 4):.1f} VALU/px; rare blocks (VALU) {sorted(rare_valu.values())}, other loops {other_loops}; VGPRs {nv.group(1) if nv else '?'}, "
              f"scratch {sc.group(1) if sc else '?'} B")
        print("   ", c.most_common(12))
        if dump and dump in m.group(1):
            print("\n".join(body[a:b + 1]))
    if out_json:
        with open(out_json, "w") as fh:
            json.dump({"kernel": f"gclm::sweep_kernel<MODEL, up, up_conf, lat_conf, log-focal, float4, SLAT={slat}>", "flags": os.environ.get("EXTRA", ""),
                       "models": results}, fh, indent=1)


if __name__ == "__main__":
    main()
