#!/usr/bin/env python
"""Scan the gfx950 disassembly of every translation unit for the packed-fp32 form that miscomputes beside another wave's bf16 MFMAs
(round 6, tools/probes/pk_hazard/xwave2.hip): a v_pk_*_f32 whose LOW result selects the HIGH half of its second source, i.e.
`op_sel:[x,1]` (with any op_sel_hi) -- 0.1-0.3 % wrong low halves while a v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 of another wave runs on the
same SIMD; every other selection, and an f32 MFMA as the neighbour, is right.  The SLP vectoriser emits the form for horizontal and crossed
pairings (`v_pk_add_f32 v, v, v op_sel:[0,1] op_sel_hi:[1,0]`); the library is built with -fno-slp-vectorize.
python tools/scan_pk_opsel_hazard.py   -> per unit counts, last line "total N"."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_abi as T

obj_dir = os.path.join(ROOT, "siu3r_amd", "csrc", "_obj")
pk = re.compile(r"^\s*(v_pk_\w+_f32)\b(.*)$")
total = 0
for unit in sorted(f for f in os.listdir(obj_dir) if f.endswith(".o")):
    try:
        asm = T._device_disassembly(unit)
    except Exception:
        continue  # (no device code)
    fn, hits, packed = "?", {}, 0
    for l in asm.split("\n"):
        l = re.sub(r"\s*//.*", "", l)
        m = re.match(r"^[0-9a-f]{16} <(.*)>:", l)
        if m:
            fn = m.group(1)
            continue
        m = pk.match(l)
        if not m:
            continue
        packed += 1
        o = re.search(r"op_sel:\[([01,]+)\]", m.group(2))
        if o:
            bits = o.group(1).split(",")
            if len(bits) >= 2 and bits[1] == "1":
                hits.setdefault(fn, []).append(l.strip())
    n = sum(len(v) for v in hits.values())
    total += n
    print(f"{unit}: {packed} packed fp32 instructions, {n} with op_sel[1] = 1" + "".join(f"\n     {len(v):4d} {k[:80]}  e.g. {v[0][:90]}" for k, v in list(hits.items())[:6]))
print(f"total {total}")
