#!/usr/bin/env python
"""Scan the gfx950 disassembly of every translation unit for the sequence that corrupted the pre-split planes (round 4): a 12 / 16-byte
buffer / global store whose data registers are written again by one of the next few instructions with no wait state in between.  With an
SGPR `soffset` hipcc inserts no s_nop (the ISA manual: "no wait states required"), and on gfx950 the store of the 256 x 256 tile's row
pass read the overwritten register in lanes 12..15 of every row.  python tools/scan_store_hazard.py [window=2]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_abi as T

WINDOW = int(sys.argv[1]) if len(sys.argv) > 1 else 2
obj_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "siu3r_amd", "csrc", "_obj")
store = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$")
vrange = re.compile(r"v\[(\d+):(\d+)\]")
total = 0
for unit in sorted(f for f in os.listdir(obj_dir) if f.endswith(".o")):
    try:
        asm = T._device_disassembly(unit)
    except Exception:
        continue  # (no device code)
    fn = "?"
    lines = [re.sub(r"\s*//.*", "", l) for l in asm.split("\n")]
    hits = 0
    for i, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]{16} <(.*)>:", l)
        if m:
            fn = m.group(1)
            continue
        m = store.match(l)
        if not m:
            continue
        ops = m.group(2)
        if m.group(1).startswith("buffer"):
            parts = [x.strip() for x in ops.split(",")]
            data, soff = parts[0], parts[3].split()[0] if len(parts) > 3 else ""
            sgpr_soffset = soff.startswith("s") or soff.startswith("m0") or soff.startswith("ttmp")
        else:
            parts = [x.strip() for x in ops.split(",")]
            data, sgpr_soffset = parts[1], False  # global_store: vaddr, vdata, saddr (the compiler keeps the manual's one wait state)
        r = vrange.search(data)
        if not r:
            continue
        lo, hi = int(r.group(1)), int(r.group(2))
        waited = 0
        for j in range(i + 1, min(i + 1 + WINDOW, len(lines))):
            n = lines[j].strip()
            if not n:
                continue
            if n.startswith("s_nop"):
                waited += 1 + int(n.split()[1]) if len(n.split()) > 1 else 1
                continue
            if n.startswith("s_") or n.startswith("buffer_store") or n.startswith("global_store") or n.startswith("ds_write"):
                continue
            if not n.startswith("v_") and not n.startswith("ds_read") and not n.startswith("buffer_load") and not n.startswith("global_load"):
                continue
            dst = n.split(None, 1)[1].split(",")[0].strip() if " " in n else ""
            regs = set()
            r2 = vrange.match(dst)
            if r2:
                regs = set(range(int(r2.group(1)), int(r2.group(2)) + 1))
            elif re.match(r"^v\d+$", dst):
                regs = {int(dst[1:])}
            if regs & set(range(lo, hi + 1)) and waited == 0 and n.startswith("v_") and (sgpr_soffset or j == i + 1):
                hits += 1
                total += 1
                if hits <= 3:
                    print(f"{unit}: {fn[:90]}\n    {l.strip()}\n    {n}   <- data register rewritten {j - i} instruction(s) later, soffset {'SGPR' if sgpr_soffset else 'imm'}")
    print(f"{unit}: {hits} site(s)")
print("total", total)
