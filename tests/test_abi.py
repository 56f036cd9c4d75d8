"""CPU tests: the C-ABI shared library loads and exports every symbol include/siu3r_hip.h declares (no compute calls
without a GPU), the ctypes structs match the C layouts, and the product fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "siu3r_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(siu3r_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from siu3r_amd import _lib

    l = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 25, syms
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/siu3r_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= set(syms) | {"siu3r_last_error", "siu3r_abi_version"}
    assert l.siu3r_abi_version() == _lib.ABI_VERSION == 10


def test_struct_layouts_match_c():
    """sizeof of the ctypes mirrors == sizeof of the C structs (compiled with the host compiler)."""
    from siu3r_amd import _lib

    src = '#include <stdio.h>\n#include "siu3r_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(siu3r_gemm_params), sizeof(siu3r_attn_params), sizeof(siu3r_raster_cam));return 0;}\n'
    exe = "/tmp/siu3r_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    sizes = list(map(int, subprocess.run([exe], capture_output=True, check=True).stdout.split()))
    assert sizes == [C.sizeof(_lib.GemmParams), C.sizeof(_lib.AttnParams), C.sizeof(_lib.RasterCam)], sizes


def test_oracle_raster_struct_matches():
    from oracle import raster_oracle as RO
    from siu3r_amd import _lib

    assert RO.lib().raster_ref_struct_size() == C.sizeof(_lib.RasterCam)


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing on the host."""
    from siu3r_amd import ops

    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-6)
    with pytest.raises(RuntimeError):
        ops.rope_2d(torch.zeros(1, 2, 2, 8), torch.zeros(1, 2, 2, dtype=torch.int64), 100.0, 1.0)
    if not torch.cuda.is_available():
        from siu3r_amd.model import SIU3RModel

        with pytest.raises(RuntimeError):
            SIU3RModel({}, image_size=(64, 64))


def test_rows_layout_host_logic():
    from siu3r_amd.ops import _rows_layout

    x = torch.zeros(2, 9, 16)
    assert _rows_layout(x) == (1, 18, 16, 0)
    assert _rows_layout(x[:, :-1]) == (2, 8, 16, 144)
    assert _rows_layout(torch.zeros(3, 4, 5, 8)[:, :, :, :]) == (1, 60, 8, 0)
    assert _rows_layout(torch.zeros(2, 2, 9, 16)[:, 0]) == (2, 9, 16, 288)


def test_gemm_kernels_keep_two_workgroups_per_cu():
    """The 128 x 64 LDS-DMA GEMM kernels (eight waves, KSPL = 2) are tuned for TWO workgroups per CU: that needs <= 128 VGPRs (4 waves
    per SIMD), no scratch, and two 72 KiB rings inside the 160 KiB LDS.  An epilogue edit that pushes a variant over the line halves
    its occupancy silently (seen this round: 110 -> 152 VGPRs cost 25 % of the step), so the compiler's own resource remarks are
    checked at build time, without a GPU."""
    from siu3r_amd import build as B

    B.build()
    res = B.kernel_resources("gemm_dma.hip")
    import re

    checked = 0
    for name, r in res.items():
        m = re.search(r"gemm_dma_kernelILi(\d)ELi(\d)ELb(\d)ELi(\d)ELb(\d)E", name)
        x = re.search(r"gemm_dma_x3_kernelILi(\d)ELb(\d)ELi(\d)ELb(\d)E", name)
        if m and m.group(1) == "1" and m.group(4) == "2" or x and x.group(3) == "2":
            checked += 1
            assert r["VGPRs"] + r.get("AGPRs", 0) <= 128, (name, r)
            assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0, (name, r)
            assert r["Occupancy [waves/SIMD]"] >= 4, (name, r)
            assert 2 * r["LDS Size [bytes/block]"] <= 160 * 1024, (name, r)
    assert checked >= 10, sorted(res)


def test_ping_pong_and_pipelined_kernels_fit_one_workgroup_per_cu():
    """gemm_pp.hip / attention_pipe.hip are 8-wave kernels tuned for ONE workgroup per CU (two waves per SIMD): at most 256 registers,
    LDS inside 160 KiB, and no scratch in the main loop (the row passes of the epilogue -- a general one and two fast ones share the
    kernel -- may spill a few registers next to the 64 / 128 live accumulators of the 256-row tiles: bounded here so that a main-loop
    spill, thousands of bytes, cannot slip in; the compiled code has every scratch access behind the last MFMA)."""
    import re

    from siu3r_amd import build as B

    B.build()
    checked = 0
    pp = {}
    for unit in ("gemm_pp_t1x.hip", "gemm_pp_t1b.hip", "gemm_pp_t2x.hip", "gemm_pp_t2b.hip", "gemm_pp_t3x.hip", "gemm_pp_t3b.hip"):
        pp.update(B.kernel_resources(unit))
    for name, r in pp.items():
        m = re.search(r"gemm_pp_kernelILb(\d)ELi(\d)ELi(\d)", name)
        if not m:
            continue
        checked += 1
        mi, nj = int(m.group(2)), int(m.group(3))
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 256 and r["Occupancy [waves/SIMD]"] >= 2, (name, r)
        assert r["LDS Size [bytes/block]"] <= 160 * 1024, (name, r)
        assert r["ScratchSize [bytes/lane]"] <= (384 if (mi, nj) == (2, 4) else 128 if mi == 2 else 0), (name, r)
    assert checked >= 12, checked
    res = B.kernel_resources("attention_pipe.hip")
    assert len(res) == 3  # bf16, bf16x3, bf16x3 with pre-split K / V
    for name, r in res.items():
        assert r["VGPRs"] + r.get("AGPRs", 0) <= 256 and r["ScratchSize [bytes/lane]"] == 0 and r["LDS Size [bytes/block]"] <= 160 * 1024, (name, r)


def _device_disassembly(unit):
    """gfx950 disassembly of one translation unit's device code: .hip_fatbin of the object file -> unbundle -> llvm-objdump"""
    import subprocess
    import tempfile

    from siu3r_amd import build as B

    B.build()
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(B.OBJ, unit.replace(".hip", ".o"))
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "unused.o")], check=True)
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        return subprocess.run([f"{llvm}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_ping_pong_main_loops_are_spill_free():
    """The 256-row ping-pong variants report scratch (264-316 B per lane for 256 x 256, test above).  That is only acceptable behind the
    main loop: in the compiled code object of every variant, no scratch access may lie inside a loop that holds MFMAs (the prologue
    may park a value, the row passes of the epilogue may spill next to the live accumulators; a K loop that spills would cost every
    step).  Checked on the shipped objects: every backward branch spanning MFMAs is a loop, all MFMAs must be inside one."""
    checked = spilling = 0
    for unit in ("gemm_pp_t1x.hip", "gemm_pp_t1b.hip", "gemm_pp_t2x.hip", "gemm_pp_t2b.hip", "gemm_pp_t3x.hip", "gemm_pp_t3b.hip"):
        asm = _device_disassembly(unit)
        funcs = re.split(r"\n([0-9a-f]+) <([^>]+)>:\n", asm)
        for base, name, body in zip(funcs[1::3], funcs[2::3], funcs[3::3]):
            if "gemm_pp_kernel" not in name:
                continue
            base = int(base, 16)
            ins = []  # (address, text)
            for l in body.split("\n"):
                m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
                if m:
                    ins.append((int(m.group(1), 16), l))
            mfma = [a for a, l in ins if "v_mfma" in l]
            scratch = [a for a, l in ins if re.search(r"\bscratch_(load|store)", l)]
            assert mfma, name
            # loops that hold MFMAs: backward branches (target <= branch address) spanning at least one MFMA
            loops = []
            for a, l in ins:
                m = re.search(r"s_c?branch\w*\s.*<[^>]*\+0x([0-9a-f]+)>", l)
                if m and base + int(m.group(1), 16) <= a and any(base + int(m.group(1), 16) <= x <= a for x in mfma):
                    loops.append((base + int(m.group(1), 16), a))
            assert loops, f"{name}: no MFMA loop found in the disassembly"
            # innermost ones only (an outer loop over tiles / slices holds the epilogue too); back-edges of one header are one loop
            heads = {lo: max(h for l_, h in loops if l_ == lo) for lo, _ in loops}
            loops = [(lo, hi) for lo, hi in heads.items() if not any(lo < lo2 and hi2 <= hi or lo <= lo2 and hi2 < hi for lo2, hi2 in heads.items())]
            assert all(any(lo <= x <= hi for lo, hi in loops) for x in mfma), f"{name}: MFMAs outside the K loop"
            checked += 1
            spilling += 1 if scratch else 0
            bad = [hex(x) for x in scratch if any(lo <= x <= hi for lo, hi in loops)]
            assert not bad, f"{unit}: {name}: scratch accesses inside the MFMA loop at {bad[:4]} (loops {[(hex(lo), hex(hi)) for lo, hi in loops]})"
    assert checked >= 12, checked
    print(f"{checked} ping-pong variants disassembled, {spilling} use scratch -- none of it inside an MFMA loop")


def _plan(**kw):
    """siu3r_gemm_plan is a host function: it can be asked about a problem without a GPU (pointers are only tested for null)"""
    from siu3r_amd import _lib

    p = _lib.GemmParams()
    p.a, p.w_hi, p.c = 64, 64, 64  # non-null, never dereferenced by the plan
    p.a_dtype, p.c_dtype = _lib.F32, _lib.F32
    p.w_x3 = 64
    p.sk_ws, p.sk_cnt, p.sk_ws_floats, p.sk_cnt_n = 64, 64, 1 << 24, 8192
    for k, v in kw.items():
        setattr(p, k, v)
    p.kpad = (p.k + 63) // 64 * 64
    p.lda = p.k
    pl = _lib.GemmPlan()
    _lib.check(_lib.lib().siu3r_gemm_plan(C.byref(p), C.byref(pl)))
    return pl


def test_gemm_plan_consults_the_measured_table_then_the_model():
    """Decision order of the dispatch (ABI 4): caller's tile_cfg > process default > csrc/gemm_tuned.h (exact problem)
    > cost model.  Every dense single-batch entry of the committed table must come back from siu3r_gemm_plan as listed (tile, and
    slice count when the entry names one); a forced tile overrides it; a problem that is not listed is the model's with or without
    the table; split-K is planned only with a workspace, and never beyond it."""
    from siu3r_amd import _lib

    lib = _lib.lib()
    text = open(os.path.join(ROOT, "siu3r_amd", "csrc", "gemm_tuned.h")).read()
    entries = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"\{(-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+)\},", text)]
    dense = [e for e in entries if e[0] > 0 and e[3] == 1 and e[4] == 0 and e[5] == 0 and e[9] == 0]
    assert len(dense) >= 3, "the table is expected to hold dense single-batch problems"
    names = {-1: "gemm_dma", 1: "gemm_pp_kernel", 2: "gemm_pp_kernel", 3: "gemm_pp_kernel"}
    try:
        for (m, n, k, _z, _am, _om, _kh, _st, x3, _ln, cfg, sk) in dense:
            kw = dict(m=m, n=n, k=k) if x3 else dict(m=m, n=n, k=k, a_dtype=_lib.BF16, c_dtype=_lib.BF16, w_x3=0)
            pl = _plan(**kw)
            assert pl.tile_cfg == cfg and names[cfg] in pl.kernel.decode(), (m, n, k, x3, cfg, pl.tile_cfg, pl.kernel)
            if sk > 0:
                assert pl.splitk == sk, (m, n, k, sk, pl.splitk)
        listed = dict(m=dense[0][0], n=dense[0][1], k=dense[0][2]) if dense[0][8] else None
        if listed:
            other = 3 if dense[0][10] != 3 else 2
            forced = _plan(tile_cfg=other, **listed)
            assert forced.tile_cfg == other
        a = _plan(m=2050, n=1024, k=4032)
        _lib.check(lib.siu3r_gemm_tune(3, 1))
        b = _plan(m=2050, n=1024, k=4032)
        assert (a.tile_cfg, a.splitk, a.skinny_rows) == (b.tile_cfg, b.splitk, b.skinny_rows)
        none = _plan(sk_ws=0, sk_cnt=0, m=200, n=136, k=4608)
        assert none.splitk == 1
        small = _plan(sk_ws_floats=1024, m=200, n=136, k=4608)
        assert small.splitk == 1 or small.ws_floats <= 1024
        with_ws = _plan(m=200, n=136, k=4608)
        assert with_ws.splitk >= 2, "few tiles and a long K: the model splits when it may"
    finally:
        _lib.check(lib.siu3r_gemm_tune(3, 0))


def test_gemm_plan_validates_like_the_launch():
    """siu3r_gemm_plan runs the same parameter validation as siu3r_gemm: a malformed conv block (oh * ow == 0: the plan divides by it)
    comes back as an error string, not as SIGFPE; so do a bad kpad and a null operand."""
    with pytest.raises(RuntimeError, match="conv geometry"):
        _plan(m=4096, n=256, k=2304, a_mode=1, ih=64, iw=64, cin=256, kh=3, kw=3, stride=1, pad=1, oh=0, ow=0)
    with pytest.raises(RuntimeError, match="kh\\*kw\\*cin"):
        _plan(m=4096, n=256, k=2304, a_mode=1, ih=64, iw=64, cin=128, kh=3, kw=3, stride=1, pad=1, oh=64, ow=64)
    with pytest.raises(RuntimeError, match="null operand"):
        _plan(m=128, n=128, k=128, a=0)
    ok = _plan(m=4096, n=256, k=2304, a_mode=1, ih=64, iw=64, cin=256, kh=3, kw=3, stride=1, pad=1, oh=64, ow=64)
    assert ok.bm > 0 and ok.kernel


def test_batch_strided_views_host_logic():
    from siu3r_amd.ops import _batch_strided

    seq = torch.zeros(2, 50, 8)
    lvl = seq[:, 10:42].view(2, 4, 8, 8)  # a level cut out of the middle of each item's sequence
    assert not lvl.is_contiguous() and _batch_strided(lvl, 4 * 8 * 8) == 50 * 8
    assert _batch_strided(torch.zeros(2, 4, 8, 8), 256) == 256
    assert _batch_strided(seq[:1, :32].view(1, 4, 8, 8), 256) == 256  # one item: its stride does not matter
    with pytest.raises(AssertionError):
        _batch_strided(torch.zeros(2, 4, 8, 16)[..., :8], 256)  # rows that are not dense


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the ROCm LLVM tools")
@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_volume_sampling_kernels_hold_no_packed_fp32():
    """Round 6 traced the round-4 panoptic label flake to ONE instruction of the compiled argmax loop, `v_pk_add_f32 … op_sel:[0,1]
    op_sel_hi:[1,0]`: a packed fp32 form that miscomputes beside another wave's bf16 MFMAs on gfx950 (csrc/postprocess.hip, sample256;
    tools/pk_hazard_probe.py reproduces it with the old code object and one-edit variants of its assembly, tools/probes/pk_hazard/xwave2.hip
    with the instruction alone).  The two kernels that sample the probability volume must not contain packed-fp32 arithmetic at all
    (register pins in the source on top of the library-wide -fno-slp-vectorize): checked on the shipped object."""
    asm = _device_disassembly("postprocess.hip")
    funcs = re.split(r"\n([0-9a-f]+) <([^>]+)>:\n", asm)
    seen = 0
    for name, body in zip(funcs[2::3], funcs[3::3]):
        if "pp_argmax_kernel" in name or "pp_qcl_kernel" in name:
            seen += 1
            packed = [l.strip() for l in body.split("\n") if re.search(r"\bv_pk_\w+_f32\b", l)]
            assert not packed, f"{name}: {packed[:4]}"
            assert "global_load_dword" in body and " sc0 sc1" not in body, name  # plain loads: nothing was stale
    assert seen == 2, seen


def test_no_packed_fp32_selects_the_high_half_of_src1():
    """Round 6: on gfx950 `v_pk_add_f32 / v_pk_mul_f32 … op_sel:[x,1]` (the LOW result reads the HIGH half of the second source) returns a
    wrong low half in 0.1-0.3 % of its executions while a bf16 MFMA of another wave runs on the same SIMD (tools/probes/pk_hazard/xwave2.hip;
    the round-4 panoptic label flake).  The SLP vectoriser emits the form for crossed pairings; the library
    is built with -fno-slp-vectorize and no shipped code object may contain it (tools/scan_pk_opsel_hazard.py walks every translation unit)."""
    from siu3r_amd import build as B

    assert "-fno-slp-vectorize" in B.FLAGS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_pk_opsel_hazard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "total 0", r.stdout[-3000:]


def test_no_store_data_register_is_rewritten_without_a_wait_state():
    """Round 4 found a 16-byte buffer store (SGPR soffset, hence no compiler-inserted s_nop) whose data register the NEXT instruction
    rewrote: on gfx950 the store read the new value in lanes 12..15 of every row of 16 (448 wrong words of 4 M in the pre-split planes of
    the 256 x 256 tile).  The round-3 build had 210 such sites in the ping-pong row pass; none may come back (tools/scan_store_hazard.py
    walks the disassembly of every translation unit)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_store_hazard.py"), "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().splitlines()[-1] == "total 0", r.stdout[-3000:]
