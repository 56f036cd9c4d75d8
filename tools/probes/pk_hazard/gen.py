"""Code objects for tools/pk_hazard_probe.py: the compiler's own code of the round-3/4 panoptic argmax kernel (ppa.hip, hipcc -O3 for gfx950:
ppa_e0) and one-edit variants of its ASSEMBLY, each assembled into its own code object under _build/.  The edits are text substitutions on the
inner loop; an assert fails if the compiler's output no longer contains the expected lines (another ROCm version: regenerate by hand).
python tools/probes/pk_hazard/gen.py"""
import os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
os.makedirs(OUT, exist_ok=True)
LL = '/opt/rocm/lib/llvm/bin/'
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", os.path.join(OUT, "ppa_e0.s"), os.path.join(HERE, "ppa.hip")],
                      stderr=subprocess.DEVNULL)
os.chdir(OUT)
base = open('ppa_e0.s').read()
def variant(n, edits, bump=None):
    s = base
    for a, b in edits:
        assert s.count(a) >= 1, (n, a)
        s = s.replace(a, b)
    if bump:
        s = s.replace('.amdhsa_next_free_vgpr 30', f'.amdhsa_next_free_vgpr {bump}').replace('.vgpr_count:     30', f'.vgpr_count:     {bump}').replace('.set ppa_e0.num_vgpr, 30', f'.set ppa_e0.num_vgpr, {bump}')
    s = s.replace('ppa_e0', f'ppa_e{n}')
    open(f'ppa_e{n}.s', 'w').write(s)
    subprocess.check_call([LL + 'clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', f'ppa_e{n}.s', '-o', f'ppa_e{n}.o'])
    subprocess.check_call([LL + 'ld.lld', '-shared', f'ppa_e{n}.o', '-o', f'ppa_e{n}.hsaco'])
W1 = '\ts_waitcnt vmcnt(1)\n\tv_pk_mul_f32 v[18:19], v[14:15], v[28:29]\n'
W0 = '\ts_waitcnt vmcnt(0)\n\tv_pk_mul_f32 v[20:21], v[14:15], v[26:27]\n'
TAIL = '\ts_nop 0\n\tv_pk_add_f32 v[18:19], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 v[18:19], v[4:5], v[18:19]\n\ts_nop 0\n\tv_add_f32_e32 v18, v19, v18\n'
assert W1 in base and W0 in base and TAIL in base
variant(0, [])
variant(1, [(W1, W1.replace('vmcnt(1)', 'vmcnt(0)'))])                                   # full wait before the first packed multiply
variant(2, [(TAIL, TAIL.replace('s_nop 0', 's_nop 7'))])                                  # 8 wait states between the dependent packed operations
variant(3, [(W1, '\ts_waitcnt vmcnt(1)\n\ts_nop 7\n\tv_pk_mul_f32 v[18:19], v[14:15], v[28:29]\n'), (W0, '\ts_waitcnt vmcnt(0)\n\ts_nop 7\n\tv_pk_mul_f32 v[20:21], v[14:15], v[26:27]\n')])  # idle cycles behind the waits
# 4: products into fresh registers (not the finished loads' address registers)
variant(4, [(W1, W1.replace('v[18:19],', 'v[30:31],')), (W0, W0.replace('v[20:21],', 'v[32:33],')), (TAIL, TAIL.replace('v_pk_add_f32 v[18:19], v[18:19], v[20:21]', 'v_pk_add_f32 v[18:19], v[30:31], v[32:33]'))], bump=34)
# 5: scalar tail: products packed as before, the rest scalar.  pk_add op_sel:[0,1] op_sel_hi:[1,0]: lo = a.lo + b.hi, hi = a.hi + b.lo; then pk_mul by v[4:5]; then sum
variant(5, [(TAIL, '\ts_nop 0\n\tv_add_f32_e32 v18, v18, v21\n\tv_add_f32_e32 v19, v19, v20\n\ts_nop 0\n\tv_mul_f32_e32 v18, v4, v18\n\tv_mul_f32_e32 v19, v5, v19\n\ts_nop 0\n\tv_add_f32_e32 v18, v19, v18\n')])
# 6: scalar products, packed tail
variant(6, [(W1, '\ts_waitcnt vmcnt(1)\n\tv_mul_f32_e32 v18, v14, v28\n\tv_mul_f32_e32 v19, v15, v29\n'), (W0, '\ts_waitcnt vmcnt(0)\n\tv_mul_f32_e32 v20, v14, v26\n\tv_mul_f32_e32 v21, v15, v27\n')])
# 7: loads issued in register order (v26, v27, v28, v29) -- pairs filled by consecutive loads -- waits adjusted: first product needs loads 3, 4 -> vmcnt(0)... keep semantic: first product pair v[28:29] = loads 3,4
variant(7, [('\tglobal_load_dword v26, v[18:19], off\n\tglobal_load_dword v29, v[20:21], off\n\tglobal_load_dword v28, v[22:23], off\n\tglobal_load_dword v27, v[24:25], off\n',
             '\tglobal_load_dword v28, v[22:23], off\n\tglobal_load_dword v29, v[20:21], off\n\tglobal_load_dword v26, v[18:19], off\n\tglobal_load_dword v27, v[24:25], off\n'),
            (W1, W1.replace('vmcnt(1)', 'vmcnt(2)'))])   # pair (v28, v29) = loads 1, 2 -> vmcnt(2); pair (v26, v27) = loads 3, 4 -> vmcnt(0)
print("ok")
# ---- second round
PADD = 'v_pk_add_f32 v[18:19], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]'
PMUL2 = 'v_pk_mul_f32 v[18:19], v[4:5], v[18:19]'
# 8: sources v[18:21] as compiled, the sum NOT in place (fresh destination)
variant(8, [(PADD, 'v_pk_add_f32 v[30:31], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]'), (PMUL2, 'v_pk_mul_f32 v[18:19], v[4:5], v[30:31]')], bump=34)
# 9: products in fresh registers (as e4) but the sum IN PLACE on them
variant(9, [(W1, W1.replace('v[18:19],', 'v[30:31],')), (W0, W0.replace('v[20:21],', 'v[32:33],')), (PADD, 'v_pk_add_f32 v[30:31], v[30:31], v[32:33] op_sel:[0,1] op_sel_hi:[1,0]'), (PMUL2, 'v_pk_mul_f32 v[18:19], v[4:5], v[30:31]')], bump=34)
# 10: only the packed add replaced by two scalar adds, packed multiply behind it kept
variant(10, [('\t' + PADD + '\n', '\tv_add_f32_e32 v18, v18, v21\n\tv_add_f32_e32 v19, v19, v20\n')])
# 11: the four gathers take their addresses from v[30:37] (computed there); v[18:21] are never address operands, the products still land in v[18:21]
ADDR = '\tv_lshl_add_u64 v[18:19], v[6:7], 0, s[20:21]\n\tv_lshl_add_u64 v[20:21], v[8:9], 0, s[20:21]\n\tv_lshl_add_u64 v[22:23], v[10:11], 0, s[20:21]\n\tv_lshl_add_u64 v[24:25], v[12:13], 0, s[20:21]\n\tglobal_load_dword v26, v[18:19], off\n\tglobal_load_dword v29, v[20:21], off\n\tglobal_load_dword v28, v[22:23], off\n\tglobal_load_dword v27, v[24:25], off\n'
assert ADDR in base
variant(11, [(ADDR, ADDR.replace('v[18:19]', 'v[30:31]').replace('v[20:21]', 'v[32:33]').replace('v[22:23]', 'v[34:35]').replace('v[24:25]', 'v[36:37]'))], bump=38)
print("ok2")

# the neighbour kernel of standalone.py
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", "burn.s", os.path.join(HERE, "burn.hip")], stderr=subprocess.DEVNULL)
subprocess.check_call([LL + 'clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', 'burn.s', '-o', 'burn.o'])
subprocess.check_call([LL + 'ld.lld', '-shared', 'burn.o', '-o', 'burn.hsaco'])
print("burn ok")

# the shipped form of the kernel (csrc/postprocess.hip: plain loads, pinned scalar arithmetic, -fno-slp-vectorize) as a stand-alone code object
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", "ppa_fix.s", os.path.join(HERE, "ppa_fix.hip")], stderr=subprocess.DEVNULL)
assert "v_pk_" not in open("ppa_fix.s").read().split("ppa_fix:")[1].split("s_endpgm")[0]
subprocess.check_call([LL + 'clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', 'ppa_fix.s', '-o', 'ppa_fix.o'])
subprocess.check_call([LL + 'ld.lld', '-shared', 'ppa_fix.o', '-o', 'ppa_fix.hsaco'])
print("fix ok")
