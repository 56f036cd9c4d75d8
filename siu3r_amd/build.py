"""Build libsiu3r_hip.so (HIP kernels + C ABI) for gfx950, in-tree.

`python -m siu3r_amd.build` or `siu3r_amd.build.build()`.  hipcc cross-compiles without a GPU.
The .so lands next to this file so that it travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsiu3r_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]
# per-file extras.  gemm_pp*.hip (the ping-pong GEMM: host side + one unit per tile shape and operand mode): its epilogue walks the accumulator blocks in fully unrolled loops whose bodies are large; above the default
# pragma-unroll size limit the loops stay rolled, the block index becomes a run-time value and the accumulators move to scratch memory
_PP = ["-mllvm", "-pragma-unroll-threshold=1000000"]
# -fno-slp-vectorize for every translation unit: on gfx950 a packed fp32 instruction whose LOW result selects the HIGH half of its second
# source (op_sel:[x,1]: what the SLP vectoriser emits for horizontal / crossed pairings, e.g. v_pk_add_f32 v, v, v op_sel:[0,1]
# op_sel_hi:[1,0]) returns a wrong low half in 0.1-0.3 % of its executions while a bf16 MFMA of ANOTHER wave runs on the same SIMD
# (tools/probes/pk_hazard/xwave2.hip; the round-4 panoptic label flake).  Without the vectoriser no kernel of the library contains that form
# (tests/test_abi.py walks the shipped code objects).
_NOSLP = ["-fno-slp-vectorize"]
FLAGS = FLAGS + _NOSLP
EXTRA_FLAGS = {"gemm_pp.hip": _PP, **{f"gemm_pp_t{n}{m}.hip": _PP for n in (1, 2, 3) for m in "xb"}}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path):
    h = hashlib.sha1()
    for f in [path, *sorted(os.path.join(CSRC, h_) for h_ in os.listdir(CSRC) if h_.endswith(".h")), os.path.join(HERE, "..", "include", "siu3r_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + EXTRA_FLAGS.get(os.path.basename(path), [])).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dg = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg and os.path.exists(obj.replace(".o", ".resources.txt")):
        return obj
    cmd = ["hipcc", *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    with open(obj.replace(".o", ".resources.txt"), "w") as fh:  # per-kernel VGPR / LDS / occupancy remarks (tests/test_abi.py reads them)
        fh.write(r.stderr)
    with open(stamp, "w") as fh:
        fh.write(dg)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, _sources()))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return OUT


def kernel_resources(src: str) -> dict:
    """{mangled kernel name: {"VGPRs": n, "AGPRs": n, "ScratchSize [bytes/lane]": n, "Occupancy [waves/SIMD]": n, "LDS Size [bytes/block]": n, ...}}
    from the compiler's resource remarks of one source file (written by the last compile of that file)."""
    import re

    path = os.path.join(OBJ, src.replace(".hip", ".resources.txt"))
    if not os.path.exists(path):
        build()
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
