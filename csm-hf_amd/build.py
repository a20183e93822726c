"""Build libcsm_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain C-ABI
shared object (include/csm_hip.h).  `python -m csm_hf_amd.build` or `build_library()`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libcsm_hip.so")
UNITS = ["gemv", "gemv_w0_k1", "gemv_w0_k2", "gemv_w0_k4", "gemv_w1_k1", "gemv_w1_k2", "gemv_w1_k4", "gemv_w2_k1", "gemv_w2_k2", "gemv_w2_k4", "gemm16", "gemm32", "gemm128", "gemm_mx", "train", "attn_prefill", "launchers", "engine", "mimi"]
# gemm16 / gemm32: no implicit multiply-add contraction -- the batched decode launches of every width (16 / 32 / 64 / 128 rows,
# several template instantiations of the same epilogues) must round alike, bit for bit (tests: logits of wide launches against
# 16-row launches); the few fused operations these epilogues want are spelled out (__fmaf_rn)
UNIT_FLAGS = {"attn_prefill": ["-mllvm", "--amdgpu-mfma-vgpr-form"], "gemm16": ["-ffp-contract=off"], "gemm32": ["-ffp-contract=off"], "gemm128": ["-ffp-contract=off", "-mllvm", "--amdgpu-mfma-vgpr-form"]}
# -amdgpu-kernarg-preload-count: leading scalar kernel parameters (up to 14 dwords) arrive in SGPRs, initialised by the dispatcher, instead of
# being s_loaded by every wave at its first instruction (gemv.h GEMV_HOT_PARAMS; round 6)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _sources_mtime() -> float:
    m = os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "csm_hip.h"))
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h", ".inc")):
            m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    return m


def sources_sha256() -> str:
    """sha256 over the kernel sources and the C header (sorted file names + contents): identifies the BUILD INPUT of the library.  Measurement
    records (profiles/hbm_traffic.json, profiles/launch_kinds_b*.json) are stamped with it and bench.py marks them stale when it differs --
    a rebuilt library of the same sources is the same build, whatever its bytes are."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".inc"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "csm_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False, defines=(), out: str = None) -> str:
    """`defines` / `out`: A/B variants of the same ABI (e.g. ("CSM_NO_PROBE",) -> libcsm_hip_noprobe.so), selected at
    run time with the CSM_HIP_LIB environment variable."""
    global OBJ
    lib_out = out or LIB
    if not force and not defines and os.path.exists(LIB) and os.path.getmtime(LIB) >= _sources_mtime():
        return LIB
    hipcc = _hipcc()
    obj_dir = OBJ if not defines else OBJ + "_" + "_".join(defines)
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(u):
        cmd = [hipcc, *FLAGS, *UNIT_FLAGS.get(u, []), *[f"-D{d}=1" for d in defines], "-c", os.path.join(CSRC, u + ".hip"), "-o", os.path.join(obj_dir, u + ".o")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {u}.hip:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[build] {u}.o", file=sys.stderr)

    with ThreadPoolExecutor(min(len(UNITS), os.cpu_count() or 4)) as ex:
        list(ex.map(compile_one, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[os.path.join(obj_dir, u + ".o") for u in UNITS], "-o", lib_out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return lib_out


if __name__ == "__main__":
    defs = tuple(a[2:] for a in sys.argv[1:] if a.startswith("-D"))
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build_library(force="--force" in sys.argv or bool(defs), verbose=True, defines=defs, out=outs[0] if outs else None))
