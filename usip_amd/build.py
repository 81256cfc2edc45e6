"""Build libusip_hip.so (hand-written HIP for gfx950 + the C-ABI) in-tree with hipcc.

    python -m usip_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the working tree.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libusip_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc's SLP vectoriser pairs independent fp32 operations of neighbouring channels into packed
# instructions (v_pk_fma_f32 ...) and, when the register allocator has the pair in the other order, selects the halves
# with op_sel.  On gfx950 (ROCm 7.2) such an instruction -- `v_pk_fma_f32 v[78:79], v[192:193], v[168:169], v[200:201]
# op_sel:[0,1,0] op_sel_hi:[1,0,1]` in the 64 -> 128 fused layer backward -- returned c3 instead of c2 * y + c3 in its
# LOW half for lanes 48-63, in 1-4 of 512 workgroups per launch, only with two waves on a SIMD (found by
# dumping the LDS image of the wrong tiles).  Packed fp32 written out by hand (split_common.h, the distance loops, the
# BatchNorm-backward prologue of the split GEMMs) never reads a high register into a low half and stays; everything
# else is left scalar (tests/test_kernel_isa.py checks every translation unit).  Measured cost: none (the kernels are not VALU-bound).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         # what usip_version() reports and usip_amd/_lib.py insists on: the three flags results depend on
         '-DUSIP_BUILD_FLAGS="no-contract,no-fast-math,no-slp"']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")) + [os.path.abspath(__file__)]   # (the flags)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + \
            glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h")) + [os.path.abspath(__file__)]
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
