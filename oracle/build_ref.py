"""oracle/build_ref.py -- TEST INFRASTRUCTURE ONLY.

Builds the reference's OWN `index_max` CPU entry points, from the source where it
lies under /root/reference, into oracle/_ref/index_max.so.  Nothing is copied into
this repository and no stand-in source is written:

  * the one translation unit models/index_max_ext/index_max.cpp is compiled
    unmodified with g++ against the torch headers of this image;
  * the two CUDA entry points that file only *declares* (index_max.cpp:124-130;
    their definitions live in the .cu, which needs nvcc) are left as undefined
    symbols of the shared object.  A shared object may carry undefined symbols;
    `load_ref_index_max()` below imports it with RTLD_LAZY so they are never bound
    unless someone calls forward_cuda*, which the oracle never does.

ball_query has no CPU source in the reference (ball_query.cpp:23-31 is a stub,
the only implementation is CUDA) -> unbuildable here; see usip_oracle.c.

The output directory oracle/_ref/ is git-ignored (kept out of history) but is NOT
in .gpurunignore, so the built .so travels to the GPU box like our own .so files.
/root/reference itself does not exist there; nothing at run time reads it.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/models/index_max_ext/index_max.cpp"
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "index_max.so")


def build_ref(force: bool = False) -> str:
    """Compile the reference index_max.cpp -> oracle/_ref/index_max.so. Returns the path
    (or '' when /root/reference is absent and no prebuilt .so exists)."""
    if os.path.exists(OUT_SO) and not force:
        if not os.path.exists(REF_SRC) or os.path.getmtime(OUT_SO) >= os.path.getmtime(REF_SRC):
            return OUT_SO
    if not os.path.exists(REF_SRC):
        return ""
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT_DIR, exist_ok=True)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w",
           "-DTORCH_EXTENSION_NAME=index_max", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in ce.include_paths() + [sysconfig.get_paths()["include"]]:
        cmd += ["-isystem", inc]
    cmd += [REF_SRC, "-o", OUT_SO, "-L" + torch_lib, "-Wl,-rpath," + torch_lib,
            "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python", "-pthread"]
    subprocess.check_call(cmd)
    return OUT_SO


def load_ref_index_max():
    """Import oracle/_ref/index_max.so as a module named `index_max` (the name its
    PYBIND11_MODULE hard-codes, index_max.cpp:154) without registering it in
    sys.modules. Returns None if the .so is not available."""
    so = build_ref()
    if not so:
        return None
    import importlib.machinery
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        loader = importlib.machinery.ExtensionFileLoader("index_max", so)
        spec = importlib.util.spec_from_loader("index_max", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod


if __name__ == "__main__":
    print(build_ref(force="--force" in sys.argv) or "reference source not present; nothing built")
