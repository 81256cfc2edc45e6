"""Build a VARIANT of libusip_hip.so for same-box A/B runs or measurement builds (never the product):
    python tools/build_variant.py <tag> <file.hip>[,<file.hip>...] [-DMACRO ...]
recompiles the named translation units with the extra flags and links them with the product's other objects
(usip_amd/build/*.o, built by `python -m usip_amd.build`) into tools/variants/libusip_hip_<tag>.so; run a tool against it
with USIP_LIB=tools/variants/libusip_hip_<tag>.so (usip_amd/_lib.py)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usip_amd import build as B  # noqa: E402

tag, files = sys.argv[1], sys.argv[2].split(",")
extra = sys.argv[3:]
B.build()
objdir = os.path.join(ROOT, "tools", "variants", tag)
os.makedirs(objdir, exist_ok=True)
objs = []
for src in B.sources():
    base = os.path.basename(src)
    if base in files:
        obj = os.path.join(objdir, base + ".o")
        subprocess.check_call([B.HIPCC] + B.FLAGS + extra + ["-x", "hip", "-c", src, "-o", obj], stderr=subprocess.DEVNULL)
    else:
        obj = os.path.join(B.HERE, "build", base + ".o")
    objs.append(obj)
out = os.path.join(ROOT, "tools", "variants", "libusip_hip_%s.so" % tag)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"])
print(out)
