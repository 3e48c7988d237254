"""A/B builds of the whole library with extra compiler flags: python scripts/ab_build.py NAME [-DFLAG=..] ...  -> ab_tmp/lib_NAME.so
(run on the GPU box as: cp ab_tmp/lib_NAME.so nlp-cube_amd/cube_amd/lib/libcubehip.so && python bench.py ...)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "ab_tmp"), exist_ok=True)
csrc = os.path.join(ROOT, "nlp-cube_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-I" + os.path.join(ROOT, "include")]
                      + flags + ["-o", os.path.join(ROOT, "ab_tmp/lib_%s.so" % name)] + [os.path.join(csrc, f) for f in g.SOURCES])
print("built", name, flags)
