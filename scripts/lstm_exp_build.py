"""Builds ab_tmp/lib_lstm_<name>.so: csrc/lstm_i8.hip (+ rowops.hip for the error plumbing) with a timing-experiment variant.
    python scripts/lstm_exp_build.py BASE|GXREG4|NOUNPACK|NOGX|SCREG   ->  python scripts/bench_lstm.py 256 33 200 ab_tmp/lib_lstm_<name>.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nlp-cube_amd/csrc")
name = sys.argv[1]
s = open(os.path.join(CSRC, "lstm_i8.hip")).read()


def rep(old, new, count=1):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new, count)


if name == "GXREG4":      # nibble slices with the input projections in registers: isolates the register path from the unpack
    rep("constexpr bool gxl = WLB == 4 && C::lds(true) <= 160 * 1024;", "constexpr bool gxl = false;")
elif name == "NOUNPACK":  # 6-bit slices read but not unpacked (wrong results): the cost of the nine VALU operations
    rep("(int)(((p.x >> 6) & 0x03030303u) | ((p.y >> 4) & 0x0c0c0c0cu) | ((p.z >> 2) & 0x30303030u))};", "(int)(p.x ^ p.y ^ p.z)};")
    rep("wl = (i32x4){(int)(p.x & 0x3f3f3f3fu), (int)(p.y & 0x3f3f3f3fu), (int)(p.z & 0x3f3f3f3fu),", "wl = (i32x4){(int)p.x, (int)p.y, (int)p.z,")
elif name == "NOGX":      # no input projections at all (wrong results)
    rep("        else gxv[g][r] = *p;", "        else gxv[g][r] = (float)(g + r);")
elif name != "BASE":
    raise SystemExit("unknown variant")
os.makedirs(os.path.join(ROOT, "ab_tmp"), exist_ok=True)
src = os.path.join(CSRC, "_lstm_exp_%s.hip" % name)
open(src, "w").write(s)
out = os.path.join(ROOT, "ab_tmp", "lib_lstm_%s.so" % name)
try:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out, src,
                    os.path.join(CSRC, "rowops.hip")], check=True, cwd=CSRC)
finally:
    os.remove(src)
print(out)
