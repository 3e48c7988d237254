"""Builds ab_tmp/mst_prof.so: csrc/mst.hip with cycle counters around the phases of cle_run (read by scripts/bench_mst.py)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "nlp-cube_amd/csrc/mst.hip")).read()


def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)


rep('template <typename IdT>\n__device__ bool cle_run(const MstLdsT<IdT>& m, int n, int lane, bool has_nan) {', '''__device__ long long g_prof[16];
#define PT(i) { long long t_ = clock64(); prof[i] += t_ - tlast; tlast = t_; }
template <typename IdT>
__device__ bool cle_run(const MstLdsT<IdT>& m, int n, int lane, bool has_nan) {
  long long prof[10] = {0,0,0,0,0,0,0,0,0,0}; long long tlast = clock64(); long long tstart = tlast;''')
rep('''    MST_SYNC();
    // ---- 2. cycles by pointer doubling''', '''    MST_SYNC(); PT(0)
    // ---- 2. cycles by pointer doubling''')
rep('''    chosen = wave_max_i(chosen);
    if (chosen < 0) break;  // a tree''', '''    chosen = wave_max_i(chosen); PT(1)
    if (chosen < 0) break;  // a tree''')
rep('''    MST_SYNC();
    // ---- 4. product of the cycle's arc scores, in cycle order''', '''    MST_SYNC(); PT(2)
    // ---- 4. product of the cycle's arc scores, in cycle order''')
rep('''    IdT* rec_d = rec_h + nn;
    bool new_nan = false;''', '''    IdT* rec_d = rec_h + nn; PT(3)
    bool new_nan = false;''')
rep('''    MST_SYNC();
    // ---- 6. contracted node takes the slot''', '''    MST_SYNC(); PT(4)
    // ---- 6. contracted node takes the slot''')
rep('''    cur = nn + 1;
    rec_end += need;
    ++levels;''', '''    cur = nn + 1;
    rec_end += need;
    ++levels; PT(5)''')
rep('''    MST_SYNC();
  }
  return true;
}''', '''    MST_SYNC();
  }
  PT(6)
  if (lane == 0) { for (int i = 0; i < 7; ++i) atomicAdd((unsigned long long*)&g_prof[i], (unsigned long long)prof[i]); atomicAdd((unsigned long long*)&g_prof[8], (unsigned long long)levels); atomicAdd((unsigned long long*)&g_prof[9], 1ull); atomicAdd((unsigned long long*)&g_prof[10], (unsigned long long)(clock64() - tstart)); }
  return true;
}''')
s += '''
extern "C" int cube_mst_prof(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 16) == hipSuccess ? 0 : 1; }
'''
s = s.replace('#include "common.h"', '#include "../nlp-cube_amd/csrc/common.h"').replace('#include "../../include/cube_hip.h"', '#include "../include/cube_hip.h"')
os.makedirs(os.path.join(ROOT, "ab_tmp"), exist_ok=True)
open(os.path.join(ROOT, "ab_tmp/mst_prof.hip"), "w").write(s)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-o",
                       os.path.join(ROOT, "ab_tmp/mst_prof.so"), os.path.join(ROOT, "ab_tmp/mst_prof.hip"),
                       os.path.join(ROOT, "nlp-cube_amd/csrc/rowops.hip")])
