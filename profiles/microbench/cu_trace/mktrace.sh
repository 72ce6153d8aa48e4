# builds kiwi_amd/libkiwi_hip_$SFX.so = the current sources + per-CU event trace in accumulate_multi_kernel
set -e
cd "$(dirname "$0")/../../../kiwi_amd/csrc"
SFX=${SFX:-T}
D=/tmp/ktr_$SFX
mkdir -p $D build_$SFX
KIWI_ASM_VARIANT=$VARIANT python3 tools/gen_apply_asm.py > $D/kiwi_apply_asm.inc
export D
python3 - <<'PY'
import os
D = os.environ['D']
s = open("kiwi_accum.inc").read()
k0 = s.index("accumulate_multi_kernel(", s.index("// Build of accumulate_multi_kernel"))
k1 = s.index("// accumulate, cell groups: the raw node traces stay in registers")
t = s[k0:k1]
def rep(a, b):
    global t
    assert t.count(a) == 1, a[:60]
    t = t.replace(a, b)
rep("    const int u0 = 256 * (wv % WPS) + lane;              // the lane's first sample of its source's tile (the others: + 64 q)\n",
"""    const int u0 = 256 * (wv % WPS) + lane;
    const unsigned hwid_ = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    const unsigned xcc_ = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
    const bool rec0_ = ((hwid_ >> 8) & 0xffu) == 0u && xcc_ == 0u && wv == 0;
    unsigned blk_ = 0xffffffffu, grp_ = 0;
    unsigned long long tent_ = 0;
    if (rec0_) { tent_ = __builtin_amdgcn_s_memtime(); unsigned b_ = 0; if (lane == 0) b_ = atomicAdd(&g_trace_n, 1u); blk_ = __builtin_amdgcn_readfirstlane(b_); }
    const bool rec_ = rec0_ && blk_ < 3000u;
#define KIWI_TICK(k) do { if (rec_ && grp_ < 30u) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_trace[(blk_ * 32 + grp_) * 6 + (k)] = tn_; } } while (0)
""")
rep("    while (c < nc) {\n        GeoRec g[NS];\n        int smaxs[NS], smins[NS], npos = 0;", "    while (c < nc) {\n        KIWI_TICK(0);\n        GeoRec g[NS];\n        int smaxs[NS], smins[NS], npos = 0;")
rep("        __builtin_amdgcn_s_setprio(1);\n        if (shared) KIWI_MULTI_BUILD(ta, tb, 0, -1, npos, shadow);", "        KIWI_TICK(1);\n        __builtin_amdgcn_s_setprio(1);\n        if (shared) KIWI_MULTI_BUILD(ta, tb, 0, -1, npos, shadow);")
rep("""        if (cend < nc) { ta = tab_of(0)[(size_t)cend * 128 + lane]; tb = tab_of(0)[(size_t)cend * 128 + 64 + lane]; }
        __syncthreads();
        __builtin_amdgcn_s_setprio(0);""", """        if (cend < nc) { ta = tab_of(0)[(size_t)cend * 128 + lane]; tb = tab_of(0)[(size_t)cend * 128 + 64 + lane]; }
        KIWI_TICK(2);
        __syncthreads();
        KIWI_TICK(3);
        __builtin_amdgcn_s_setprio(0);""")
rep("        c = cend;                                        // (the barrier in front of the next group's LDS writes: multi_build)\n    }",
    "        KIWI_TICK(4);\n        grp_++;\n        c = cend;\n    }\n    unsigned long long tle_ = 0; if (rec_) tle_ = __builtin_amdgcn_s_memtime();")
rep("""                fp.partial[((size_t)js * fp.nmis + rv.slot0 + k) * fp.nparts + tile * WPS + (wv % WPS)] = acc;
        }
    }
}""", """                fp.partial[((size_t)js * fp.nmis + rv.slot0 + k) * fp.nparts + tile * WPS + (wv % WPS)] = acc;
        }
    }
    if (rec_) { const unsigned long long tx_ = __builtin_amdgcn_s_memtime();
        if (lane == 0) { unsigned long long *o_ = &g_trace[(blk_ * 32 + 30) * 6]; o_[0] = tent_; o_[1] = tle_; o_[2] = tx_; o_[3] = grp_; o_[4] = (unsigned long long)(blockIdx.y * gridDim.x + blockIdx.x); o_[5] = hwid_; } }
}""")
s = s[:k0] + t + s[k1:]
open(D + "/ka_timing.inc", "w").write(s)
h = open("kiwi_accum.hip").read()
h = h.replace("namespace kiwi {\nnamespace KIWI_ARITH_NS {\n", "namespace kiwi {\nnamespace KIWI_ARITH_NS {\n__device__ unsigned long long g_trace[3000 * 32 * 6];\n__device__ unsigned g_trace_n;\n")
h = h.replace('#include "kiwi_accum.inc"', '#include "' + D + '/ka_timing.inc"')
h += """
#if KIWI_FAMILY == 3
#define KIWI_CAT2(a, b) a##b
#define KIWI_CAT(a, b) KIWI_CAT2(a, b)
extern "C" int KIWI_CAT(kiwi_hip_exp_trace_, KIWI_ARITH_NS)(unsigned long long *out, unsigned *n, int reset)
{
    using namespace kiwi::KIWI_ARITH_NS;
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 3000 * 32 * 6);
    hipMemcpyFromSymbol(n, HIP_SYMBOL(g_trace_n), sizeof(unsigned));
    if (reset) { unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &z, sizeof(z)); hipMemset((void *)0, 0, 0); }
    return 0;
}
#endif
"""
open(D + "/ka_timing.hip", "w").write(h)
PY
mkdir -p build_$SFX
cp build/kiwi_hip.o build/accum_1_*.o build/accum_2_*.o build/accum_4_*.o build_$SFX/
for ar in 0 1; do n=$([ $ar = 0 ] && echo exact || echo fused); c=$([ $ar = 0 ] && echo off || echo fast)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -fopenmp -Wall -Wno-unused-result -I. -ffp-contract=$c -DKIWI_ARITH=$ar -DKIWI_FAMILY=3 -c -o build_$SFX/accum_3_$n.o $D/ka_timing.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -fopenmp -shared -Wl,--version-script=kiwi_hip.map -o ../libkiwi_hip_$SFX.so build_$SFX/*.o -lhipfft 2>/dev/null || /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -fopenmp -shared -o ../libkiwi_hip_$SFX.so build_$SFX/*.o -lhipfft
nm -D ../libkiwi_hip_$SFX.so | grep -c exp_trace
