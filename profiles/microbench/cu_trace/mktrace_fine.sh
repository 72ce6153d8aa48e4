# builds kiwi_amd/libkiwi_hip_trace.so = the current sources + the fine per-CU event trace of round 5 in accumulate_multi_kernel (ten events
# per centroid group, waves 0 and 3).  Then, on the GPU box:
#   KIWI_HIP_LIB=$PWD/kiwi_amd/libkiwi_hip_trace.so python profiles/microbench/cu_trace/trace_fine.py cfg3 4096
set -e
cd "$(dirname "$0")/../../../kiwi_amd/csrc"
D=/tmp/ktr_fine
mkdir -p $D $D/build
export D
python3 - <<'PY'
import os
D = os.environ['D']
s = open("kiwi_accum.inc").read()
k0 = s.index("template <int NG, bool FAST, int NS, typename Shadow>")
k1 = s.index("// accumulate, cell groups: the raw node traces stay in registers")
t = s[k0:k1]
def rep(a, b):
    global t
    assert t.count(a) == 1, (t.count(a), a[:70])
    t = t.replace(a, b)
NT = 12
# multi_build gets a tick functor
rep("template <int NG, bool FAST, int NS, typename Shadow>", "template <int NG, bool FAST, int NS, typename Shadow, typename Tick>")
rep("                                            Shadow shadow)\n{", "                                            Shadow shadow, Tick tick)\n{")
rep("    __syncthreads();\n    // In the shadow of the loads", "    tick(2);\n    __syncthreads();\n    tick(3);\n    // In the shadow of the loads")
rep("    shadow();\n#pragma unroll\n    for (int i = 0; i < N; i++) {", "    shadow();\n    tick(4);\n#pragma unroll\n    for (int i = 0; i < N; i++) {")
rep("    if (whalo) {\n#pragma unroll\n        for (int s = 0; s < NS; s++)\n            if (only < 0 || only == s) halo_finish", "    tick(5);\n    if (whalo) {\n#pragma unroll\n        for (int s = 0; s < NS; s++)\n            if (only < 0 || only == s) halo_finish")
rep("    const int u0 = 256 * (wv % WPS) + lane;              // the lane's first sample of its source's tile (the others: + 64 q)\n",
"""    const int u0 = 256 * (wv % WPS) + lane;
    const unsigned hwid_ = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    const unsigned xcc_ = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);
    const bool rec0_ = ((hwid_ >> 8) & 0xffu) == 0u && xcc_ == 0u;
    __shared__ unsigned blk_sh_;
    unsigned blk_ = 0xffffffffu, grp_ = 0;
    if (rec0_ && tid == 0) blk_sh_ = atomicAdd(&g_trace_n, 1u);
    __syncthreads();
    if (rec0_) blk_ = blk_sh_;
    const bool rec_ = rec0_ && blk_ < 1500u && (wv == 0 || wv == 3);
    const unsigned wslot_ = wv == 0 ? 0u : 1u;
#define KIWI_TICK(k) do { if (rec_ && grp_ < 24u) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_trace[((blk_ * 2 + wslot_) * 24 + grp_) * 12 + (k)] = tn_; } } while (0)
    auto tick = [&](int k) { KIWI_TICK(k); };
""")
rep("    while (c < nc) {\n        GeoRec g[NS];\n        int smaxs[NS], smins[NS], npos = 0;", "    while (c < nc) {\n        KIWI_TICK(0);\n        GeoRec g[NS];\n        int smaxs[NS], smins[NS], npos = 0;")
rep("        __builtin_amdgcn_s_setprio(1);\n        if (shared) KIWI_MULTI_BUILD(ta, tb, 0, -1, npos, shadow);", "        KIWI_TICK(1);\n        __builtin_amdgcn_s_setprio(1);\n        if (shared) KIWI_MULTI_BUILD(ta, tb, 0, -1, npos, shadow);")
t = t.replace("hact, whalo, hig, hph, SHADOW);", "hact, whalo, hig, hph, SHADOW, tick);")
assert t.count("SHADOW, tick)") == 2
rep("        // head records and descriptors of the NEXT group: in flight while this group is applied\n", "        KIWI_TICK(6);\n        // head records and descriptors of the NEXT group: in flight while this group is applied\n")
rep("        __syncthreads();\n        __builtin_amdgcn_s_setprio(0);", "        KIWI_TICK(7);\n        __syncthreads();\n        KIWI_TICK(8);\n        __builtin_amdgcn_s_setprio(0);")
rep("        c = cend;                                        // (the barrier in front of the next group's LDS writes: multi_build)\n    }",
    "        KIWI_TICK(9);\n        grp_++;\n        c = cend;\n    }")
s = s[:k0] + t + s[k1:]
open(D + "/ka_timing.inc", "w").write(s)
h = open("kiwi_accum.hip").read()
h = h.replace("namespace kiwi {\nnamespace KIWI_ARITH_NS {\n", "namespace kiwi {\nnamespace KIWI_ARITH_NS {\n__device__ unsigned long long g_trace[1500 * 2 * 24 * 12];\n__device__ unsigned g_trace_n;\n")
h = h.replace('#include "kiwi_accum.inc"', '#include "' + D + '/ka_timing.inc"')
h += """
#if KIWI_FAMILY == 3
#define KIWI_CAT2(a, b) a##b
#define KIWI_CAT(a, b) KIWI_CAT2(a, b)
extern "C" int KIWI_CAT(kiwi_hip_exp_trace_, KIWI_ARITH_NS)(unsigned long long *out, unsigned *n, int reset)
{
    using namespace kiwi::KIWI_ARITH_NS;
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 1500 * 2 * 24 * 12);
    hipMemcpyFromSymbol(n, HIP_SYMBOL(g_trace_n), sizeof(unsigned));
    if (reset) { unsigned z = 0; hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &z, sizeof(z)); hipMemset(out, 0, 0); }
    return 0;
}
#endif
"""
open(D + "/ka_timing.hip", "w").write(h)
PY
make -s
cp build/kiwi_hip.o build/accum_1_*.o build/accum_2_*.o build/accum_4_*.o $D/build/
for ar in 0 1; do n=$([ $ar = 0 ] && echo exact || echo fused); c=$([ $ar = 0 ] && echo off || echo fast)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -fopenmp -Wall -Wno-unused-result -I. -ffp-contract=$c -DKIWI_ARITH=$ar -DKIWI_FAMILY=3 -c -o $D/build/accum_3_$n.o $D/ka_timing.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -fopenmp -shared -o ../libkiwi_hip_trace.so $D/build/*.o -lhipfft
nm -D ../libkiwi_hip_trace.so | grep -c exp_trace
