#!/usr/bin/env python3
"""Generates the LDS read statements of accumulate_multi_kernel's carried register sets (kiwi_accum.inc, "set3"):
python tools/gen_set3.py > /tmp/set3.inc, pasted between the `---- (generated text) set3` markers.

Tile-set layout the reads assume (floats; LDS_TILE = 64 K positions per component, K = 5 or 9):
  pair row r (r < NP):  2 * LDS_TILE floats at r * 2 * LDS_TILE, position p of the row's two components at 2 p, 2 p + 1
  single row c (c < 2): LDS_TILE floats at NP * 2 * LDS_TILE + c * LDS_TILE
Register set, member m (Set2_10: 20 pairs, Set2_8: 16):
  m = 4 r + q        (b_A[p_q], b_B[p_q]): the pair row's two components at the lane's output q (position e + u0 + 64 q): ONE ds_read_b64
  m = 4 NP + 2 c + h (b[p_2h], b[p_2h+1]): single component c at the lane's outputs 2 h, 2 h + 1: ds_read2st64_b32
"""


def gen(ng, K):
    npair = 4 if ng == 10 else 3
    nm = 4 * npair + 4
    lines = []
    lines.append("template <int SKIP> __device__ __forceinline__ void set3_read_%d_k%d(int d, unsigned a8, unsigned a4, const Set2_%d &S)" % (ng, K, ng))
    lines.append("{")
    o = []
    for m in range(4 * npair):
        r, q = divmod(m, 4)
        ofs = r * 8 * 64 * K + 512 * q
        o.append('                 "ds_read_b64 %%%d, %%%d%s\\n\\t"' % (m, nm, (" offset:%d" % ofs) if ofs else ""))
    for m in range(4 * npair, nm):
        c, h = divmod(m - 4 * npair, 2)
        o0 = 2 * npair * K + c * K + 2 * h
        o.append('                 "ds_read2st64_b32 %%%d, %%%d offset0:%d offset1:%d\\n\\t"' % (m, nm + 1, o0, o0 + 1))
    lines.append('    asm volatile("s_cmp_eq_u32 %%%d, %%%d\\n\\ts_cbranch_scc1 .Lkiwi_skip%%=\\n\\t"' % (nm + 2, nm + 3))
    lines += o
    lines.append('                 "\\n.Lkiwi_skip%=:"')
    half = nm // 2
    outs = ", ".join('"+v"(S.a%d)' % i for i in range(half)) + ", " + ", ".join('"+v"(S.b%d)' % i for i in range(half))
    lines.append('                 : %s : "v"(a8), "v"(a4), "s"(d), "i"(SKIP) : "memory", "scc");' % outs)
    lines.append("}")
    return "\n".join(lines)


if __name__ == "__main__":
    print("// ---- (generated text: tools/gen_set3.py) set3: reads of the component-pair tile layout")
    for ng in (10, 8):
        for K in (9, 5):
            print(gen(ng, K))
    print("// ---- (end of generated text) set3")
