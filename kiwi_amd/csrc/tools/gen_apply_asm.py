#!/usr/bin/env python3
"""Generates kiwi_apply_asm.inc: the apply of ONE centroid group of accumulate_multi_kernel as a single hand-allocated
assembly routine (round 4).  python tools/gen_apply_asm.py > kiwi_apply_asm.inc

Why assembly: the apply wants the LDS reads of step k + 1 in flight while the arithmetic of step k runs, plus the next step's
coefficient line in a second scalar buffer.  With the shift moving by one sample per step (the regular case) step k + 1 needs only
its b[j-1] -- its b[j] is step k's b[j-1], in place --, and a register pair of step k's b[j] set is free the moment its one multiply
has issued: the next step's b[j-1] is read INTO it right there ("read behind").  Two register sets that swap roles every step, every
read of a step issued 100-400 cycles before the step needs it.  Written in C++ with asm read statements (three sets; the compiler
cannot express the read-behind) the register allocator needs 244 vector registers ("What did not pay in round
4" (iii)); allocated by hand it is 80 + 20.  The routine fixes its registers (clobber lists) and takes the accumulators, the LDS
address, the shifts and the coefficient pointer as operands.

Per accumulator the operations and their order are carry2_apply's (the reference's): application order of the GF components
1 2 3 [9] -> radial sum t1, 4 5 -> transverse sum t2, 6 7 8 [10] -> vertical; per component  T += wl * b[j];  T += wr * b[j-1]
(sparse_trace.f90:684-703), then the rotation by the back-azimuth change (seismogram.f90:196-203).  Radial and vertical
components are issued alternately (independent accumulators: four dependency chains instead of two).
exact: v_pk_mul_f32 + v_pk_add_f32, every product and sum rounded on its own.   fused: v_pk_fma_f32.

Registers (kernel budget: 168 VGPRs at three waves per SIMD):
  v[28:35]  t1a t1b t2a t2b     v[36:43] m0..m3 (products, exact only)     v44 va  v45 va4  v46 vn  v47 vn4
  v[48:87] bank 0   v[88:127] bank 1      (member m of a bank: pair 2m; m < NG: outputs q0,q1 of the m-th component in application
                                          order, m >= NG: outputs q2,q3)
  s[36:55], s[56:75] coefficient lines of the step at hand and of the next one, swapping roles every step   s[76:77] (cl, sl)
  s80 k  s81, s82 4 x integer shift of this step / the next (swapping)  s83 tmp  s84 4 d  s[86:87] coefficient pointer
  v44 address of position smax  v45 4 x shifts (lane k: step k)  v46, v47 addresses of the next reads
"""
import os
import sys

VARIANT = set(filter(None, os.environ.get('KIWI_ASM_VARIANT', '').split(',')))     # timing experiments only (wrong results)

SEQ = {10: [0, 1, 2, 8, 3, 4, 5, 6, 7, 9], 8: [0, 1, 2, 3, 4, 5, 6, 7]}
BANK = [48, 88]
T1A, T1B, T2A, T2B = 28, 30, 32, 34
M = [36, 38, 40, 42]
VA, VA4, VN, VN4 = 44, 45, 46, 47
CA, CB, CLSL = 36, 56, 76
SK, SE, SEN, ST, SD, SK1, SCP = 80, 81, 82, 83, 84, 85, 86


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def sp(r):
    return "s[%d:%d]" % (r, r + 1)


def member(bank, m):
    return BANK[bank] + 2 * m


def reads(bank, ng, K, addr):
    out = []
    for m in range(2 * ng):
        o = SEQ[ng][m % ng] * K + (2 if m >= ng else 0)
        out.append("ds_read2st64_b32 %s, v%d offset0:%d offset1:%d" % (vp(member(bank, m)), addr, o, o + 1))
    return out


def mac(fused, T, cpair, hi, X, tmp, first=False):
    """T += c * X  (c = low / high half of the scalar pair), as one or two instructions; first: T = 0 + c * X"""
    sel = "op_sel:[1,0]" if hi else "op_sel_hi:[0,1]"
    if fused and not first:
        sel3 = "op_sel:[1,0,0]" if hi else "op_sel_hi:[0,1,1]"
        return ["v_pk_fma_f32 %s, %s, %s, %s %s" % (vp(T), sp(cpair), vp(X), vp(T), sel3)], []
    mul = "v_pk_mul_f32 %s, %s, %s %s" % (vp(tmp), sp(cpair), vp(X), sel)
    if first:
        add = "v_pk_add_f32 %s, %s, 0 op_sel_hi:[1,0]" % (vp(T), vp(tmp))
    else:
        add = "v_pk_add_f32 %s, %s, %s" % (vp(T), vp(T), vp(tmp))
    return [mul], [add]


def arithmetic(ng, rot, fused, H, L, acc, K=None, behind=False, C=None):
    """instruction list of one step: acc = operand names of ar1a ar1b ar2a ar2b dza dzb.  behind: behind the multiplies that
    consume a member of the b[j] bank H, the NEXT step's b[j-1] of that member is read into its registers (address VN)"""
    nH1 = 4 if ng == 10 else 3
    nH = nH1 + 2
    ar1a, ar1b, ar2a, ar2b, dza, dzb = acc
    out = []
    t1 = (T1A, T1B) if rot else None
    t2 = (T2A, T2B) if rot else None

    def tgt(a):
        if a < nH1:
            return (vp(T1A), vp(T1B)) if rot else (ar1a, ar1b)
        if a < nH:
            return (vp(T2A), vp(T2B)) if rot else (ar2a, ar2b)
        return (dza, dzb)

    def emit_pair(items):
        """items: list of (a) application indices issued together (independent accumulators); each contributes
        Ta += wl*H.a; Tb += wl*H.b; Ta += wr*L.a; Tb += wr*L.b"""
        for half, src in ((False, H), (True, L)):          # wl * b[j] first, then wr * b[j-1]
            muls, adds = [], []
            ti = 0
            for a in items:
                Ta, Tb = tgt(a)
                cpair = C + 2 * a
                first = rot and (not half) and (a == 0 or a == nH1)      # t1 / t2 start from 0 + product
                for T, m in ((Ta, a), (Tb, ng + a)):
                    X = member(src, m)
                    tmp = M[ti % 4]
                    ti += 1
                    sel = "op_sel:[1,0]" if half else "op_sel_hi:[0,1]"
                    if first:
                        # t = 0 + p (the reference zeroes the sum, then adds) differs from t = p only where p is -0, and a
                        # sum that differs in the sign of a zero never reaches an output bit: t1 / t2 are only multiplied by
                        # cl / sl and ADDED to accumulators that start from +0 -- a round-to-nearest sum is -0 only if both
                        # operands are, so those are never -0, and x + (+-0) = x for every other x (NaN, infinities alike)
                        muls.append("v_pk_mul_f32 %s, %s, %s %s" % (T, sp(cpair), vp(X), sel))
                    elif fused:
                        sel3 = "op_sel:[1,0,0]" if half else "op_sel_hi:[0,1,1]"
                        muls.append("v_pk_fma_f32 %s, %s, %s, %s %s" % (T, sp(cpair), vp(X), T, sel3))
                    else:
                        muls.append("v_pk_mul_f32 %s, %s, %s %s" % (vp(tmp), sp(cpair), vp(X), sel))
                        adds.append("v_pk_add_f32 %s, %s, %s" % (T, T, vp(tmp)))
            out.extend(muls)
            if behind and not half:
                for a in items:
                    for m in (a, ng + a):
                        o = SEQ[ng][m % ng] * K + (2 if m >= ng else 0)
                        out.append("ds_read2st64_b32 %s, v%d offset0:%d offset1:%d" % (vp(member(H, m)), VN, o, o + 1))
            out.extend(adds)

    # radial and vertical components alternately, then the transverse pair, then the rotation
    for i in range(nH1):
        emit_pair([i, nH + i])
    emit_pair([nH1])
    emit_pair([nH1 + 1])
    if rot:
        # ar1 = ar1 + cl*t1 - sl*t2 ; ar2 = ar2 + cl*t2 + sl*t1   (seismogram.f90:200-203), cl = s76, sl = s77
        cl, sl = "op_sel_hi:[0,1]", "op_sel:[1,0]"
        if fused:
            cl3, sl3 = "op_sel_hi:[0,1,1]", "op_sel:[1,0,0]"
            out += ["v_pk_fma_f32 %s, %s, %s, %s %s" % (ar1a, sp(CLSL), vp(T1A), ar1a, cl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s" % (ar1b, sp(CLSL), vp(T1B), ar1b, cl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s" % (ar2a, sp(CLSL), vp(T2A), ar2a, cl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s" % (ar2b, sp(CLSL), vp(T2B), ar2b, cl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s neg_lo:[0,1,0] neg_hi:[0,1,0]" % (ar1a, sp(CLSL), vp(T2A), ar1a, sl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s neg_lo:[0,1,0] neg_hi:[0,1,0]" % (ar1b, sp(CLSL), vp(T2B), ar1b, sl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s" % (ar2a, sp(CLSL), vp(T1A), ar2a, sl3),
                    "v_pk_fma_f32 %s, %s, %s, %s %s" % (ar2b, sp(CLSL), vp(T1B), ar2b, sl3)]
        else:
            out += ["v_pk_mul_f32 %s, %s, %s %s" % (vp(M[0]), sp(CLSL), vp(T1A), cl),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[1]), sp(CLSL), vp(T1B), cl),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[2]), sp(CLSL), vp(T2A), cl),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[3]), sp(CLSL), vp(T2B), cl),
                    "v_pk_add_f32 %s, %s, %s" % (ar1a, ar1a, vp(M[0])),
                    "v_pk_add_f32 %s, %s, %s" % (ar1b, ar1b, vp(M[1])),
                    "v_pk_add_f32 %s, %s, %s" % (ar2a, ar2a, vp(M[2])),
                    "v_pk_add_f32 %s, %s, %s" % (ar2b, ar2b, vp(M[3])),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[0]), sp(CLSL), vp(T2A), sl),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[1]), sp(CLSL), vp(T2B), sl),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[2]), sp(CLSL), vp(T1A), sl),
                    "v_pk_mul_f32 %s, %s, %s %s" % (vp(M[3]), sp(CLSL), vp(T1B), sl),
                    "v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (ar1a, ar1a, vp(M[0])),
                    "v_pk_add_f32 %s, %s, %s neg_lo:[0,1] neg_hi:[0,1]" % (ar1b, ar1b, vp(M[1])),
                    "v_pk_add_f32 %s, %s, %s" % (ar2a, ar2a, vp(M[2])),
                    "v_pk_add_f32 %s, %s, %s" % (ar2b, ar2b, vp(M[3]))]
    return out


def routine(ng, K, rot, fused):
    name = "apply_group_asm_%d_k%d_%s_%s" % (ng, K, "rot" if rot else "plain", "fused" if fused else "exact")
    acc = ["%0", "%1", "%2", "%3", "%4", "%5"]
    # operands: %6 abase (v), %7 ishv (v), %8 smax (s), %9 n (s), %10 coef (s, 64 bit), %11 cl (s), %12 sl (s)
    L = []
    a = L.append
    SI = [SE, SEN]                        # 4 x integer shift of the step at hand / of the next one, swapping roles like the register sets
    a("s_mov_b32 s%d, %%11" % CLSL)
    a("s_mov_b32 s%d, %%12" % (CLSL + 1))
    a("s_mov_b64 %s, %%10" % sp(SCP))
    a("s_mov_b32 s%d, 0" % SK)
    a("s_load_dwordx16 s[%d:%d], %s, 0x0" % (CA, CA + 15, sp(SCP)))
    if ng == 10:
        a("s_load_dwordx4 s[%d:%d], %s, 0x40" % (CA + 16, CA + 19, sp(SCP)))
    a("v_lshlrev_b32 v%d, 2, %%7" % VA4)               # v45: 4 x the steps' integer shifts (lane k: step k)
    a("s_lshl_b32 s%d, %%8, 2" % ST)
    a("v_add_u32 v%d, s%d, %%6" % (VA, ST))           # v44: LDS address of position smax of the lane's first sample ...
    a("s_nop 1")                                       # (a vector register written by the instruction in front is not yet there for v_readlane)
    a("v_readlane_b32 s%d, v%d, 0" % (SI[0], VA4))
    a("v_subrev_u32 v%d, s%d, v%d" % (VN, SI[0], VA))  # ... of position e = smax - ishift: b[j-1] of step 0
    a("v_add_u32 v%d, 4, v%d" % (VN4, VN))
    L += reads(0, ng, K, VN4)            # bank 0 = b[j]
    L += reads(1, ng, K, VN)             # bank 1 = b[j-1]
    tail = []                            # the steps without a regular successor: out of line, the regular path falls through
    for r in range(2):
        H, Lb = r, 1 - r
        Cc, Cn = (CA, CB) if r == 0 else (CB, CA)        # the two coefficient buffers swap roles with the register sets
        a(".Lkiwi_step%d_%%=:" % r)
        a("s_waitcnt lgkmcnt(0)")
        a("s_add_u32 s%d, s%d, 1" % (SK, SK))
        a("s_mov_b32 s%d, 0x7fffffff" % SD)      # d = 0x7fffffff: "no successor" (neither read-behind nor reads behind the step)
        a("s_cmp_ge_u32 s%d, %%9" % SK)
        a("s_cbranch_scc1 .Lkiwi_plain%d_%%=" % r)
        a("v_readlane_b32 s%d, v%d, s%d" % (SI[1 - r], VA4, SK))
        a("s_add_u32 s%d, s%d, %d" % (SCP, SCP, 80))
        a("s_addc_u32 s%d, s%d, 0" % (SCP + 1, SCP + 1))
        a("s_load_dwordx16 s[%d:%d], %s, 0x0" % (Cn, Cn + 15, sp(SCP)))
        if ng == 10:
            a("s_load_dwordx4 s[%d:%d], %s, 0x40" % (Cn + 16, Cn + 19, sp(SCP)))
        a("s_sub_i32 s%d, s%d, s%d" % (SD, SI[1 - r], SI[r]))     # 4 x (positions the tile moves back by)
        a("v_subrev_u32 v%d, s%d, v%d" % (VN, SI[1 - r], VA))
        a("s_cmp_lg_u32 s%d, 4" % SD)
        a("s_cbranch_scc1 .Lkiwi_plain%d_%%=" % r)
        # regular successor: its b[j-1] is read behind this step's multiplies, into the b[j] bank
        L += arithmetic(ng, rot, fused, H, Lb, acc, K, True, C=Cc)
        if r == 1:
            a("s_branch .Lkiwi_step0_%=")
        t = tail.append
        t(".Lkiwi_plain%d_%%=:" % r)
        tail += arithmetic(ng, rot, fused, H, Lb, acc, C=Cc)
        t("s_cmp_eq_u32 s%d, 0x7fffffff" % SD)
        t("s_cbranch_scc1 .Lkiwi_end_%=")
        # irregular successor: its b[j] over this step's b[j-1] bank, its b[j-1] over the b[j] bank
        t("v_add_u32 v%d, 4, v%d" % (VN4, VN))
        tail += reads(Lb, ng, K, VN4)
        tail += reads(H, ng, K, VN)
        t("s_branch .Lkiwi_step%d_%%=" % (1 - r))
    L += tail
    a(".Lkiwi_end_%=:")
    if VARIANT:
        steps = [i for i, x in enumerate(L) if x.startswith(".Lkiwi_step0")][0]
        def keep(i, x):
            if "noarith" in VARIANT and x.startswith("v_pk_"): return False
            if "nolds" in VARIANT and i > steps and x.startswith("ds_read"): return False
            if "nocoef" in VARIANT and i > steps and x.startswith("s_load"): return False
            return True
        L = [x for i, x in enumerate(L) if keep(i, x)]
    text = "\\n\\t\"\n        \"".join(L)
    clob_v = ", ".join('"v%d"' % i for i in range(28, 128))
    clob_s = ", ".join('"s%d"' % i for i in list(range(CA, CB + 20)) + [CLSL, CLSL + 1] + list(range(SK, SCP + 2)))
    out = []
    out.append("__device__ __forceinline__ void %s(f2v &ar1a, f2v &ar1b, f2v &ar2a, f2v &ar2b, f2v &dza, f2v &dzb, unsigned abase, int ishv,\n"
               "        int smax, int n, const float *coef, float cl, float sl)" % name)
    out.append("{")
    out.append('    asm volatile(\n        "%s\\n\\t"' % text)
    out.append('        : "+v"(ar1a), "+v"(ar1b), "+v"(ar2a), "+v"(ar2b), "+v"(dza), "+v"(dzb)')
    out.append('        : "v"(abase), "v"(ishv), "s"(smax), "s"(n), "s"(coef), "s"(cl), "s"(sl)')
    out.append('        : "memory", "scc", "vcc", %s,\n          %s);' % (clob_s, clob_v))
    out.append("}")
    return "\n".join(out)


if __name__ == "__main__":
    print("// ---- (generated text: tools/gen_apply_asm.py) the apply of one centroid group as a hand-allocated assembly routine")
    print("// KIWI_ARITH selects the exact (v_pk_mul_f32 + v_pk_add_f32) or the fused (v_pk_fma_f32) text")
    for fused in (False, True):
        print("#if KIWI_ARITH == %d" % (1 if fused else 0))
        for ng in (10, 8):
            for K in (5, 9, 17):
                for rot in (True, False):
                    print(routine(ng, K, rot, fused))
        print("#endif")
    print("// ---- (end of generated text)")
