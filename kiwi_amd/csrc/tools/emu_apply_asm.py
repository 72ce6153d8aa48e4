#!/usr/bin/env python3
"""A CPU interpreter for the hand-allocated apply routines of kiwi_apply_asm.inc (test infrastructure: tests/test_apply_asm_emulated.py).

The routines are straight-line gfx950 assembly with fixed registers, two register banks that swap roles every step and LDS reads issued
"behind" the multiplies that free their destination.  What can go wrong in such text is data flow -- a register read before the wait
that makes it valid, a read issued into a register that is still needed, a coefficient buffer used in the wrong step -- and all of that
is decidable without a GPU.  This interpreter executes the instruction subset the generator emits on 64 lanes of numpy float32 with the
memory model that matters here:

* `ds_read2st64_b32` and `s_load_*` deliver at the next `s_waitcnt lgkmcnt(0)`: until then their destinations are PENDING, and an
  instruction that reads a pending register is an error (the hardware would read stale data);
* a load's destination keeps its OLD value until that wait (the data arrives late, in program order behind every earlier instruction's
  operand fetch): an instruction between the issue and the wait that reads it sees the old value -- which is flagged as above --, and the
  value the load returns is the memory content at issue (the tile is read-only during an apply);
* packed fp32 arithmetic: `v_pk_mul_f32` / `v_pk_add_f32` round each operation to fp32 on its own; `v_pk_fma_f32` is one rounding
  (product exact in fp64, sum rounded once to fp64 and then to fp32: double rounding is possible in principle and irrelevant here -- the
  reference model of the test uses the same expression).

Operands %0 .. %12 of the inline-assembly statement are bound to registers outside the routine's fixed ranges (v[200:211], s[90:95])."""
import re

import numpy as np

F32 = np.float32


class EmuError(AssertionError):
    pass


def routines(text):
    """{name: [instruction, ...]} of every routine in the generated file"""
    out = {}
    for m in re.finditer(r"void (apply_group_asm_\w+)\(.*?asm volatile\(\n(.*?)\n\s*: \"\+v\"", text, re.S):
        ins = []
        for line in m.group(2).split("\n"):
            line = line.strip()
            if not line.startswith('"'):
                continue
            s = line[1:line.rindex('"')]
            s = s.replace("\\n\\t", "").replace("%=", "").strip()
            if s:
                ins.append(s)
        out[m.group(1)] = ins
    return out


OPERANDS = {"%0": "v[200:201]", "%1": "v[202:203]", "%2": "v[204:205]", "%3": "v[206:207]", "%4": "v[208:209]", "%5": "v[210:211]",
            "%6": "v212", "%7": "v213", "%8": "s90", "%9": "s91", "%10": "s[92:93]", "%11": "s94", "%12": "s95"}


def _bind(ins):
    for k in sorted(OPERANDS, key=len, reverse=True):       # %10 .. %12 before %1
        ins = ins.replace(k, OPERANDS[k])
    return ins


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


class Machine:
    def __init__(self, lds, coef):
        self.v = np.zeros((256, 64), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.scc = 0
        self.lds = np.asarray(lds, F32)                   # dwords
        self.coef = np.asarray(coef, F32)                 # the scalar loads' memory: a "pointer" is a byte offset into it
        self.pend_v, self.pend_s = {}, {}                 # register -> value that arrives at the next lgkmcnt(0) wait
        self.count = {}

    # ---- operand access
    def _chk_v(self, r, ins):
        if r in self.pend_v:
            raise EmuError("v%d read before the wait behind its load: %s" % (r, ins))

    def _chk_s(self, r, ins):
        if r in self.pend_s:
            raise EmuError("s%d read before the wait behind its load: %s" % (r, ins))

    def rv(self, r, ins):
        self._chk_v(r, ins)
        return self.v[r]

    def rs(self, r, ins):
        self._chk_s(r, ins)
        return self.s[r]

    def wv(self, r, val):
        self.pend_v.pop(r, None)          # (a write over a pending destination: the late data would clobber it -- flagged below)
        self.v[r] = np.asarray(val).view(np.uint32) if np.asarray(val).dtype == F32 else np.asarray(val, np.uint32)

    def src32(self, tok, ins):
        """a 32-bit source as a uint32 lane vector"""
        tok = tok.strip()
        if re.fullmatch(r"v\d+", tok):
            return self.rv(int(tok[1:]), ins)
        if re.fullmatch(r"s\d+", tok):
            return np.full(64, self.rs(int(tok[1:]), ins), np.uint32)
        return np.full(64, int(tok, 0) & 0xffffffff, np.uint32)

    def src64(self, tok, ins):
        """a packed source: (lo, hi) float32 lane vectors"""
        tok = tok.strip()
        m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
        if m:
            a = int(m.group(2))
            if m.group(1) == "v":
                return self.rv(a, ins).view(F32), self.rv(a + 1, ins).view(F32)
            return np.full(64, self.rs(a, ins), np.uint32).view(F32), np.full(64, self.rs(a + 1, ins), np.uint32).view(F32)
        raise EmuError("packed operand: " + ins)

    # ---- execution
    def run(self, ins_list, max_steps=200000):
        ins_list = [_bind(i) for i in ins_list]
        labels = {i[:-1]: n for n, i in enumerate(ins_list) if i.endswith(":")}
        pc, steps = 0, 0
        while pc < len(ins_list):
            steps += 1
            if steps > max_steps:
                raise EmuError("no end")
            ins = ins_list[pc]
            pc += 1
            if ins.endswith(":"):
                continue
            op, _, rest = ins.partition(" ")
            self.count[op] = self.count.get(op, 0) + 1
            mods = {}
            for mm in re.finditer(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([\d,]+)\]", rest):
                mods[mm.group(1)] = [int(x) for x in mm.group(2).split(",")]
            args = [a.strip() for a in re.sub(r"\s+(op_sel_hi|op_sel|neg_lo|neg_hi):\[[\d,]+\]", "", rest).split(",")] if rest else []
            if op in ("s_nop",):
                continue
            if op == "s_waitcnt":
                if rest.strip() != "lgkmcnt(0)":
                    raise EmuError("wait: " + ins)
                for r, val in self.pend_v.items():
                    self.v[r] = val
                for r, val in self.pend_s.items():
                    self.s[r] = val
                self.pend_v, self.pend_s = {}, {}
                continue
            if op == "s_branch":
                pc = labels[args[0]]
                continue
            if op == "s_cbranch_scc1":
                if self.scc:
                    pc = labels[args[0]]
                continue
            if op == "s_mov_b32":
                self.s[int(args[0][1:])] = self.src32(args[1], ins)[0]
                continue
            if op == "s_mov_b64":
                d = int(re.match(r"s\[(\d+)", args[0]).group(1))
                a = int(re.match(r"s\[(\d+)", args[1]).group(1))
                self.s[d], self.s[d + 1] = self.rs(a, ins), self.rs(a + 1, ins)
                continue
            if op in ("s_add_u32", "s_addc_u32", "s_sub_i32", "s_lshl_b32"):
                a, b = int(self.src32(args[1], ins)[0]), int(self.src32(args[2], ins)[0])
                if op == "s_add_u32":
                    r = a + b
                    self.scc = 1 if r > 0xffffffff else 0
                elif op == "s_addc_u32":
                    r = a + b + self.scc
                    self.scc = 1 if r > 0xffffffff else 0
                elif op == "s_sub_i32":
                    r = a - b
                else:
                    r = a << (b & 31)
                self.s[int(args[0][1:])] = r & 0xffffffff
                continue
            if op in ("s_cmp_ge_u32", "s_cmp_lg_u32", "s_cmp_eq_u32"):
                a, b = int(self.src32(args[0], ins)[0]), int(self.src32(args[1], ins)[0])
                self.scc = int({"s_cmp_ge_u32": a >= b, "s_cmp_lg_u32": a != b, "s_cmp_eq_u32": a == b}[op])
                continue
            if op in ("s_load_dwordx16", "s_load_dwordx4"):
                n = 16 if op.endswith("16") else 4
                d = int(re.match(r"s\[(\d+)", args[0]).group(1))
                p = int(re.match(r"s\[(\d+)", args[1]).group(1))
                addr = (int(self.rs(p + 1, ins)) << 32 | int(self.rs(p, ins))) + int(args[2], 0)
                if addr % 4 or addr // 4 + n > len(self.coef):
                    raise EmuError("scalar load outside of the coefficient lines: " + ins)
                for i in range(n):
                    self.pend_s[d + i] = self.coef[addr // 4 + i].view(np.uint32)
                continue
            if op == "v_lshlrev_b32":
                self.wv(int(args[0][1:]), (self.src32(args[2], ins).astype(np.uint64) << (int(args[1], 0) & 31)).astype(np.uint32))
                continue
            if op in ("v_add_u32", "v_subrev_u32"):
                a, b = self.src32(args[1], ins).astype(np.int64), self.src32(args[2], ins).astype(np.int64)
                self.wv(int(args[0][1:]), (((a + b) if op == "v_add_u32" else (b - a)) & 0xffffffff).astype(np.uint32))
                continue
            if op == "v_readlane_b32":
                lane = int(self.src32(args[2], ins)[0]) & 63
                self.s[int(args[0][1:])] = self.rv(int(args[1][1:]), ins)[lane]
                continue
            if op == "ds_read2st64_b32":
                d = int(re.match(r"v\[(\d+)", args[0]).group(1))
                am = re.match(r"v(\d+)(.*)", args[1])
                addr = self.rv(int(am.group(1)), ins).astype(np.int64)
                o0 = re.search(r"offset0:(\d+)", ins)
                o1 = re.search(r"offset1:(\d+)", ins)
                o0, o1 = (int(o0.group(1)) if o0 else 0), (int(o1.group(1)) if o1 else 0)
                for k, o in ((0, o0), (1, o1)):
                    a = addr + 256 * o
                    if (a % 4).any() or (a < 0).any() or (a // 4 >= len(self.lds)).any():
                        raise EmuError("LDS read outside of the tile set: " + ins)
                    if d + k in self.pend_v:
                        raise EmuError("two loads into v%d in flight: %s" % (d + k, ins))
                    self.pend_v[d + k] = self.lds[a // 4].view(np.uint32)
                continue
            if op in ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32"):
                nsrc = 3 if op == "v_pk_fma_f32" else 2
                d = int(re.match(r"v\[(\d+)", args[0]).group(1))
                srcs = []
                for i in range(nsrc):
                    t = args[1 + i]
                    if re.fullmatch(r"-?\d+(\.\d+)?", t):
                        c = np.full(64, F32(float(t)), F32)
                        srcs.append((c, c))
                    else:
                        srcs.append(self.src64(t, ins))
                sel = mods.get("op_sel", [0] * nsrc) + [0] * nsrc
                selh = mods.get("op_sel_hi", [1] * nsrc) + [1] * nsrc
                ngl = mods.get("neg_lo", [0] * nsrc) + [0] * nsrc
                ngh = mods.get("neg_hi", [0] * nsrc) + [0] * nsrc
                lo = [(-srcs[i][sel[i]] if ngl[i] else srcs[i][sel[i]]) for i in range(nsrc)]
                hi = [(-srcs[i][selh[i]] if ngh[i] else srcs[i][selh[i]]) for i in range(nsrc)]
                if op == "v_pk_mul_f32":
                    rl, rh = lo[0] * lo[1], hi[0] * hi[1]
                elif op == "v_pk_add_f32":
                    rl, rh = lo[0] + lo[1], hi[0] + hi[1]
                else:
                    rl, rh = fma32(lo[0], lo[1], lo[2]), fma32(hi[0], hi[1], hi[2])
                for k, val in ((0, rl), (1, rh)):
                    if d + k in self.pend_v:
                        raise EmuError("arithmetic into v%d while a load into it is in flight: %s" % (d + k, ins))
                    self.v[d + k] = val.astype(F32).view(np.uint32)
                continue
            raise EmuError("instruction the interpreter does not know: " + ins)
        if self.pend_v or self.pend_s:
            # loads still in flight at the end are legal only if nothing reads them: the routine's clobber list covers them
            pass
        return self
