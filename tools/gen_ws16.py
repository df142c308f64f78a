#!/usr/bin/env python3
"""Generator of the compute wavefronts' instruction stream of conv_bf16_ws16_kernel (yogo_amd/csrc/conv_bf16_ws16.hip):
    python tools/gen_ws16.py > yogo_amd/csrc/conv_bf16_ws16_asm.inc
The whole compute role -- tile loop, period loop, tile seam -- is ONE asm statement (W16_TXT_ROLE / W16_TXT_ROLEB with the bias): every
register that carries state (operand quads, operand addresses, accumulators, loop counters) is a fixed register named here and listed
as a clobber, so the compiler never sees, copies or spills any of them.  (A first form with one statement per period and the sixteen
operand quads as "+v" operands made hipcc shuffle 64 registers between the statements and spill a hundred of them.)

period = (chunk pair, kernel row): 3 taps x 8 row blocks x 4 pixel blocks = 96 v_mfma_f32_16x16x32_bf16 per wavefront.  Its text runs
from barrier to barrier: the last four row-block slots of the PREVIOUS period (operands already in registers) with the first operand
reads of the new period in their gaps, then slots 0..19 of the new period, s_waitcnt lgkmcnt(0), s_barrier -- every s_waitcnt lgkmcnt in
between is counted by the generator from the issue order of the DS operations.
  slot   = one (tap, row block): 4 MFMAs (pixel blocks 0..3); accumulator tile (rb, pb) = a[4 * (4 rb + pb) : +3]
  A ring = 8 quads v[24:55]: slot s uses quad s % 8, read 6 slots ahead (ds_read_b128 at %[pa] + weight slot + kx * 8192 + rb * 256)
  B sets = X = v[56:71], Y = v[72:87]: pass q + 1's four quads are read in the first slots of pass q (address v[20 + pb] + kx * 16)
  v[16:19] = this lane's pixel-block addresses of the tile (pair slot 0, kernel row 0); v[20:23] = ... of the period; the next period's
  are set in the gaps behind the period's last B read (v_add_u32 with the scalar s_io = pair slot + kernel row * row pitch)
kinds: M  tap-major (passes = taps, 8 slots each; the B set in use flips from period to period)
       F  group-major first period of a tile (row blocks 0-3 all taps, then 4-7; a tile's first MFMAs start from 0), barrier X in front of
          slot 3, then the epilogue of the PREVIOUS tile's row blocks 4-7 in the MFMA gaps of slots 3..14 -> staging area
       F0 the same without prefix and epilogue (the workgroup's first tile)
       L  group-major last period: the epilogue of THIS tile's row blocks 0-3 in the gaps of slots 10..19 -> staging area; the mailbox of the
          next tile (written by the loaders in period 1) is read at its start and turned into v[32:39] / the scalars at its end
       T  tail of the workgroup's last tile: prefix MFMAs, barrier X, the epilogue of row blocks 4-7 on its own, barrier
epilogue of an accumulator tile: 4 v_accvgpr_read, 2 v_cvt_pk_bf16_f32, one ds_write_b64 (temporaries v88..v95); the bias is the C operand of a
tile's first MFMA (a[128:159], written once at the statement's start)
"""
import re
import sys

NA, DA = 8, 6
# timing-only ablations (wrong results): python tools/gen_ws16.py noepi|nobarx|... > yogo_amd/csrc/obj_var/TAG.inc, then
# bash yogo_amd/csrc/build.sh variant TAG conv_bf16_ws16 -DW16_ASM_INC='"obj_var/TAG.inc"'
ABL = set(sys.argv[1:])
WSLOT, ISLOT = 24576, 32768
# register map (96 arch VGPRs: v0..v15 hold the statement's inputs, the rest is named here; 160 accumulator registers: a[0:127] the tiles,
# a[128:159] the bias of this lane's channels -- row block rb: a[128 + 4 rb : + 3] --, the C operand of a tile's first MFMA)
T_RD = [[88, 89, 90, 91]]                                    # accumulator read-outs
T_OUT = [[92, 93], [94, 95]]                                 # converted pairs (two rotating)
V_PBL, V_P, V_A, V_X, V_Y = 16, 20, 24, 56, 72
A_BIAS = 128
# scalars (fixed, clobbered): row pitch of the staged tile, next period's offset, kernel row / pair of the next period, loop count, temporaries
S_LW, S_IO, S_R, S_P, S_CNT, S_T0, S_T1, S_HAS, S_PERKB = "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s87"


def slots_of(group_major):
    if not group_major:
        passes = [(kx, list(range(8))) for kx in range(3)]
    else:
        passes = [(kx, [g * 4 + i for i in range(4)]) for g in range(2) for kx in range(3)]
    flat = []
    for q, (kx, rbs) in enumerate(passes):
        for rb in rbs:
            flat.append((q, kx, rb))
    assert len(flat) == 24 and [f[1:] for f in flat[20:]] == [(2, 4), (2, 5), (2, 6), (2, 7)]
    return passes, flat


def quad(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


class Stream:
    def __init__(self, par, wslot, bias=False):
        self.bias = bias
        self.lines = []
        self.issued = 0          # DS operations issued so far
        self.done = 0            # ... known complete (by an emitted s_waitcnt)
        self.producer = {}       # register name -> index of the DS read that loads it
        self.par, self.wslot = par, wslot

    def reg(self, name):         # symbolic operand quad -> physical registers
        k, i = name[0], int(name[1:])
        if k == "a":
            return quad(V_A, i)
        cur_is_x = self.par == 0
        if k == "c":
            return quad(V_X if cur_is_x else V_Y, i)
        return quad(V_Y if cur_is_x else V_X, i)

    def emit(self, s):
        self.lines.append(s)

    def need(self, *regs):
        idx = max([self.producer.get(r, -1) for r in regs])
        if idx >= self.done:
            n = self.issued - (idx + 1)
            assert 0 <= n <= 15, n
            self.emit(f"s_waitcnt lgkmcnt({n})")
            self.done = idx + 1

    def a_read(self, dst, imm):
        imm += self.wslot * WSLOT
        assert imm < 65536
        self.emit(f"ds_read_b128 {self.reg(dst)}, %[pa] offset:{imm}")
        self.producer[dst] = self.issued
        self.issued += 1

    def b_read(self, dst, pb, imm):
        self.emit(f"ds_read_b128 {self.reg(dst)}, v{V_P + pb} offset:{imm}")
        self.producer[dst] = self.issued
        self.issued += 1

    def read_fixed(self, regs, addr, imm, tag):
        self.emit(f"ds_read_b128 v[{regs[0]}:{regs[3]}], {addr} offset:{imm}")
        self.producer[tag] = self.issued
        self.issued += 1

    def ds_write(self, pair, imm):
        self.emit(f"ds_write_b64 %[stg], v[{pair[0]}:{pair[1]}] offset:{imm}")
        self.issued += 1

    def drain(self):
        self.emit("s_waitcnt lgkmcnt(0)")
        self.done = self.issued

    def mfma(self, tile, a, b, zero):
        # a tile's first MFMA: C = 0, or this lane's bias of the tile's row block (tile = 4 * (4 rb + pb))
        c = (f"a[{A_BIAS + 4 * (tile // 16)}:{A_BIAS + 4 * (tile // 16) + 3}]" if self.bias else "0") if zero else f"a[{tile}:{tile + 3}]"
        self.emit(f"v_mfma_f32_16x16x32_bf16 a[{tile}:{tile + 3}], {self.reg(a)}, {self.reg(b)}, {c}")


def epilogue_ops(rbs, bias):
    # (the bias is in the accumulators already: a tile's first MFMA takes it as its C operand)
    ops, k = [], 0
    for bi, rb in enumerate(rbs):
        for pb in range(4):
            t = 4 * (4 * rb + pb)
            rd, out = T_RD[0], T_OUT[k & 1]
            for i in range(4):
                ops.append(("v", f"v_accvgpr_read_b32 v{rd[i]}, a{t + i}"))
            ops.append(("v", f"v_cvt_pk_bf16_f32 v{out[0]}, v{rd[0]}, v{rd[1]}"))
            ops.append(("v", f"v_cvt_pk_bf16_f32 v{out[1]}, v{rd[2]}, v{rd[3]}"))
            ops.append(("w", (out, (rb & 3) * 2048 + pb * 256)))
            k += 1
    return ops


def emit_op(st, op):
    kind, payload = op
    if kind == "biasrd":
        st.read_fixed(payload[0], "%[eb]", payload[1], f"bias{payload[0][0]}")
    elif kind == "v":
        st.emit(payload)
    elif kind == "vb":
        st.need(f"bias{payload[1][0]}")
        st.emit(payload[0])
    else:
        st.ds_write(payload[0], payload[1])


def tail_ops(kind):
    """VALU work behind the period's last B read: the pixel-block addresses of the NEXT period (L: of the next tile, out of the mailbox)"""
    if kind == "L":
        # (behind the period's last B read the address registers are free: the mailbox row of this lane -- 4 pixel-block units -- lands in
        #  v[32:35], {there is a next tile, row pitch, bytes of a channel block} in v[36:39]; turned into the next tile's addresses behind the drain)
        return [("v", f"v_readfirstlane_b32 {S_HAS}, v{V_P}"), ("v", f"v_readfirstlane_b32 {S_LW}, v{V_P + 1}"), ("v", f"v_readfirstlane_b32 {S_PERKB}, v{V_P + 2}"),
                ("v", f"v_and_b32 v{T_RD[0][0]}, {S_PERKB}, %[kmask]"), ("v", f"v_add_u32 v{T_RD[0][0]}, v{T_RD[0][0]}, %[kconst]")] + \
               [("v", f"v_add_u32 v{V_PBL + i}, v{V_PBL + i}, v{T_RD[0][0]}") for i in range(4)] + [("v", f"v_mov_b32 v{V_P + i}, v{V_PBL + i}") for i in range(4)]
    return [("v", f"v_add_u32 v{V_P + i}, v{V_PBL + i}, {S_IO}") for i in range(4)]


def gen(kind, bias, par=0, wslot=0):
    group_major = kind != "M"
    st = Stream(par, wslot, bias)
    passes, flat = slots_of(group_major)
    zero_first = kind in ("F", "F0")

    def a_read(s):
        q, kx, rb = flat[s]
        st.a_read(f"a{s % NA}", kx * 8192 + rb * 256)

    def b_read(q, pb):
        st.b_read(("n" if q % 2 == 0 else "c") + str(pb), pb, passes[q][0] * 16)

    if kind == "T":
        for s in range(20, 24):
            for pb in range(4):
                st.mfma(4 * (4 * (s - 16) + pb), f"a{s % NA}", f"c{pb}", False)
        st.emit("s_nop 15")
        st.emit("s_nop 15")
        if "nobarx" not in ABL:
            st.emit("s_barrier")
        for op in ([] if "noepi" in ABL else epilogue_ops([4, 5, 6, 7], bias)):
            emit_op(st, op)
        st.drain()
        st.emit("s_barrier")
        return st.lines
    if kind == "F0":
        for pb in range(4):
            b_read(0, pb)
        for s in range(DA):
            a_read(s)
    else:
        # prefix: slots 20..23 of the previous period = (tap 2, row blocks 4..7) out of registers (ring quads 4..7, the B set in use)
        new_reads = {20: [("b", 0), ("b", 1), ("b", 2), ("b", 3), ("a", 0)], 21: [("a", 1), ("a", 2)], 22: [("a", 3), ("a", 4)], 23: [("a", 5)]}
        for s in range(20, 24):
            rds = list(new_reads[s])
            for pb in range(4):
                st.mfma(4 * (4 * (s - 16) + pb), f"a{s % NA}", f"c{pb}", False)
                take = (len(rds) + (3 - pb)) // (4 - pb) if rds else 0
                for _ in range(take):
                    k, i = rds.pop(0)
                    if k == "b":
                        b_read(0, i)
                    else:
                        a_read(i)
            assert not rds
    epi, epi_from, epi_to = [], None, None
    if "noepi" in ABL:
        pass
    elif kind == "F":
        epi, epi_from, epi_to = epilogue_ops([4, 5, 6, 7], bias), 3, 14
    elif kind == "L":
        epi, epi_from, epi_to = epilogue_ops([0, 1, 2, 3], bias), 10, 19
    ngaps = (epi_to - epi_from + 1) * 4 if epi else 0
    gap_i = 0
    tail = tail_ops(kind)
    last_b_slot = max(s for s in range(20) if flat[s][0] + 1 < len(passes) and
                      ((not group_major and 1 <= s - 8 * flat[s][0] <= 4) or (group_major and s - 4 * flat[s][0] <= 1)))
    for s in range(20):
        q, kx, rb = flat[s]
        pos_in_pass = s - min(i for i, f in enumerate(flat) if f[0] == q)
        bset = "n" if q % 2 == 0 else "c"
        rds = []
        if s + DA < 24:
            rds.append(("a", s + DA))
        if q + 1 < len(passes):   # the B quads of the next pass
            if not group_major:
                if 1 <= pos_in_pass <= 4:
                    rds.append(("b", pos_in_pass - 1))
            else:
                if pos_in_pass == 0:
                    rds += [("b", 0), ("b", 1)]
                elif pos_in_pass == 1:
                    rds += [("b", 2), ("b", 3)]
        if kind in ("F", "F0") and s == 3 and "nobarx" not in ABL:
            st.emit("s_barrier")   # (X: the loaders have taken the staged half of the previous tile into registers)
        if kind == "L" and s == 18:   # the next tile's mailbox (written by the loaders in period 1): rides the DS queue of the period
            st.read_fixed([V_PBL, V_PBL + 1, V_PBL + 2, V_PBL + 3], "%[mba]", 0, "mbl")
            st.read_fixed([V_P, V_P + 1, V_P + 2, V_P + 3], "%[mbs]", 0, "mbs")
        for pb in range(4):
            tile = 4 * (4 * rb + pb)
            st.need(f"a{s % NA}", f"{bset}{pb}")
            st.mfma(tile, f"a{s % NA}", f"{bset}{pb}", zero_first and kx == 0)
            take = (len(rds) + (3 - pb)) // (4 - pb) if rds else 0
            for _ in range(take):
                k, i = rds.pop(0)
                if k == "a":
                    a_read(i)
                else:
                    b_read(q + 1, i)
            if epi and epi_from <= s <= epi_to:
                left_gaps = ngaps - gap_i
                n = (len(epi) + left_gaps - 1) // left_gaps
                for _ in range(min(n, len(epi))):
                    emit_op(st, epi.pop(0))
                gap_i += 1
            if s > last_b_slot and s >= 18 and tail and kind != "L":
                st.emit(tail.pop(0)[1])
        assert not rds and last_b_slot < 18
    assert not epi
    st.drain()
    for op in tail:   # (L: the next tile's addresses out of the mailbox)
        st.emit(op[1])
    st.emit("s_barrier")
    verify(kind, st.lines)
    return st.lines


def verify(kind, lines):
    """an accumulator is read out at least 8 MFMAs after its last MFMA and before the MFMA that starts it again"""
    last_mfma, n_mfma, reads = {}, 0, {}
    for l in lines:
        m = re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):", l)
        if m:
            t = int(m.group(1))
            if l.endswith(", 0") and kind == "F" and "noepi" not in ABL:
                for r in range(t, t + 4):
                    assert t < 64 or r in reads, f"{kind}: a{r} started again before its read-out"
            for r in range(t, t + 4):
                last_mfma[r] = n_mfma
            n_mfma += 1
        m = re.match(r"v_accvgpr_read_b32 v\d+, a(\d+)", l)
        if m:
            r = int(m.group(1))
            assert r not in reads
            reads[r] = n_mfma
            assert n_mfma - last_mfma.get(r, -100) >= 8, f"{kind}: a{r} read {n_mfma - last_mfma[r]} MFMAs after its last MFMA"
    if kind == "F" and "noepi" not in ABL:
        assert sorted(reads) == list(range(64, 128)), kind
    if kind == "L" and "noepi" not in ABL:
        assert sorted(reads) == list(range(0, 64)), kind


def set_io():
    """s_io = (pair & 1) * ISLOT + kernel row * row pitch for the period (s_P, s_R); then (s_P, s_R) step to the following period"""
    return [f"s_and_b32 {S_T0}, {S_P}, 1", f"s_lshl_b32 {S_T0}, {S_T0}, 15", f"s_mul_i32 {S_T1}, {S_R}, {S_LW}", f"s_add_u32 {S_IO}, {S_T0}, {S_T1}",
            f"s_add_u32 {S_R}, {S_R}, 1", f"s_cmp_eq_u32 {S_R}, 3", f"s_cselect_b32 {S_R}, 0, {S_R}", f"s_addc_u32 {S_P}, {S_P}, 0"]


def role(bias):
    L = []
    # entry: the tile's pixel-block addresses, the row pitch; (s_P, s_R) = the period FOLLOWING the one whose text comes next
    L += [f"v_mov_b32 v{V_PBL + i}, %[q{i}]" for i in range(4)] + [f"v_mov_b32 v{V_P + i}, %[q{i}]" for i in range(4)]
    if bias:   # this lane's bias (channels 16 rb + 4 g + 0..3: a float4 per row block at %[eb] + 64 rb) -> a[128:159], once
        for rb in range(8):
            L += [f"ds_read_b128 v[{T_RD[0][0]}:{T_RD[0][3]}], %[eb] offset:{rb * 64}", "s_waitcnt lgkmcnt(0)"]
            L += [f"v_accvgpr_write_b32 a{A_BIAS + 4 * rb + i}, v{T_RD[0][i]}" for i in range(4)]
    L += [f"s_mov_b32 {S_LW}, %[lw16]", f"s_mov_b32 {S_P}, 0", f"s_mov_b32 {S_R}, 1"] + set_io()
    L += gen("F0", bias, 0, 0)
    L += ["s_branch Lw16_mid%="]
    L += ["Lw16_tile%=:", f"s_mov_b32 {S_P}, 0", f"s_mov_b32 {S_R}, 1"] + set_io()
    L += gen("F", bias, 0, 0)
    # periods 1 .. nper - 2 in pairs (odd period: weight slot 1, B set X in use; even: slot 0, set Y)
    L += ["Lw16_mid%=:", f"s_mov_b32 {S_CNT}, %[npp]", "Lw16_m%=:"] + set_io()
    L += gen("M", bias, 0, 1)
    L += set_io()
    L += gen("M", bias, 1, 0)
    L += [f"s_sub_u32 {S_CNT}, {S_CNT}, 1", f"s_cmp_lg_u32 {S_CNT}, 0", "s_cbranch_scc1 Lw16_m%="]
    L += gen("L", bias, 0, 1)
    L += [f"s_cmp_lg_u32 {S_HAS}, 0", "s_cbranch_scc1 Lw16_tile%="]
    L += gen("T", bias, 0, 0)
    return L


def main():
    print("// GENERATED by tools/gen_ws16.py -- do not edit (see the generator for the schedule)")
    for bias in (False, True):
        name = "W16_TXT_ROLE" + ("B" if bias else "")
        lines = role(bias)
        print(f"// {name}: {sum(1 for l in lines if l.startswith('v_mfma'))} MFMAs, {sum(1 for l in lines if l.startswith('ds_'))} DS operations, {len(lines)} lines")
        print(f"#define {name} \\")
        for i, l in enumerate(lines):
            end = "" if i + 1 == len(lines) else " \\"
            print(f'  "{l}\\n\\t"{end}')
        print()


if __name__ == "__main__":
    main()
