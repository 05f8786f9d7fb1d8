#!/usr/bin/env python3
"""Generator of devit_amd/csrc/gemm4_kloop.inc: the hand-scheduled K loop of the FOUR-wave 256x256x64 bf16 GEMM
(csrc/gemm.hip, gemm4_kernel) as one inline-asm string, plus the accumulator read-out helpers.

Why generated asm: a wave that has its SIMD to itself issues about one instruction per four cycles, MFMAs included, so the
64 MFMAs of a phase (1024 cycles of matrix pipe) leave room for ~2 other instructions per MFMA; hipcc's schedule of the
same loop had 83-157 per 16 MFMAs (profiles/r03_y_gemm_four_wave.txt).  Here every non-MFMA instruction of the loop has a
fixed place between two MFMAs.

Design (one workgroup = 4 waves = one per SIMD, 512 registers each; wave w = (wm, wn) = (w >> 1, w & 1) owns the 128 x 128
sub-tile rows 128 wm.., columns 128 wn.. of a 256 x 256 output tile):
  * accumulators: a[0:255], tile (i, q) = m-tile i (16 rows), n-tile q (16 columns) at a[4 (8 i + q) : +4]; the weight operand
    is on the MFMA's row side, as in the eight-wave kernel, so lane (g, c) register r holds C[m = 16 i + c][n = ncol(q, 4 g + r)]
  * fragments: two buffers of (8 A + 8 B) x 4 VGPRs: buf0 v[128:191] (k-half 0), buf1 v[192:255] (k-half 1)
  * LDS ring as the eight-wave kernel: three A slots (32 KB each, at 0) and two B slots (at 96 KB), lane-linear images
    with the bank swizzle on the DMA source address / the fragment read address (swz_row in gemm.hip)
  * one K-step (stage t) = two phases of 64 MFMAs:
      phase 1: MFMAs on buf0 = (t, kk 0) | ds_read (t, kk 1) -> buf1 | LDS-DMA request A(t + 2)
      middle : s_waitcnt vmcnt(8) (everything but that A request has landed: stage t + 1 is in LDS), lgkmcnt(0), s_barrier
      phase 2: MFMAs on buf1 = (t, kk 1) | ds_read (t + 1, kk 0) -> buf0 | LDS-DMA request B(t + 2) (into B(t)'s slot)
    RAW: a stage is read after the barrier that follows every wave's counted wait for its share of it.  WAR: A(t + 2) overwrites
    A(t - 1), last read in phase 1 of step t - 1 (retired by the lgkmcnt(0) + barrier in the middle of that step); B(t + 2)
    overwrites B(t), last read in phase 1 of step t, retired by this step's middle.
  * placement inside a phase (64 MFMA gaps, gap n = behind MFMA n): one fragment read per three gaps (0, 3, ..., 45: every read
    is back long before the wait that ends the phase); the eight requests of the phase at gaps 6 k + o(w), o = 1, 2, 4, 5 for
    waves 0..3 -- a request costs the issuing wave ~60-180 cycles when all four waves issue theirs in the same gap (measured with
    the first version: a K-step took 2730 cycles against 2048 of MFMA), the CU's address path takes one request per ~16 cycles --
    which makes FOUR copies of the loop, one per wave, selected by a scalar branch at entry; M0 is written one gap ahead of its
    request (the MFMA between them is the wait state); pointer / slot bookkeeping and the next phase's read addresses in the gaps
    from 48 on, two per gap
  * the request stream runs across tile boundaries: stage index nk, nk + 1 are stages 0, 1 of the workgroup's NEXT tile
    (pointers a_next / b_next; the last tile passes its own pointers again: two harmless extra stages into free slots)
  * at entry the statement reads (0, kk 0) itself (fragment registers do not survive the compiler's epilogue code between two
    tiles) and at exit it leaves the accumulators in a[0:255] for the epilogue (gemm4_read_acc below)

Operands of the asm statement (named; declared in gemm4_kernel):
  [g3] +s LDS byte offset of the A slot of this tile's stage 0 (0 / 32768 / 65536), [g2] +s of the B slot (98304 / 131072)
  [t0]..[t3] =&v scratch (current read addresses)   [wv] s wave index 0..3
  [aptr] s 64-bit &A[m0][k0], [bptr], [anext], [bnext] (the workgroup's next tile; its own pointers again on the last tile)
  [nk] s K-steps (>= 3)   [lda64] s 64 * lda bytes (32 rows)   [ldb64]   [wlds] s LDS byte address of the ring + wave * 8192
  [dsa0..3] v fragment-read byte offsets of A inside a slot for (kk 0, parity 0), (kk 0, 1), (kk 1, 0), (kk 1, 1)   [dsb0..3]
  [dmaa0..3] v per-lane source byte offsets of slabs 0..3 (slabs 4..7 = + lda64)   [dmab0..3]
"""
import os
import sys

A_SLOT = 32768
B_BASE = 3 * A_SLOT

# fixed registers (all listed as clobbers of the statement)
FA = [128, 192]           # A fragments of buffer 0 / 1: v[FA + 4 i : +4]
FB = [160, 224]           # B fragments: v[FB + 4 q : +4]
VT = ["%[t0]", "%[t1]", "%[t2]", "%[t3]"]   # current read addresses (A parity 0, A parity 1, B parity 0, B parity 1): scratch operands
S = dict(pa=70, pa_hi=72, pb=74, pb_hi=76, ia=78, ib=79, t=80, m0save=81, dstA=82, dstB=83, tmp=84, a0=85, a1=86, a2=87,
         b0=88, b1=89, tend=90)
DMA_OFF = [1, 2, 4, 5]    # request gaps of waves 0..3: 6 k + DMA_OFF[w]
TAIL0 = 48                # first gap of the bookkeeping tail

FIXSRC = os.environ.get("GEMM4_FIXSRC") == "1"     # experiment: every request re-reads stage 2 of the tile (L1 / L2 hits): is the loop issue- or memory-bound?
NODMA = os.environ.get("GEMM4_NODMA") == "1"       # experiment: no requests at all (s_nop in their place)
A_NT = os.environ.get("GEMM4_A_NT") == "1"         # experiment: non-temporal requests for the activation operand (round 5: what the full-row kernel gains from)
STAMP = False      # diagnostic variant: s_memtime deltas of the loop's segments summed in SGPRs, returned in [d0]..[d3]
T0, T1, T2, TN, D1, D2, D3 = 92, 94, 96, 98, 100, 101, 91


def stamp(e, reg):
    e(f"s_memtime s[{reg}:{reg + 1}]")
    e("s_waitcnt lgkmcnt(0)")


def acc(i, q):
    return 4 * (8 * i + q)


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)


def mfma(e, buf, i, q, zero):
    a = acc(i, q)
    c = "0" if zero else f"a[{a}:{a + 3}]"
    e(f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{FB[buf] + 4 * q}:{FB[buf] + 4 * q + 3}], v[{FA[buf] + 4 * i}:{FA[buf] + 4 * i + 3}], {c}")


def read_list(buf):
    """the 16 fragment reads into buffer `buf`: B tiles first (every MFMA row needs all of them), then A tiles in use order"""
    out = []
    for q in range(8):
        out.append(f"ds_read_b128 v[{FB[buf] + 4 * q}:{FB[buf] + 4 * q + 3}], {VT[2 + (q & 1)]} offset:{4096 * (q >> 1)}")
    for i in range(8):
        out.append(f"ds_read_b128 v[{FA[buf] + 4 * i}:{FA[buf] + 4 * i + 3}], {VT[i & 1]} offset:{4096 * (i >> 1)}")
    return out


def addr_setup(kk, a_off_sreg, b_off_sreg):
    """the four scratch operands = read addresses of k-half kk in the slots whose byte offsets are in the two SGPRs"""
    return [f"v_add_u32 {VT[0]}, s{a_off_sreg}, %[dsa{2 * kk}]", f"v_add_u32 {VT[1]}, s{a_off_sreg}, %[dsa{2 * kk + 1}]",
            f"v_add_u32 {VT[2]}, s{b_off_sreg}, %[dsb{2 * kk}]", f"v_add_u32 {VT[3]}, s{b_off_sreg}, %[dsb{2 * kk + 1}]"]


def dma_list(which):
    """the 8 LDS-DMA requests of one operand stage: [(M0 write, request), ...]"""
    out = []
    dst = S["dstA"] if which == "A" else S["dstB"]
    lo = S["pa"] if which == "A" else S["pb"]
    hi = S["pa_hi"] if which == "A" else S["pb_hi"]
    op = "dmaa" if which == "A" else "dmab"
    for i in range(8):
        base = lo if i < 4 else hi
        out.append((f"s_add_u32 m0, s{dst}, {i * 1024}", "s_nop 0" if NODMA else f"global_load_lds_dwordx4 %[{op}{i & 3}], s[{base}:{base + 1}]" +
                    (" nt" if A_NT and which == "A" else "")))
    return out


def advance_ptr(which):
    """after a stage of `which` was requested: index + 1, pointer + one K-step (128 bytes) or, when the index reaches nk, the next
    tile's pointer; then the pointer of slabs 4..7.  (The tail deals its list two per gap, in order: an instruction that reads
    SCC stays in the gap of the one that sets it, or in the next -- MFMAs do not touch SCC.)"""
    lo = S["pa"] if which == "A" else S["pb"]
    hi = S["pa_hi"] if which == "A" else S["pb_hi"]
    idx = S["ia"] if which == "A" else S["ib"]
    nxt = "[anext]" if which == "A" else "[bnext]"
    ld64 = "[lda64]" if which == "A" else "[ldb64]"
    if FIXSRC or NODMA:
        return ["s_nop 0"] * 8
    return [f"s_add_u32 s{lo}, s{lo}, 128", f"s_addc_u32 s{lo + 1}, s{lo + 1}, 0",
            f"s_add_u32 s{idx}, s{idx}, 1", f"s_cmp_eq_u32 s{idx}, %[nk]",
            f"s_cselect_b64 s[{lo}:{lo + 1}], %{nxt}, s[{lo}:{lo + 1}]", "s_nop 0",
            f"s_add_u32 s{hi}, s{lo}, %{ld64}", f"s_addc_u32 s{hi + 1}, s{lo + 1}, 0"]


def rotate_slots():
    """stage t -> t + 1: (a0, a1, a2) = LDS offsets of the A slots of stages (t, t + 1, t + 2) rotate, (b0, b1) swap; request slots"""
    t = S["tmp"]
    return [f"s_mov_b32 s{t}, s{S['a0']}", f"s_mov_b32 s{S['a0']}, s{S['a1']}", f"s_mov_b32 s{S['a1']}, s{S['a2']}",
            f"s_mov_b32 s{S['a2']}, s{t}", f"s_add_u32 s{S['dstA']}, s{S['a2']}, %[wlds]",
            f"s_mov_b32 s{t}, s{S['b0']}", f"s_mov_b32 s{S['b0']}, s{S['b1']}", f"s_mov_b32 s{S['b1']}, s{t}",
            f"s_add_u32 s{S['dstB']}, s{S['b0']}, %[wlds]"]


def init_slots(e):
    """from [g3] / [g2] (slots of stage 0): a0, a1, a2, b0, b1 and the request slots of A(2) / B(2)"""
    e(f"s_mov_b32 s{S['a0']}, %[g3]")
    for src, dst in (("a0", "a1"), ("a1", "a2")):
        e(f"s_add_u32 s{S[dst]}, s{S[src]}, {A_SLOT}")
        e(f"s_cmp_eq_u32 s{S[dst]}, {B_BASE}")
        e(f"s_cselect_b32 s{S[dst]}, 0, s{S[dst]}")
    e(f"s_mov_b32 s{S['b0']}, %[g2]")
    e(f"s_xor_b32 s{S['b1']}, s{S['b0']}, {B_BASE ^ (B_BASE + A_SLOT)}")
    e(f"s_add_u32 s{S['dstA']}, s{S['a2']}, %[wlds]")
    e(f"s_add_u32 s{S['dstB']}, s{S['b0']}, %[wlds]")


def phase(e, wave, buf, zero, reads, dmas, tail):
    """64 MFMAs on buffer `buf` with the other instructions dealt into the gaps (gap n = behind MFMA n): fragment read k in gap
    3 k; request k in gap 6 k + DMA_OFF[wave], its M0 write one gap earlier; `tail` two per gap from TAIL0 on"""
    dgap = {6 * k + DMA_OFF[wave]: k for k in range(8)} if dmas else {}
    assert not dgap or (max(dgap) < TAIL0 and min(dgap) >= 1)
    assert len(tail) <= 2 * (64 - TAIL0), len(tail)
    n = 0
    for i in range(8):
        for q in range(8):
            mfma(e, buf, i, q, zero)
            if reads and n % 3 == 0 and n // 3 < 16:
                e(reads[n // 3])
            if n + 1 in dgap:
                e(dmas[dgap[n + 1]][0])
            if n in dgap:
                e(dmas[dgap[n]][1])
            if n >= TAIL0:
                for s in tail[2 * (n - TAIL0): 2 * (n - TAIL0) + 2]:
                    e(s)
            n += 1


def step(e, wave, kind):
    """kind: 'first' (accumulators start from zero), 'mid', 'last' (no fragment reads for a following step)"""
    # phase 1 reads (t, kk 1) through addresses set up in the previous step's phase-2 tail (or at entry); its own tail sets up
    # the addresses of (t + 1, kk 0) for phase 2
    tail1 = advance_ptr("A") + (addr_setup(0, S["a1"], S["b1"]) if kind != "last" else [])
    phase(e, wave, 0, kind == "first", read_list(1), dma_list("A"), tail1)
    if STAMP:
        e(f"s_memtime s[{T1}:{T1 + 1}]")          # phase 1 issued (the lgkmcnt(0) below covers the stamp)
    e("s_waitcnt vmcnt(0)" if NODMA else "s_waitcnt vmcnt(8)")
    e("s_waitcnt lgkmcnt(0)")
    e("s_barrier")
    if STAMP:
        stamp(e, T2)                              # through the middle
    tail2 = advance_ptr("B") + rotate_slots()
    if kind != "last":
        tail2 += addr_setup(1, S["a0"], S["b0"])  # (after the rotation a0 / b0 are the next stage's slots)
    phase(e, wave, 1, False, read_list(0) if kind != "last" else None, dma_list("B"), tail2)
    if kind != "last":
        e("s_waitcnt lgkmcnt(0)")
    if STAMP:
        stamp(e, TN)                              # step over: d1 += T1 - T0, d2 += T2 - T1, d3 += TN - T2, T0 = TN
        e(f"s_sub_u32 s{T1 + 1}, s{T1}, s{T0}")
        e(f"s_add_u32 s{D1}, s{D1}, s{T1 + 1}")
        e(f"s_sub_u32 s{T2 + 1}, s{T2}, s{T1}")
        e(f"s_add_u32 s{D2}, s{D2}, s{T2 + 1}")
        e(f"s_sub_u32 s{TN + 1}, s{TN}, s{T2}")
        e(f"s_add_u32 s{D3}, s{D3}, s{TN + 1}")
        e(f"s_mov_b32 s{T0}, s{TN}")


def kloop():
    e = Emit()
    e(f"s_mov_b32 s{S['m0save']}, m0")
    if STAMP:
        stamp(e, T1)                              # entry
        e(f"s_mov_b32 s{D1}, 0")
        e(f"s_mov_b32 s{D2}, 0")
        e(f"s_mov_b32 s{D3}, 0")
    # request pointers start at stage 2 (stages 0 and 1 of this tile were requested by the previous tile's last two steps, or by
    # the kernel's prologue)
    e(f"s_mov_b64 s[{S['pa']}:{S['pa'] + 1}], %[aptr]")
    e(f"s_mov_b64 s[{S['pb']}:{S['pb'] + 1}], %[bptr]")
    e(f"s_add_u32 s{S['pa']}, s{S['pa']}, 256")
    e(f"s_addc_u32 s{S['pa'] + 1}, s{S['pa'] + 1}, 0")
    e(f"s_add_u32 s{S['pb']}, s{S['pb']}, 256")
    e(f"s_addc_u32 s{S['pb'] + 1}, s{S['pb'] + 1}, 0")
    e(f"s_add_u32 s{S['pa_hi']}, s{S['pa']}, %[lda64]")
    e(f"s_addc_u32 s{S['pa_hi'] + 1}, s{S['pa'] + 1}, 0")
    e(f"s_add_u32 s{S['pb_hi']}, s{S['pb']}, %[ldb64]")
    e(f"s_addc_u32 s{S['pb_hi'] + 1}, s{S['pb'] + 1}, 0")
    e(f"s_mov_b32 s{S['ia']}, 2")
    e(f"s_mov_b32 s{S['ib']}, 2")
    e(f"s_sub_u32 s{S['tend']}, %[nk], 1")                    # the last step's index
    init_slots(e)
    # fragments (0, kk 0) of this tile: stage 0 has landed and every wave knows (previous tile's last middle barrier / prologue)
    for s in addr_setup(0, S["a0"], S["b0"]):
        e(s)
    for s in read_list(0):
        e(s)
    for s in addr_setup(1, S["a0"], S["b0"]):              # phase 1 of step 0 reads (0, kk 1) (the reads above are issued:
        e(s)                                               # their address registers may be rewritten)
    e("s_waitcnt lgkmcnt(0)")
    if STAMP:
        stamp(e, T0)                              # first fragments in registers: d0 = T0 - entry
        e(f"s_sub_u32 %[d0], s{T0}, s{T1}")
    # one copy of the loop per wave (request gaps differ), selected here
    for w in range(1, 4):
        e(f"s_cmp_eq_u32 %[wv], {w}")
        e(f"s_cbranch_scc1 L_gemm4_w{w}_%=")
    for w in range(4):
        e(f"L_gemm4_w{w}_%=:")
        step(e, w, "first")
        e(f"s_mov_b32 s{S['t']}, 1")
        e(f"L_gemm4_loop{w}_%=:")
        step(e, w, "mid")
        e(f"s_add_u32 s{S['t']}, s{S['t']}, 1")
        e(f"s_cmp_lt_u32 s{S['t']}, s{S['tend']}")
        e(f"s_cbranch_scc1 L_gemm4_loop{w}_%=")
        step(e, w, "last")
        if w < 3:
            e("s_branch L_gemm4_done_%=")
    e("L_gemm4_done_%=:")
    # slots of the next tile's stage 0 (the rotation of the last step has run)
    e(f"s_mov_b32 %[g3], s{S['a0']}")
    e(f"s_mov_b32 %[g2], s{S['b0']}")
    e(f"s_mov_b32 m0, s{S['m0save']}")
    if STAMP:
        e(f"s_mov_b32 %[d1], s{D1}")
        e(f"s_mov_b32 %[d2], s{D2}")
        e(f"s_mov_b32 %[d3], s{D3}")
    e("s_nop 15")          # MFMA results -> v_accvgpr_read of the epilogue (the assembler pads nothing inside or after asm)
    e("s_nop 15")
    return e.lines


def clobbers():
    c = [f"v{r}" for r in range(128, 256)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(70, 102 if STAMP else 91)]
    return c + ["memory", "scc"]


def main(out):
    global STAMP
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm4.py -- do not edit (tests/test_abi.py::test_gemm4_inc_is_current regenerates and compares).\n")
        f.write("// The K loop of the four-wave 256x256x64 GEMM as one inline-asm statement; operands and register plan: the generator's docstring.\n")
        for name, flag in (("DEVIT_GEMM4_KLOOP", False), ("DEVIT_GEMM4_KLOOP_STAMPED", True)):
            STAMP = flag
            lines = kloop()
            n_mfma = sum(1 for s in lines if s.startswith("v_mfma"))
            assert n_mfma == 4 * 3 * 128, n_mfma
            if flag:
                f.write("// diagnostic variant (-DDEVIT_GEMM4_STAMP): s_memtime deltas of the loop's segments, operands [d0]..[d3] in addition\n")
            f.write(f"#define {name}_ASM \\\n")
            for s in lines:
                f.write(f'  "{s}\\n\\t" \\\n')
            f.write('  ""\n')
            f.write(f"#define {name}_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n\n")
        STAMP = False
        # accumulator read-out: n-half h (64 columns), m-tiles i0, i0 + 1 -> f32x4 acc[2][4]
        f.write("// accumulators of m-tiles I0, I0 + 1 and the four n-tiles of column half H (tile q = 4 H + j) out of a[0:255]\n")
        f.write("template <int H, int I0>\n__device__ __forceinline__ void gemm4_read_acc(f32x4 (&acc)[2][4]) {\n")
        f.write("  static_assert(H >= 0 && H < 2 && I0 >= 0 && I0 < 8 && I0 % 2 == 0, \"\");\n")
        for h in range(2):
            for i0 in range(0, 8, 2):
                f.write(f"  if constexpr (H == {h} && I0 == {i0}) {{\n")
                for u in range(2):
                    for j in range(4):
                        a = acc(i0 + u, 4 * h + j)
                        f.write(f"    {{ float x0, x1, x2, x3; asm volatile(\"v_accvgpr_read_b32 %0, a{a}\\n\\tv_accvgpr_read_b32 %1, a{a + 1}\\n\\t"
                                f"v_accvgpr_read_b32 %2, a{a + 2}\\n\\tv_accvgpr_read_b32 %3, a{a + 3}\" : \"=v\"(x0), \"=v\"(x1), \"=v\"(x2), \"=v\"(x3) :: \"memory\"); "
                                f"acc[{u}][{j}] = (f32x4){{x0, x1, x2, x3}}; }}\n")
                f.write("  }\n")
        f.write("}\n")
    print(f"wrote {out}: {len(lines)} asm lines in the stamped variant, {n_mfma} MFMAs")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "devit_amd", "csrc", "gemm4_kloop.inc"))
