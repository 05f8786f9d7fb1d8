#!/usr/bin/env python3
"""Generator of devit_amd/csrc/gemmfr_kloop.inc: the hand-scheduled K loop of the FULL-ROW 256 x 384 x 64 bf16 GEMM
(csrc/gemm.hip, gemmfr_kernel; activation operand row-major, weight operand k-major, N = 384 = the student's whole output row)
as one inline-asm string, plus the accumulator read-out helper.

Why this tile (verdict r04 #1): every K loop on this chip runs at the rate a CU's LDS-DMA requests are served (DESIGN.md section
4.1a).  Two 128 x 128 workgroups per CU need 64 B per cycle of matrix pipe, a 256 x 256 tile 32, this one (256 + 384) x 128 B per
3072 cycles = 26.7 -- and it reads the activation panel once instead of once per n-tile.

Design (one workgroup = 4 waves = one per SIMD, 512 registers each; wave w = (wm, wn) = (w >> 1, w & 1) owns rows 128 wm..,
columns 192 wn.. of the tile: 8 m-tiles x 12 n-tiles of 16 x 16):
  * accumulators: 96 tiles = 384 registers.  n-tiles q < 8 in a[4 (8 i + q) : +4] (the four-wave kernel's plan), n-tiles 8..11 in
    v[128 + 4 (4 i + q - 8) : +4] -- declared to the compiler as four pinned 32-register outputs ([c0]..[c3] = v[128:159], ...),
    so the epilogue reads them as ordinary values.  The weight operand is on the MFMA's row side (lane (g, c), register r holds
    C[m = 16 i + c][n = ncol(q, 4 g + r)]), as in every kernel of gemm.hip: same epilogue code, same accumulation order per element.
  * fragments: ONE buffer, A tiles v[48 + 4 i : +4] (8), B tiles v[80 + 4 q : +4] (12).  The 96 MFMAs of a phase (one k-half of 32)
    run row by row (i outer, q inner); A[i] is dead after MFMA (i, 11) and B[q] after MFMA (7, q), so the next phase's A[i] is read
    right behind row i and its B tiles (two ds_read_b64_tr_b16 each) behind the MFMAs of the last row.  LDS returns in order, so every
    use is guarded by a counted lgkmcnt: before MFMA (0, q) all but the reads issued after B[q]'s, before MFMA (7, 0) all but the seven
    A reads of the running phase.
  * LDS ring: two A slots (32 KB: [256 rows][64 k], at 0 / 32768) + two B slots (48 KB: [64 k][384 cols], at 65536 / 114688) = the CU's
    160 KB; lane-linear images with the bank swizzle on the DMA source address / the fragment read address (swz_row / swz_krow).
  * A request blocks the issuing wave until the CU's one address path has taken it (~30 cycles each with the operands coming from L2 /
    HBM, tools/fill_probe2.hip; 80 requests per K-step): they must be SPREAD over the K-step, and two slots of whole stages release
    nothing between the two k-halves of a stage (round 5's first version: 6200 cycles per K-step against 3072 of MFMA).  So the B
    stages are SHIFTED by half a stage: B stage u covers k rows [64 u - 32, 64 u + 32) (free for a k-major operand: whole 768-byte
    rows either way), cyclic in K (stage 0's first half = the last 32 k rows: never read; stage nk = stage 0 of the next tile, the
    same weight).  With p = the k-half index: phase p multiplies A (stage p >> 1, half p & 1) by B (stage (p + 1) >> 1, half (p + 1) & 1);
    an A slot is read out at the end of an EVEN phase, a B slot at the end of an ODD one:
      even phase 2 s : barrier: vmcnt -> B (s + 1) landed | MFMAs | reads A (s, 1), B (s + 1, 0) | requests B (s + 2) -> B (s)'s slot
      odd phase 2 s+1: barrier: vmcnt -> A (s + 1) landed | MFMAs | reads A (s + 1, 0), B (s + 1, 1) | requests A (s + 2) -> A (s)'s slot
    Every phase: behind its first MFMA row the counted vmcnt (the requests of the phase before may stay in flight: two phases of
    look-ahead), lgkmcnt(0), s_barrier.  RAW: a stage is read after the barrier that follows every wave's wait for its share.  WAR: a
    slot is requested into after the barrier that follows every wave's lgkmcnt(0) over its last reads of it.
  * requests: wave w issues its k-th request of a phase in gap 13 + STRIDE k + w, M0 written one gap ahead; four copies of the loop
    (request gaps and read offsets differ per wave), selected at entry.
  * the request stream runs across tile boundaries: A (nk), A (nk + 1) = stages 0, 1 of the workgroup's NEXT tile ([anext]; only with
    [hasnext]), B is the same cyclic stream for every tile (N = 384: one n-tile).
  * at entry the statement reads the fragments of phase 0 itself; at exit the accumulators stay in a[0:255] / [c0]..[c3].

Operands of the asm statement (named; declared in gemmfr_kernel):
  [c0]..[c3] =&{v[128:159]} ... f32x32   [acur] +s LDS byte offset of the A slot of this tile's stage 0 (0 / 32768)
  [t0]..[t9] =&v scratch (read addresses; the PAIRED variant uses t0..t5)   [wv] s wave index   [hasnext] s
  [aptr] s 64-bit &A[m0][k0], [anext]; [bplo], [bphi] s the two halves of &B[0][n0]
  [bv2] s byte offset of this wave's 16 k rows of B stage 2 (k row 96 + 16 wave)
  [kb] s K * ldb * 2 (the cyclic stream's period in bytes), [ldbs] s 128 * ldb (64 k rows)
  [nk] s K-steps (>= 3)   [lda64] s 64 * lda bytes (32 rows)
  [wldsa] s LDS address of the ring + wave * 8192, [wldsb] + wave * 12288 (the B slots' offsets come on top)
  [dsa0..3] v fragment-read byte offsets of A inside a slot for (kk 0, parity 0), (kk 0, 1), (kk 1, 0), (kk 1, 1)
  [dsb0..7] v of B, both halves (the half is an immediate): PAIRED variant dsb0..3 by (chunk group & 3); NATURAL variant dsb0..7 by
            2 (chunk group & 3) + (n-tile & 1) -- see b_reads()
  [dmaa0..3] v per-lane source byte offsets of A slabs 0..3 (4..7 = + lda64)
  [dmab0..11] v of this wave's twelve B slabs, relative to its first k row (16 k rows per wave)

The WEIGHT-GRADIENT variant (round 6: DEVIT_WGRADFR_KLOOP_ASM, wgradfr_kernel in gemm.hip; KMA = True below): the A operand is K-MAJOR as
well (dW = dY^T X: both operands are [K token rows][features]), one tile and one K slice per workgroup (no next tile), column sums of A
on the side.  What changes against the PAIRED variant above, and nothing else does (same ring protocol, phases, waits, request gaps):
  * A slot image [64 k][256 cols] (512-byte rows, swz_krow); the wave requests its 16 k rows as eight 1-KiB slabs of two k rows
    ([dmaa0..7], one pointer, [ldas] = 64 lda 2 bytes per stage);
  * A fragments by two ds_read_b64_tr_b16 per tile, PAIRED column order on the A side too (tile i, MFMA column c = tile row
    32 (i >> 1) + 8 (c >> 2) + 4 (i & 1) + (c & 3) of the wave's 128): with x = 4 wm + (i >> 1) the address inside the slot is
    512 krow + 64 (x ^ q4) + 16 (p ^ 2 (G & 1)) + 8 (i & 1) -> FOUR base registers [t0..t3] by i >> 1 ([dsa0..3]); the k-half (16384), i & 1
    and the second read (2048) are immediates.  B reads use [t4..t7] / [dsb0..3];
  * two reads per A tile: the counted lgkmcnt in front of MFMA (7, 0) leaves 14 reads in flight, pending_after_b() counts A[7] twice;
  * column sums of A (= the bias gradient when A = dY): wave (wm, wn) adds up its A tiles with (i & 1) == wn -- four v_dot2c_f32_bf16 of
    the fragment registers against [ones] (0x3f803f80, or 0: no sums wanted) into [rs0..3], behind the first MFMAs of row i (the fragment
    has landed, its re-read is issued behind the row's last MFMA).  Lane (G, c) ends with the sum over ITS k of tile row c: the kernel
    folds the four G.
"""
import os
import sys

A_SLOT = 32768
B_BASE = 2 * A_SLOT
B_SLOT = 49152
NI, NQ = 8, 12              # m-tiles, n-tiles per wave
NM = NI * NQ                # MFMAs per phase
FA, FB, VACC = 48, 80, 128  # first registers of the A / B fragments and of the VGPR-resident accumulators
BAR_GAP = NQ - 1            # a phase's barrier sits behind MFMA (0, 11)
S = dict(pa=70, pa1=72, pb=74, bv=76, t=80, m0save=81, dstA=82, dstB=83, tmp=84, acur=85, anxt=86, bcur=87, bnxt=88, tend=89)
S_LAST = 90
STAMP = False
ST = dict(t0=92, t1=94, d1=96, d2=97)     # stamped variant: s_memtime scratch pairs, sums
NOREQ = os.environ.get("GEMMFR_NOREQ") == "1"      # experiment: no requests at all (wrong results: timing of the MFMA / read stream alone)
FIXSRC = os.environ.get("GEMMFR_FIXSRC", "")       # experiment: "ab" / "a" / "b": those operands' requests re-read stage 2 (L1 / L2 hits): the issue or the memory system?
A_NT = os.environ.get("GEMMFR_A_NT", "1") == "1"   # non-temporal A requests (GEMMFR_A_NT=0: default cache policy, the ablation).  The activation panel is read ONCE per
                                                   # launch by this kernel; with the default policy its lines push the weight (which every CU re-reads) out of the
                                                   # XCD's L2: K-step 4932-5478 cycles from cold HBM against 3713 with nt = the no-request stream's 3698 (stamped builds)


def acc(i, q):
    return f"a[{4 * (8 * i + q)}:{4 * (8 * i + q) + 3}]" if q < 8 else f"v[{VACC + 4 * (4 * i + q - 8)}:{VACC + 4 * (4 * i + q - 8) + 3}]"


def fa(i):
    return f"v[{FA + 4 * i}:{FA + 4 * i + 3}]"


def fb(q, half=None):
    if half is None:
        return f"v[{FB + 4 * q}:{FB + 4 * q + 3}]"
    return f"v[{FB + 4 * q + 2 * half}:{FB + 4 * q + 2 * half + 1}]"


def a_read(i, kk=0):
    """fragment of A tile i for k-half kk -> list of LDS reads (row-major A: the k-half is in the address registers; k-major: an immediate)"""
    if KMA:
        off = 16384 * kk + 8 * (i & 1)
        return [f"ds_read_b64_tr_b16 v[{FA + 4 * i}:{FA + 4 * i + 1}], %[t{i >> 1}] offset:{off}",
                f"ds_read_b64_tr_b16 v[{FA + 4 * i + 2}:{FA + 4 * i + 3}], %[t{i >> 1}] offset:{off + 2048}"]
    return [f"ds_read_b128 {fa(i)}, %[t{i & 1}] offset:{4096 * (i >> 1)}"]


def a_reads_per_tile():
    return 2 if KMA else 1


PAIRED = True      # column order of the n-tiles: tile_row<PAIRED>() in gemm.hip (bf16 store: PAIRED, fp32 residual: natural)
KMA = False        # the weight-gradient variant: A k-major too, column sums of A, no next tile (see the docstring)
TB = 2             # first [t] register of the B read addresses (KMA: 4 -- A takes t0..t3)


def b_reads(q, half, wave):
    """the two transposed reads of n-tile q, k rows 32 half .. + 31 of the slot (addresses: read_frag<true, W = 384, PAIRED> in gemm.hip).
    Lane (G, q4, p) reads k row 8 G + q4 (+ 32 half, + 4 for the second read); with x = 6 wn + q / 2 the 64-byte chunk group of the
    n-tile pair in the 768-byte k row, its address inside the slot is
      PAIRED : 768 krow + 64 (x ^ q4) + 16 (p ^ 2 (G & 1)) + 8 (q & 1)                        -> base register by x & 3
      natural: 768 krow + 64 (x ^ q4) + 32 ((q & 1) ^ (G & 1)) + 16 (p >> 1) + 8 (p & 1)      -> base register by (x & 3, q & 1)
    and 64 (x & ~3), the half and the second read are immediates."""
    x = 6 * (wave & 1) + (q >> 1)
    off = half * 32 * 768 + 64 * (x & ~3)
    if PAIRED:
        t, off = TB + (x & 3), off + 8 * (q & 1)
    else:
        t = TB + 2 * (x & 3) + (q & 1)
    return [f"ds_read_b64_tr_b16 {fb(q, 0)}, %[t{t}] offset:{off}", f"ds_read_b64_tr_b16 {fb(q, 1)}, %[t{t}] offset:{off + 4 * 768}"]


def addr_setup(kk, a_sreg, b_sreg):
    if KMA:
        return ([f"v_add_u32 %[t{x}], s{a_sreg}, %[dsa{x}]" for x in range(4)] +
                [f"v_add_u32 %[t{TB + x}], s{b_sreg}, %[dsb{x}]" for x in range(4)])
    return ([f"v_add_u32 %[t0], s{a_sreg}, %[dsa{2 * kk}]", f"v_add_u32 %[t1], s{a_sreg}, %[dsa{2 * kk + 1}]"] +
            [f"v_add_u32 %[t{2 + x}], s{b_sreg}, %[dsb{x}]" for x in range(4 if PAIRED else 8)])


def requests_a():
    out = []
    if KMA:
        return [(f"s_add_u32 m0, s{S['dstA']}, {i * 1024}", f"global_load_lds_dwordx4 %[dmaa{i}], s[{S['pa']}:{S['pa'] + 1}]" + (" nt" if A_NT else ""))
                for i in range(8)]
    for i in range(8):
        base = S["pa"] if i < 4 else S["pa1"]
        out.append((f"s_add_u32 m0, s{S['dstA']}, {i * 1024}", f"global_load_lds_dwordx4 %[dmaa{i & 3}], s[{base}:{base + 1}]" + (" nt" if A_NT else "")))
    return out


def requests_b():
    return [(f"s_add_u32 m0, s{S['dstB']}, {i * 1024}", f"global_load_lds_dwordx4 %[dmab{i}], s[{S['pb']}:{S['pb'] + 1}]") for i in range(12)]


def derive_pa1():
    return [f"s_add_u32 s{S['pa1']}, s{S['pa']}, %[lda64]", f"s_addc_u32 s{S['pa1'] + 1}, s{S['pa'] + 1}, 0"]


def advance_a():
    """after an A stage was requested: 128 bytes along the rows.  Units of (s_add, s_addc): a unit never straddles an M0 write (SCC)."""
    if KMA:      # 64 k rows further
        return [[f"s_add_u32 s{S['pa']}, s{S['pa']}, %[ldas]", f"s_addc_u32 s{S['pa'] + 1}, s{S['pa'] + 1}, 0"]]
    return [[f"s_add_u32 s{S['pa']}, s{S['pa']}, 128", f"s_addc_u32 s{S['pa'] + 1}, s{S['pa'] + 1}, 0"], derive_pa1()]


def b_pointer():
    return [f"s_add_u32 s{S['pb']}, %[bplo], s{S['bv']}", f"s_addc_u32 s{S['pb'] + 1}, %[bphi], 0"]


def advance_b():
    """after a B stage was requested: 64 k rows further, cyclic in K; then the 64-bit pointer of the next stage"""
    return [[f"s_add_u32 s{S['bv']}, s{S['bv']}, %[ldbs]", f"s_cmp_ge_u32 s{S['bv']}, %[kb]", f"s_cselect_b32 s{S['tmp']}, %[kb], 0",
             f"s_sub_u32 s{S['bv']}, s{S['bv']}, s{S['tmp']}"], b_pointer()]


def swap(a, b):
    t = S["tmp"]
    return [f"s_mov_b32 s{t}, s{S[a]}", f"s_mov_b32 s{S[a]}, s{S[b]}", f"s_mov_b32 s{S[b]}, s{t}"]


def pending_after_b(q):
    """LDS reads issued after the last read of B[q] in the phase before: the later B tiles (two reads each) and A[7]"""
    return 2 * (NQ - 1 - q) + a_reads_per_tile()


def phase(e, wave, zero, reads, reqs, vmcnt, units_head, units_tail, akk=0):
    """96 MFMAs of one k-half, row by row; everything else dealt into the gaps (gap n = behind MFMA n).
    reads: None or the half of the B slot the next phase's B fragments come from.  vmcnt: the count of the barrier's wait.
    units_head: units for the gaps from 0 on (read addresses, request destination), units_tail: for the gaps from 84 on (pointer /
    slot bookkeeping), one unit per gap.  akk: the k-half of the A fragments read for the NEXT phase (an immediate of the k-major reads)."""
    rgap = {}
    if reqs and not NOREQ:
        stride = (NM - NQ - 13 - 4) // len(reqs)            # 12 requests: every 5 gaps, 8: every 8
        assert stride >= 3                                   # (the M0 write at gap - 1 follows the previous request's issue)
        for k, rq in enumerate(reqs):
            gap = 13 + stride * k + wave
            assert BAR_GAP < gap - 1 and gap < NM - NQ, gap
            rgap.setdefault(gap - 1, []).append(rq[0])
            rgap.setdefault(gap, []).append(rq[1])
    assert len(units_head) <= BAR_GAP and len(units_tail) <= NQ
    n = 0
    for i in range(NI):
        for q in range(NQ):
            if i == 0:
                e(f"s_waitcnt lgkmcnt({min(15, pending_after_b(q))})")
            if i == NI - 1 and q == 0:
                e(f"s_waitcnt lgkmcnt({a_reads_per_tile() * (NI - 1) if reads is not None else 0})")
            c = "0" if zero else acc(i, q)
            e(f"v_mfma_f32_16x16x32_bf16 {acc(i, q)}, {fb(q)}, {fa(i)}, {c}")
            if n < len(units_head):
                for s in units_head[n]:
                    e(s)
            if KMA and (i & 1) == (wave & 1) and 1 <= q <= 4:    # column sums of A: this wave's tiles, behind the row's first MFMAs
                e(f"v_dot2c_f32_bf16 %[rs{i >> 1}], %[ones], v{FA + 4 * i + q - 1}")
            if n == BAR_GAP:
                vm_split = STAMP and os.environ.get("GEMMFR_STAMP_VM") == "1"   # experiment: d2 = the vmcnt wait alone, d1 = the barrier alone
                if STAMP:
                    e(f"s_memtime s[{ST['t1']}:{ST['t1'] + 1}]")
                if vm_split:
                    e("s_waitcnt lgkmcnt(0)")
                    e(f"s_mov_b32 s{ST['t0']}, s{ST['t1']}")
                e(f"s_waitcnt vmcnt({0 if NOREQ else vmcnt})")
                e("s_waitcnt lgkmcnt(0)")
                if vm_split:
                    e(f"s_memtime s[{ST['t1']}:{ST['t1'] + 1}]")
                    e("s_waitcnt lgkmcnt(0)")
                    e(f"s_sub_u32 s{ST['t0'] + 1}, s{ST['t1']}, s{ST['t0']}")
                    e(f"s_add_u32 s{ST['d2']}, s{ST['d2']}, s{ST['t0'] + 1}")
                    e(f"s_mov_b32 s{ST['t0']}, s{ST['t1']}")
                e("s_barrier")
                if vm_split:
                    e(f"s_memtime s[{ST['t1']}:{ST['t1'] + 1}]")
                    e("s_waitcnt lgkmcnt(0)")
                    e(f"s_sub_u32 s{ST['t0'] + 1}, s{ST['t1']}, s{ST['t0']}")
                    e(f"s_add_u32 s{ST['d1']}, s{ST['d1']}, s{ST['t0'] + 1}")
                elif STAMP:   # d1 += phase (stamp to stamp), d2 += barrier wait
                    e(f"s_sub_u32 s{ST['t0'] + 1}, s{ST['t1']}, s{ST['t0']}")
                    e(f"s_add_u32 s{ST['d1']}, s{ST['d1']}, s{ST['t0'] + 1}")
                    e(f"s_mov_b32 s{ST['t0']}, s{ST['t1']}")
                    e(f"s_memtime s[{ST['t1']}:{ST['t1'] + 1}]")
                    e("s_waitcnt lgkmcnt(0)")
                    e(f"s_sub_u32 s{ST['t1'] + 1}, s{ST['t1']}, s{ST['t0']}")
                    e(f"s_add_u32 s{ST['d2']}, s{ST['d2']}, s{ST['t1'] + 1}")
            if reads is not None:
                if q == NQ - 1 and i < NI - 1:
                    for s in a_read(i, akk):
                        e(s)
                if i == NI - 1:
                    for s in b_reads(q, reads, wave):
                        e(s)
                    if q == NQ - 1:
                        for s in a_read(NI - 1, akk):
                            e(s)
            for s in rgap.get(n, []):
                e(s)
            if n >= NM - NQ and n - (NM - NQ) < len(units_tail):
                for s in units_tail[n - (NM - NQ)]:
                    e(s)
            n += 1


def step(e, wave, kind, req_b, req_a, vm_even, vm_odd):
    """one K-step s = an even and an odd phase.  kind: 'first' (accumulators start from zero), 'mid', 'last' (its odd phase reads
    no fragments).  req_b / req_a: B (s + 2) is requested in the even phase, A (s + 2) in the odd one."""
    # even phase: reads A (s, 1) from the current A slot and B (s + 1, half 0) from the other B slot; requests B (s + 2) into the current B slot
    head = [[s] for s in addr_setup(1, S["acur"], S["bnxt"])] + [[f"s_add_u32 s{S['dstB']}, s{S['bcur']}, %[wldsb]"]]
    tail = (advance_b() if req_b and "b" not in FIXSRC else []) + [swap("bcur", "bnxt")]
    phase(e, wave, kind == "first", 0, requests_b() if req_b else None, vm_even, head, tail, akk=1)
    # odd phase: reads A (s + 1, 0) from the other A slot and B (s + 1, half 1) from the (now) current B slot; requests A (s + 2)
    last = kind == "last"
    head = ([[s] for s in addr_setup(0, S["anxt"], S["bcur"])] if not last else []) + [[f"s_add_u32 s{S['dstA']}, s{S['acur']}, %[wldsa]"]]
    tail = (advance_a() if req_a and "a" not in FIXSRC else []) + [swap("acur", "anxt")]
    phase(e, wave, False, None if last else 1, requests_a() if req_a else None, vm_odd, head, tail, akk=0)


def kloop():
    e_lines = []
    e = e_lines.append
    e(f"s_mov_b32 s{S['m0save']}, m0")
    if STAMP:
        e(f"s_mov_b32 s{ST['d1']}, 0")
        e(f"s_mov_b32 s{ST['d2']}, 0")
    # request pointers start at stage 2 (stages 0 and 1 of this tile were requested by the previous tile's last steps, or by the
    # kernel's prologue)
    e(f"s_mov_b64 s[{S['pa']}:{S['pa'] + 1}], %[aptr]")
    if KMA:
        for _ in range(2):
            e(f"s_add_u32 s{S['pa']}, s{S['pa']}, %[ldas]")
            e(f"s_addc_u32 s{S['pa'] + 1}, s{S['pa'] + 1}, 0")
    else:
        e(f"s_add_u32 s{S['pa']}, s{S['pa']}, 256")
        e(f"s_addc_u32 s{S['pa'] + 1}, s{S['pa'] + 1}, 0")
        for s in derive_pa1():
            e(s)
    e(f"s_mov_b32 s{S['bv']}, %[bv2]")
    for s in b_pointer():
        e(s)
    e(f"s_sub_u32 s{S['tend']}, %[nk], 2")                    # index of the last step but one
    # slots: A at 0 / 32768, B at 65536 / 114688 (= 65536 + 1.5 x the A slot's offset); both flip once per step
    e(f"s_mov_b32 s{S['acur']}, %[acur]")
    e(f"s_xor_b32 s{S['anxt']}, s{S['acur']}, {A_SLOT}")
    e(f"s_lshr_b32 s{S['tmp']}, s{S['acur']}, 1")
    e(f"s_add_u32 s{S['bcur']}, s{S['acur']}, s{S['tmp']}")
    e(f"s_add_u32 s{S['bcur']}, s{S['bcur']}, {B_BASE}")
    e(f"s_lshr_b32 s{S['tmp']}, s{S['anxt']}, 1")
    e(f"s_add_u32 s{S['bnxt']}, s{S['anxt']}, s{S['tmp']}")
    e(f"s_add_u32 s{S['bnxt']}, s{S['bnxt']}, {B_BASE}")
    for w in range(1, 4):
        e(f"s_cmp_eq_u32 %[wv], {w}")
        e(f"s_cbranch_scc1 L_fr_w{w}_%=")
    for w in range(4):
        e(f"L_fr_w{w}_%=:")
        # fragments of phase 0: A (0, kk 0), B (0, half 1) -- both stages have landed and every wave knows (previous tile's barriers / prologue)
        for s in addr_setup(0, S["acur"], S["bcur"]):
            e(s)
        for i in range(NI):
            for s in a_read(i, 0):
                e(s)
        for q in range(NQ):
            for s in b_reads(q, 1, w):
                e(s)
        e("s_waitcnt lgkmcnt(0)")
        if STAMP:
            e(f"s_memtime s[{ST['t0']}:{ST['t0'] + 1}]")
            e("s_waitcnt lgkmcnt(0)")
        # step 0: its even barrier waits for everything (B (1) is the newest request of the prologue; across a tile boundary only
        # A (1) is newer)
        step(e, w, "first", True, True, 0, 12)
        e(f"s_mov_b32 s{S['t']}, 1")
        e(f"s_cmp_lt_u32 s{S['t']}, s{S['tend']}")
        e(f"s_cbranch_scc0 L_fr_tail{w}_%=")
        e(f"L_fr_loop{w}_%=:")
        step(e, w, "mid", True, True, 8, 12)
        e(f"s_add_u32 s{S['t']}, s{S['t']}, 1")
        e(f"s_cmp_lt_u32 s{S['t']}, s{S['tend']}")
        e(f"s_cbranch_scc1 L_fr_loop{w}_%=")
        e(f"L_fr_tail{w}_%=:")
        if not KMA:
            e("s_cmp_eq_u32 %[hasnext], 0")
            e(f"s_cbranch_scc1 L_fr_nonext{w}_%=")
            # steps nk - 2, nk - 1 with a next tile: A (nk), A (nk + 1) are its stages 0, 1; B runs on cyclically
            e(f"s_mov_b64 s[{S['pa']}:{S['pa'] + 1}], %[anext]")
            for s in derive_pa1():
                e(s)
            step(e, w, "mid", True, True, 8, 12)
            step(e, w, "last", True, True, 8, 12)
            e("s_branch L_fr_done_%=")
            e(f"L_fr_nonext{w}_%=:")
        # without one: step nk - 2 requests B (nk) (its first half is the tile's last k-half) and no A; step nk - 1 requests nothing
        step(e, w, "mid", True, False, 8, 12)     # even barrier: B (nk - 1) landed, A (nk - 1) may fly; odd: A (nk - 1) landed, B (nk) may fly
        step(e, w, "last", False, False, 0, 0)    # even barrier: B (nk) landed (nothing is newer)
        if w < 3:
            e("s_branch L_fr_done_%=")
    e("L_fr_done_%=:")
    e(f"s_mov_b32 %[acur], s{S['acur']}")                     # slot of the next tile's stage 0 (the last step's swaps have run)
    e(f"s_mov_b32 m0, s{S['m0save']}")
    if STAMP:
        e(f"s_mov_b32 %[d1], s{ST['d1']}")
        e(f"s_mov_b32 %[d2], s{ST['d2']}")
    e("s_nop 15")          # MFMA results -> the epilogue's reads (the assembler pads nothing inside or after asm)
    e("s_nop 15")
    return e_lines


def clobbers():
    c = [f"v{r}" for r in range(FA, VACC)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(70, 98 if STAMP else S_LAST)]
    return c + ["memory", "scc"]


def main(out):
    global STAMP
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemmfr.py -- do not edit (tests/test_abi.py::test_gemmfr_inc_is_current regenerates and compares).\n")
        f.write("// The K loop of the full-row 256x384x64 GEMM as one inline-asm statement; operands and register plan: the generator's docstring.\n")
        total = 0
        global PAIRED
        global KMA, TB
        for stamped, paired, kma in ((False, True, False), (False, False, False), (True, True, False), (True, False, False),
                                     (False, True, True), (True, True, True)):
            STAMP, PAIRED, KMA, TB = stamped, paired, kma, (4 if kma else 2)
            name = ("DEVIT_WGRADFR_KLOOP" if kma else "DEVIT_GEMMFR_KLOOP_" + ("PAIRED" if paired else "NATURAL")) + ("_STAMPED" if stamped else "")
            lines = kloop()
            n_mfma = sum(1 for s in lines if s.startswith("v_mfma"))
            assert n_mfma == 4 * (4 if kma else 6) * 2 * NM, n_mfma
            if kma:
                n_dot = sum(1 for s in lines if s.startswith("v_dot2c"))
                assert n_dot == 4 * 4 * 2 * 16, n_dot
            total += len(lines)
            if stamped:
                f.write("// diagnostic variant (-DDEVIT_GEMMFR_STAMP): s_memtime deltas per phase and per barrier wait, operands [d1], [d2] in addition\n")
            if kma and not stamped:
                f.write("// the WEIGHT-GRADIENT variant (wgradfr_kernel): A k-major, PAIRED order on both sides, column sums of A, one tile per workgroup\n")
            f.write(f"#define {name}_ASM \\\n")
            for s in lines:
                f.write(f'  "{s}\\n\\t" \\\n')
            f.write('  ""\n')
            f.write(f"#define {name}_CLOBBERS " + ", ".join(f'"{c}"' for c in clobbers()) + "\n\n")
        KMA, TB = False, 2
        STAMP = False
        f.write("// accumulators of m-tiles I0, I0 + 1 and the four n-tiles of column group H < 2 (tile q = 4 H + j) out of a[0:255]\n")
        f.write("template <int H, int I0>\n__device__ __forceinline__ void gemmfr_read_acc(f32x4 (&acc)[2][4]) {\n")
        f.write("  static_assert(H >= 0 && H < 2 && I0 >= 0 && I0 < 8 && I0 % 2 == 0, \"\");\n")
        for h in range(2):
            for i0 in range(0, 8, 2):
                f.write(f"  if constexpr (H == {h} && I0 == {i0}) {{\n")
                for u in range(2):
                    for j in range(4):
                        a = 4 * (8 * (i0 + u) + 4 * h + j)
                        f.write(f"    {{ float x0, x1, x2, x3; asm volatile(\"v_accvgpr_read_b32 %0, a{a}\\n\\tv_accvgpr_read_b32 %1, a{a + 1}\\n\\t"
                                f"v_accvgpr_read_b32 %2, a{a + 2}\\n\\tv_accvgpr_read_b32 %3, a{a + 3}\" : \"=v\"(x0), \"=v\"(x1), \"=v\"(x2), \"=v\"(x3) :: \"memory\"); "
                                f"acc[{u}][{j}] = (f32x4){{x0, x1, x2, x3}}; }}\n")
                f.write("  }\n")
        f.write("}\n")
    print(f"wrote {out}: {total} asm lines")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "devit_amd", "csrc", "gemmfr_kloop.inc"))
