"""CPU: a replay of the issue schedule of the row-resident LayerNorm + x3 GEMM (csrc/kernels_lngemm.hip lg_step) — the
protocol that r05's first GPU run got wrong (29 items per tile on a 6-deep queue: item 0 of the next tile landed in the slot
of an unconsumed fragment).  The constants are parsed from the kernel source; the replay walks the steps of several tiles in
program order and checks, for one wavefront:
  * queue slots: the slot a step consumes holds exactly (tile, item) — nothing was overwritten before it was used;
  * counted waits: `s_waitcnt lgkmcnt(N)` in front of a step leaves at most N LDS operations outstanding, and the LDS
    operations of a wave complete in order, so the awaited fragment pair has landed iff at least ... at most N operations
    were issued BEHIND it (the epilogue's extra LDS operations only make the wait stricter);
  * the 2-stage ring: a fragment of tile t is read only after the barrier that certifies its stage (tiles 0 / 1: the prologue),
    and the DMA of tile t + 2 into the stage of tile t is issued only behind tile t's barrier, by which every read of tile t
    has been issued; all 16 pieces of a tile are issued before the barrier that certifies it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _consts():
    src = open(os.path.join(ROOT, "layout_dm_amd", "csrc", "kernels_lngemm.hip")).read()
    m = re.search(r"constexpr int LG_KS = (\d+), LG_NIT = (\d+), LG_PF = (\d+), LG_SYNC = LG_NIT - LG_PF;", src)
    assert m, "schedule constants not found: update this replay together with the kernel"
    ks, nit, pf = (int(x) for x in m.groups())
    # the structural facts the replay mirrors must still be in the source
    for needle in ("if constexpr (IT == LG_SYNC - 1)", "constexpr int RI = (IT + LG_PF) % LG_NIT;",
                   "constexpr bool hasR = kRd && IT != LG_SYNC && RI < LG_KS;", "if constexpr (kRd) lg_read<RI, W2>(s);", "constexpr bool hasD = kDm && J >= 0 && (!W2 || J < 8);",
                   "if constexpr (hasD && J == 0) lg_dma_begin(s, tile + 2);", "if (IT > LG_SYNC) return IT - LG_SYNC - 1;",
                   "if (IT + (LG_NIT - 1 - LG_SYNC) < 16) return IT + (LG_NIT - 1 - LG_SYNC);",
                   "constexpr int W = kRd ? lg_younger(IT, W2 ? 1 : 2) : 15;", "const unsigned aw = s.aW[RI & 7];",
                   "return IT == 1 || IT == 2 || IT == 3 || (IT > LG_SYNC && IT <= LG_SYNC + 4);",
                   "if constexpr (kEp && IT == LG_KS) lg_epilogue_sum_write(e, s.accA, s.accB);", "wait_lgkm<8>();   // (4 ds_write_b128 follow",
                   "for (int j = 1; j < LG_PF; ++j) n += lg_real(IT + j) ? rpi : 0;",
                   "if constexpr (!W2) lg_dsr<256 * (IT >> 3) + LG_LO>(s.ql[IT % LG_PF], s.aW[IT & 7]);", "if constexpr (IT < LG_KS) {\n    lg_dsr"):
        assert needle in src, needle
    return ks, nit, pf


def test_schedule_replay():
    KS, NIT, PF = _consts()
    _replay(KS, NIT, PF, epilogue=True)
    # the two-product form (W2, the mixed numerics mode): ONE fragment read per item (W hi only), the counted waits halve with it
    _replay(KS, NIT, PF, epilogue=True, rpi=1, n_pieces=8)


def test_schedule_replay_gemm_prologue():
    """The GEMM prologue (lp_step) runs the same protocol with 30 REAL items per stage (2 k16-steps x 15 column tiles, no pseudo
    item), no epilogue slices, its own (shallower) queue and the DMA schedule that follows from it (lp_piece); its A fragments are
    requested behind the stage's last DMA piece, so that the barrier's counted wait may leave exactly those loads outstanding."""
    _, NIT, _ = _consts()
    src = open(os.path.join(ROOT, "layout_dm_amd", "csrc", "kernels_lngemm.hip")).read()
    m = re.search(r"constexpr int LP_PF = (\d+), LP_SYNC = LP_NIT - LP_PF;", src)
    a = re.search(r"constexpr int LP_A_STEP = (\d+);", src)
    n = re.search(r"constexpr int LP_A_LOADS = (\d+);", src)
    assert m and a and n
    PF, A_STEP, A_LOADS = int(m.group(1)), int(a.group(1)), int(n.group(1))
    for needle in ("constexpr int LP_NT = 15, LP_NIT = 2 * LP_NT;", "static_assert(LP_NIT % LP_PF == 0,",
                   "constexpr int RI = (IT + LP_PF) % LP_NIT;", "constexpr bool hasR = IT != LP_SYNC;", "constexpr bool hasD = J >= 0 && (!W2 || J < 8);", "[w] \"n\"(2 * (LP_PF - 1))", "[w] \"n\"(LP_PF - 1)",
                   "if constexpr (!W2) lg_dsr<t * 2048 + LG_LO>(s.ql[IT % LP_PF], s.aS[sx]);",
                   "if constexpr (IT == LP_SYNC - 1) {", "if constexpr (hasD && J == 0) lp_dma_begin(s, stage + 2);",
                   "if (IT > LP_SYNC) return IT - LP_SYNC - 1;", "if (IT + (LP_NIT - 1 - LP_SYNC) < 16) return IT + (LP_NIT - 1 - LP_SYNC);",
                   "if constexpr (IT == LP_A_STEP) lp_load_a<(SET + 2) % 3, NP == 1>(s, stage + 2);",
                   "asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(NP == 1 ? LP_A_LOADS / 2 : LP_A_LOADS) : \"memory\");",
                   "if constexpr (IT == LP_NIT - 1) LP_STEP_ASM(LG_A_M0, LG_A_PIECE, LG_A_RDH, LG_A_RDL \"\\n\\t\", \"s_nop 15\\n\\ts_nop 15\");"):
        assert needle in src, needle
    SYNC = NIT - PF
    _replay(NIT, NIT, PF, epilogue=False)
    _replay(NIT, NIT, PF, epilogue=False, rpi=1, n_pieces=8)
    # vector-memory order inside a stage: DMA pieces at steps SYNC + 1 .. and 0 .. (16 in all), then the A loads; nothing behind them
    pieces = [it for it in range(NIT) if it > SYNC or it + (NIT - 1 - SYNC) < 16]
    assert len(pieces) == 16 and A_LOADS == 4
    assert max(p for p in pieces if p < SYNC) < A_STEP < SYNC, "the A loads must be the youngest vector memory operations at the barrier"


def _replay(KS, NIT, PF, epilogue, rpi=2, n_pieces=16):
    SYNC = NIT - PF
    assert NIT % PF == 0 and KS <= NIT
    n_tiles = 6
    lds_ops = []            # program-order list of LDS operations: ("frag", tile, item) rpi times per item (W hi, W lo | W hi), or ("epi",)
    slot = {}               # queue slot -> (tile, item) it will hold once landed
    stage_tile = {0: 0, 1: 1}            # stage -> tile whose image the prologue / the DMA put there
    certified = {0, 1}                    # tiles whose stage is known complete (prologue barrier / a tile barrier)
    barrier_done = set()                  # tiles whose step-SYNC barrier has been passed
    dma_pieces = {}                       # tile -> pieces issued
    cur_stage_of_aw = 0

    def issue_read(tile, item):
        nonlocal lds_ops
        if item >= KS:          # pseudo item: nothing is read (the slot bookkeeping alone)
            slot[item % PF] = (tile, item)
            return
        assert tile in certified, f"tile {tile} read before its stage was certified"
        assert stage_tile[cur_stage_of_aw] == tile, f"aW points at stage {cur_stage_of_aw} = tile {stage_tile[cur_stage_of_aw]}, wanted {tile}"
        slot[item % PF] = (tile, item)
        lds_ops += [("frag", tile, item)] * rpi

    for i in range(PF):                   # kernel prologue: lg_read<0..5>
        issue_read(0, i)
    for t in range(n_tiles):
        last = t + 1 >= n_tiles
        for it in range(NIT):
            # ---- the counted wait in front of the step (real items only)
            real = lambda i: (i % NIT) < KS
            if it < KS:
                n = sum(rpi for j in range(1, PF) if real(it + j))   # uniform: the reads run on behind the last tile
                assert n <= 15
                idx = max(i for i, op in enumerate(lds_ops) if op == ("frag", t, it))      # the younger of the pair
                younger = len(lds_ops) - 1 - idx
                assert n <= younger, f"tile {t} step {it}: lgkmcnt({n}) but only {younger} operations were issued behind the pair"
            # ---- consumption
            assert slot[it % PF] == (t, it), f"tile {t} step {it}: slot holds {slot[it % PF]}"
            # ---- barrier
            if it == SYNC:
                if t + 1 < n_tiles:
                    assert dma_pieces.get(t + 1, n_pieces) == n_pieces, f"tile {t + 1}: only {dma_pieces.get(t + 1)} pieces issued before its barrier"
                    certified.add(t + 1)
                barrier_done.add(t)
            # ---- reads
            if it + PF < NIT:
                issue_read(t, it + PF)
            elif not last:
                issue_read(t + 1, it + PF - NIT)
            elif it + PF - NIT < KS:
                # behind the last tile the reads go on into the other stage (bytes nobody consumes): LDS operations all the same
                lds_ops += [("frag", t + 1, it + PF - NIT)] * rpi
            if it == SYNC - 1:
                cur_stage_of_aw ^= 1
            # ---- DMA of tile t + 2 (behind this tile's barrier) / the rest of tile t + 1
            piece = None
            if it > SYNC:
                piece, td = it - SYNC - 1, t + 2
            if it + (NIT - 1 - SYNC) < 16:
                piece, td = it + (NIT - 1 - SYNC), t + 1      # (tile 0: the pieces re-load tile 1, which the prologue brought)
            if piece is not None and piece >= n_pieces:       # (W2: the hi half of the stage only — pieces 0 .. 7)
                piece = None
            if piece is not None and td < n_tiles and td >= 2:
                assert td - 2 in barrier_done, f"DMA of tile {td} into the stage of tile {td - 2} before that tile's barrier"
                assert dma_pieces.get(td, 0) == piece, f"tile {td}: piece {piece} out of order"
                dma_pieces[td] = piece + 1
                stage_tile[td & 1] = td
            if piece is not None and td >= n_tiles:
                # clamped: the last tile once more into the stage of tile td - 2, which must be free (its barrier passed)
                assert td - 2 in barrier_done
            # ---- the previous tile's epilogue: LDS operations of its three slices
            # (issued between the step's MFMAs, i.e. BEFORE the step's fragment reads: modelled by inserting them in front of the pair
            #  just appended — conservative for the waits either way)
            if t > 0 and epilogue:
                extra = 1 if it == 1 else 2 if it in (2, 3) else 0
                assert rpi * (PF - 1) + rpi + extra <= 15, "lgkmcnt is a 4-bit counter"
                lds_ops += [("epi",)] * extra
            if it == KS and epilogue:                 # the pseudo step: drain to 8, then the 4 ds_write_b128 of the tile's sum
                assert 8 + 4 <= 15
                lds_ops += [("epi",)] * 4
    for td in range(2, n_tiles):
        assert dma_pieces[td] == n_pieces
