"""Host-verifiable bookkeeping of the fused attention + out_proj kernel of the split mode (layout_dm_amd/csrc/kernels_attnout.hip):
its address formulas, restated here, are run against a byte-level model of the LDS —

  * the whole-head LDS-DMA of K / V (per-lane SOURCE permutation inside each 1-KiB piece, linear destination) followed by the K fragment reads
    (ds_read_b128) and the V transpose reads (ds_read_b64_tr_b16: a 16-lane group reads a [4 keys][16 d] block, lane i receives
    column i — cdna_hip_programming.md T10) deliver exactly the MFMA operand elements the kernel's contraction order assumes;
  * every read pattern is bank-conflict free under the gfx950 service model (MI355X_MICROARCH.md, LDS table);
  * the counted s_waitcnt vmcnt(N) of the six barriers per head equal the number of vector-memory operations really issued behind
    the awaited unit (a replay of the kernel's issue order).
No GPU."""
import numpy as np

from test_lds_swizzle import conflict_free

KH, LO = 0, 16384


def voff_k(lane):
    return ((lane >> 2) << 6) | (((lane & 3) ^ ((lane >> 4) & 3)) << 4)


def voff_v(lane):
    return ((4 * (lane >> 4) + ((lane >> 1) & 3)) << 6) | ((2 * ((lane >> 3) & 1) + (lane & 1)) << 4)


def a_k(lane, odd):
    m, g = lane & 31, lane >> 5
    return (m << 6) | (((2 * odd + g) ^ ((m >> 2) & 3)) << 4)


def a_row(lane):
    m, g = lane & 31, lane >> 5
    return (m << 5) | ((g ^ ((m >> 3) & 1)) << 4)


def dma_head_image(src, voff):
    """src: [2 panels][128 keys][32 d] uint16 (one of hi / lo, a layout's slab of each panel) -> 16 KiB LDS image (uint16 view):
    wave w moves pieces 4 w .. 4 w + 3 = panel w >> 1, keys 64 (w & 1) + 16 j."""
    lds = np.zeros(8192, np.uint16)
    for wave in range(4):
        flat = src[wave >> 1].reshape(-1)       # the layout's rows of the panel: key * 32 + d  (64 B per key)
        for piece in range(4):
            for lane in range(64):
                so = ((wave & 1) * 4096 + piece * 1024 + voff(lane)) // 2
                do = (wave * 4096 + piece * 1024 + lane * 16) // 2
                lds[do:do + 8] = flat[so:so + 8]
    return lds


def test_k_fragments_and_v_transpose_reads_deliver_the_mfma_operands():
    key, d = np.meshgrid(np.arange(128), np.arange(64), indexing="ij")
    val = (key * 64 + d).astype(np.uint16)                       # element (key, d) of the head, unique
    panels = np.stack([val[:, 32 * p:32 * p + 32] for p in range(2)])
    lds = dma_head_image(panels, voff_k)
    # K: A operand of S^T tile kt, k16-step ks: lane (m, g) holds K[32 kt + m][16 ks + 8 g + e]
    for kt in range(4):
        for ks in range(4):
            for lane in range(64):
                m, g = lane & 31, lane >> 5
                ad = KH + (ks >> 1) * 8192 + kt * 2048 + a_k(lane, ks & 1)
                got = lds[ad // 2: ad // 2 + 8]
                want = val[32 * kt + m, 16 * ks + 8 * g: 16 * ks + 8 * g + 8]
                assert np.array_equal(got, want), (kt, ks, lane)
    # V: A operand of O^T tile dt, k16-step (kt, hf): lane (m, g) element e holds V[32 kt + 16 hf + 8 (e >> 2) + 4 g + (e & 3)][32 dt + m]
    # — the k-slot order in which the lane's score registers 8 hf .. 8 hf + 7 of tile kt hold its probabilities
    lds = dma_head_image(panels, voff_v)
    for dt in range(2):
        for kt in range(4):
            for hf in range(2):
                off = dt * 8192 + kt * 2048 + hf * 1024
                for second in (0, 1):
                    raw = {}
                    for lane in range(64):
                        ad = lane * 8 + off + 512 * second
                        raw[lane] = lds[ad // 2: ad // 2 + 4]
                    for lane in range(64):
                        G, i = lane >> 4, lane & 15
                        got = [raw[16 * G + 4 * j + (i >> 2)][i & 3] for j in range(4)]   # the hardware's 4 x 16 transpose
                        m, g = lane & 31, lane >> 5
                        want = [val[32 * kt + 16 * hf + 8 * second + 4 * g + j, 32 * dt + m] for j in range(4)]
                        assert got == want, (dt, kt, hf, second, lane)


def test_read_patterns_are_bank_conflict_free():
    for odd in (0, 1):
        assert conflict_free(lambda l: a_k(l, odd))              # K fragments (64-byte rows)
    assert conflict_free(a_row)                                  # Wo stage rows (32-byte rows)
    # without the chunk swizzles rows of a service group share their banks
    assert not conflict_free(lambda l: ((l & 31) << 6) | ((l >> 5) << 4))
    assert not conflict_free(lambda l: ((l & 31) << 5) | ((l >> 5) << 4))
    # the V transpose reads are the guide's known-good form: lane l at + 8 l of a 512-byte run (nothing to model)


def test_counted_waits_match_the_issue_order():
    """Replay of the kernel's vector-memory issue order (units of 8 instructions per wave) and of its six barriers per head:
    vmcnt(N) at a barrier must leave exactly the units issued BEHIND the awaited one in flight."""
    U = 8
    issued = []                                                   # unit names in issue order

    def issue(name, n=U):
        issued.append((name, n))

    def check(awaited, n_wait):
        idx = max(i for i, (nm, _) in enumerate(issued) if nm == awaited)
        behind = sum(n for _, n in issued[idx + 1:])
        assert behind == n_wait, (awaited, behind, n_wait)

    issue("Q0"); issue("K0"); issue("V0"); issue("dummy"); issue("W0.0"); issue("W0.1")
    check("K0", 4 * U)                                            # Ba(0)
    issue("W0.2")
    for h in range(8):
        check(f"V{h}", 4 * U)                                     # Bb(h)
        issue(f"Q{h + 1}"); issue(f"K{h + 1}")
        check(f"W{h}.0", 4 * U)                                   # Bc(h)
        issue(f"V{h + 1}")
        check(f"W{h}.1", 4 * U)                                   # Bd1
        issue(f"W{h}.3")
        check(f"W{h}.2", 4 * U)                                   # Bd2
        issue(f"W{h + 1}.0")
        check(f"W{h}.3", 1 * U)                                   # Bd3
        issue(f"W{h + 1}.1")
        check(f"K{h + 1}", 4 * U)                                 # Ba(h + 1): Q and K of the next head
        issue(f"W{h + 1}.2")
    assert max(sum(n for _, n in issued[i:i + 6]) for i in range(len(issued))) <= 63   # vmcnt is a 6-bit counter


def test_ring_slots_follow_stage_mod_3():
    """The kernel tracks slot = (4 h) % 3 incrementally and addresses stage 4 h + st / its refills relative to it."""
    slot = 0
    for h in range(8):
        assert slot == (4 * h) % 3
        sl_st = [slot, 0 if slot == 2 else slot + 1, 2 if slot == 0 else slot - 1, slot]
        assert sl_st == [(4 * h + st) % 3 for st in range(4)]
        assert slot == (4 * h + 3) % 3                                       # Bd1: W_h,3 into stage 0's slot
        assert (0 if slot == 2 else slot + 1) == (4 * h + 4) % 3              # Bd2: W_h+1,0 into stage 1's slot
        assert (2 if slot == 0 else slot - 1) == (4 * h + 5) % 3              # Bd3: W_h+1,1 into stage 2's slot
        slot = 0 if slot == 2 else slot + 1
        assert (2 if slot == 0 else slot - 1) == (4 * h + 6) % 3              # Ba(h + 1): W_h+1,2 into stage 3's slot
