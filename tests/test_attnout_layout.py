"""Host-verifiable bookkeeping of the fused attention + out_proj kernel of the split mode (layout_dm_amd/csrc/kernels_attnout.hip):
its address formulas, restated here, are run against a byte-level model of the LDS —

  * the whole-panel LDS-DMA of K / V (per-lane SOURCE swizzle, linear destination) followed by the K fragment reads
    (ds_read_b128) and the V transpose reads (ds_read_b64_tr_b16: a 16-lane group reads a [4 keys][16 d] block, lane i receives
    column i — cdna_hip_programming.md T10) deliver exactly the MFMA operand elements the kernel's contraction order assumes;
  * every read pattern is bank-conflict free under the gfx950 service model (MI355X_MICROARCH.md, LDS table);
  * the counted s_waitcnt vmcnt(N) of the six barriers per head equal the number of vector-memory operations really issued behind
    the awaited unit (a replay of the kernel's issue order).
No GPU."""
import numpy as np

from test_lds_swizzle import conflict_free

KH, LO = 0, 16384


def voff_kv(lane, wave):
    pos = lane >> 1
    return ((pos ^ (4 if wave & 1 else 0)) << 5) | (((lane & 1) ^ ((pos >> 3) & 1)) << 4)


def a_row(lane, odd):
    m, g = lane & 31, lane >> 5
    return (((m ^ 4) if odd else m) << 5) | ((g ^ ((m >> 3) & 1)) << 4)


def a_v(lane, second):
    G, sl = lane >> 4, lane & 15
    key0 = (4 * (G >> 1) + (sl >> 2)) ^ (4 * (G & 1))
    chunk = ((sl & 3) >> 1) ^ (1 if second else 0)
    return (G & 1) * 4096 + (key0 << 5) + (chunk << 4) + ((sl & 1) << 3) + (256 if second else 0)


def dma_head_image(src):
    """src: [4 panels][128 keys][16 d] uint16 (one of hi / lo, a layout's slab of each panel) -> 16 KiB LDS image (uint16 view)."""
    lds = np.zeros(8192, np.uint16)
    for wave in range(4):                       # wave w moves panel w
        flat = src[wave].reshape(-1)            # the layout's rows of the panel: key * 16 + d  (32 B per key)
        for piece in range(4):
            for lane in range(64):
                so = (voff_kv(lane, wave) + piece * 1024) // 2
                do = (wave * 4096 + piece * 1024 + lane * 16) // 2
                lds[do:do + 8] = flat[so:so + 8]
    return lds


def test_k_fragments_and_v_transpose_reads_deliver_the_mfma_operands():
    key, d = np.meshgrid(np.arange(128), np.arange(64), indexing="ij")
    val = (key * 64 + d).astype(np.uint16)                       # element (key, d) of the head, unique
    panels = np.stack([val[:, 16 * p:16 * p + 16] for p in range(4)])
    lds = dma_head_image(panels)
    # K: A operand of S^T tile kt, k16-step ks: lane (m, g) holds K[32 kt + m][16 ks + 8 g + e]
    for kt in range(4):
        for ks in range(4):
            for lane in range(64):
                m, g = lane & 31, lane >> 5
                ad = KH + ks * 4096 + kt * 1024 + a_row(lane, ks & 1)
                got = lds[ad // 2: ad // 2 + 8]
                want = val[32 * kt + m, 16 * ks + 8 * g: 16 * ks + 8 * g + 8]
                assert np.array_equal(got, want), (kt, ks, lane)
    # V: A operand of O^T tile dt, k16-step (kt, hf): lane (m, g) element e holds V[32 kt + 16 hf + 8 (e >> 2) + 4 g + (e & 3)][32 dt + m]
    # — the k-slot order in which the lane's score registers 8 hf .. 8 hf + 7 of tile kt hold its probabilities
    for dt in range(2):
        for kt in range(4):
            for hf in range(2):
                off = dt * 8192 + kt * 1024 + hf * 512
                for second in (0, 1):
                    raw = {}
                    for lane in range(64):
                        ad = a_v(lane, second) + off
                        assert ad % 8 == 0                       # (G17: a misaligned tr read returns the aligned address's data)
                        raw[lane] = lds[ad // 2: ad // 2 + 4]
                    for lane in range(64):
                        G, i = lane >> 4, lane & 15
                        got = [raw[16 * G + 4 * j + (i >> 2)][i & 3] for j in range(4)]   # the hardware's 4 x 16 transpose
                        m, g = lane & 31, lane >> 5
                        want = [val[32 * kt + 16 * hf + 8 * second + 4 * g + j, 32 * dt + m] for j in range(4)]
                        assert got == want, (dt, kt, hf, second, lane)


def test_read_patterns_are_bank_conflict_free():
    for odd in (0, 1):
        assert conflict_free(lambda l: a_row(l, odd))            # K fragments; even form = the Wo stage rows as well
    # without the chunk swizzle keys m and m + 8 of a service group share their banks
    assert not conflict_free(lambda l: ((l & 31) << 5) | ((l >> 5) << 4))
    # ds_read_b64_tr_b16: two groups of 32 lanes, 8 bytes each, 64 banks
    for second in (0, 1):
        for half in (0, 1):
            banks = set()
            for lane in range(32 * half, 32 * half + 32):
                a = a_v(lane, second)
                for w in range(2):
                    b = (a // 4 + w) % 64
                    assert b not in banks, (second, lane)
                    banks.add(b)
    # ... which is what the key ^ 4 of the odd panels buys: without it lanes 0-15 and 16-31 read the same banks of adjacent panels
    banks = [((lane >> 4) & 1) * 4096 + (((lane & 15) >> 2) << 5) + ((lane & 3) << 3) for lane in range(32)]
    assert len({(a // 4) % 64 for a in banks}) < 32


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
