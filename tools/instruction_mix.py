"""Dev tool (no GPU needed): the DYNAMIC instruction mix of the one-launch loop kernel `stack_stream_k<false, 2, false>`, per phase
and per instruction class, from its gfx950 assembly (VERDICT r3 next #6 i).

    python tools/instruction_mix.py [--asm stack.s] > profiles/r04_instruction_mix_fast_loop.txt

How: hipcc -S, the kernel's basic blocks with LLVM's loop annotations ("Loop Header: Depth=n", "in Loop: Header=..."), and
the trip counts of the reference backbone (4 layers, 8 heads, 59 pipelined FFN iterations, 8 tail rounds unrolled by 2).
Loops are recognised by their MFMA signature — the head loop holds 266 MFMAs per wave (6 x 29 in_proj + 32 attention +
60 out-projection), the FFN loop 59 (29 GEMM1 + 30 GEMM2), the vocabulary head 145 (5 x 29) — and the weighted MFMA
total is checked against the algorithmic count (22 581 per wave and reverse step = what SQ_INSTS_VALU_MFMA_MOPS / 16
measured on the MI355X: profiles/r03_final2_sq_counters_fast_loop.txt: 4.6246e9 per 512 x 100 x 4 waves).

Counts are per WAVEFRONT and per reverse step.  Blocks behind a branch are counted as executed (upper bound): that only
matters in the step's tail, where the code of all five samplers and of the near-tie report is present; the tail's
sampler loops (top-k / top-p order walk) are counted with zero trips, i.e. the table is for sampling=random.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN3ldm14stack_stream_kILb0ELi2ELb0EEEvNS_9StackArgsE"
N_LAYER, N_HEAD, N_CHUNK_IT, TAIL_ROUNDS = 4, 8, 59, 8

CLASSES = ["mfma", "valu_fp", "valu_pk", "valu_trans", "valu_cvt", "valu_acc_move", "valu_move_perm", "valu_lane_id",
           "valu_int_addr", "lds_read", "lds_write", "vmem", "salu", "s_nop", "sync"]
NOTE = {"mfma": "v_mfma_f32_32x32x16_f16", "valu_fp": "v_fma/add/mul/max/min/sub f32", "valu_pk": "v_pk_* (2 flops per lane)",
        "valu_trans": "v_exp/log/rcp/rsq/sqrt", "valu_cvt": "v_cvt_* (fp32 <-> fp16 casts)",
        "valu_acc_move": "v_accvgpr_read/write (AGPR <-> VGPR tuple copies)", "valu_move_perm": "v_mov, DPP moves, v_permlane*, v_readlane",
        "valu_lane_id": "v_mbcnt (lane id re-derived per phase)", "valu_int_addr": "integer / address / compare / select",
        "lds_read": "ds_read*", "lds_write": "ds_write*", "vmem": "global_load* (incl. LDS-DMA), global_store*, scratch_*",
        "salu": "s_* scalar ALU / moves / branches", "s_nop": "s_nop (MFMA hazard padding)", "sync": "s_waitcnt, s_barrier"}


def classify(op, rest):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "valu_acc_move"
    if op.startswith("v_mbcnt"):
        return "valu_lane_id"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_cvt"):
        return "valu_cvt"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt)_", op):
        return "valu_trans"
    if op.startswith(("v_mov", "v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
        return "valu_move_perm"
    if re.match(r"v_(fma|fmac|add|sub|mul|max|min|mad|fmaak|fmamk)_(f32|f16|legacy_f32)", op) or op in ("v_max3_f32", "v_min3_f32"):
        return "valu_move_perm" if re.search(r"row_|quad_perm|wave_", rest) and op.startswith("v_mov") else "valu_fp"
    if op.startswith("v_"):
        return "valu_int_addr"
    if op.startswith("ds_read") or op.startswith("ds_load") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle"):
        return "lds_read"
    if op.startswith("ds_"):
        return "lds_write"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op in ("s_waitcnt", "s_barrier") or op.startswith("s_waitcnt"):
        return "sync"
    if op.startswith("s_"):
        return "salu"
    return "salu"


def parse(asm_path):
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "hdr": "", "ins": []}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "hdr": m.group(2) or "", "ins": []}
            continue
        s = l.strip()
        if s.startswith(";"):
            if "Loop" in s or "%bb." in s:
                cur["hdr"] += " " + s
            continue
        if not s or s.startswith("."):
            continue
        parts = s.split(None, 1)
        cur["ins"].append((parts[0], parts[1] if len(parts) > 1 else ""))
    blocks.append(cur)
    return blocks


def loop_of(b):
    """(header label, depth) of the innermost loop the block belongs to, from LLVM's annotations."""
    h = b["hdr"]
    m = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", h)
    if m:
        return "BB" + b["label"][4:], int(m.group(1))
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", h)
    if m:
        return m.group(1), int(m.group(2))
    return None, 0


def main():
    asm = None
    if "--asm" in sys.argv:
        asm = sys.argv[sys.argv.index("--asm") + 1]
    else:
        asm = "/tmp/ldm_stack_mix.s"
        src = os.path.join(ROOT, "layout_dm_amd", "csrc", "kernels_stack.hip")
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only",
                        "-S", "-o", asm, src, "-Wno-unused-function"], check=True, stderr=subprocess.DEVNULL)
    blocks = parse(asm)
    # loop membership and MFMA totals per loop
    parent = {}
    mf = collections.Counter()
    for b in blocks:
        hdr, depth = loop_of(b)
        b["loop"], b["depth"] = hdr, depth
        m = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)\s*(?:;\s*Parent Loop (BB\d+_\d+) Depth=(\d+))?", b["hdr"])
        if hdr and "Loop Header" in b["hdr"] and m:
            parent[hdr] = m.group(3) or m.group(1)
        if hdr:
            mf[hdr] += sum(1 for op, _ in b["ins"] if op.startswith("v_mfma"))
    step = next(b["loop"] for b in blocks if b["depth"] == 1 and "Loop Header" in b["hdr"])
    head = next(h for h, n in mf.items() if n == 266)
    ffn = next(h for h, n in mf.items() if n == 59)
    layer = parent[head]
    assert parent[ffn] == layer and parent[layer] == step, (parent, head, ffn, layer, step)
    # the tail loop: the depth-2 loop of the step loop behind the layer loop that holds no MFMA
    order = [b["loop"] for b in blocks]
    tail = next(b["loop"] for b in blocks[order.index(layer):] if b["depth"] == 2 and "Loop Header" in b["hdr"]
                and b["loop"] != layer and mf[b["loop"]] == 0 and sum(len(x["ins"]) for x in blocks if x["loop"] == b["loop"]) > 500)
    trips = {step: 1, layer: N_LAYER, head: N_HEAD, ffn: N_CHUNK_IT, tail: TAIL_ROUNDS // 2}

    def weight(hdr):
        w, h = 1.0, hdr
        while h is not None:
            w *= trips.get(h, 0.0)  # unknown inner loops (gather batches, sampler walks): zero trips unless listed below
            h = parent.get(h)
        return w

    # small counted loops of the prologue / layer entry (the embedding gather batches, table DMA pieces): take their trip
    # counts from the s_cmp / loop structure is not possible in general; they hold < 1 % of the instructions, so count 1 trip
    for h in list(parent):
        if h not in trips and parent.get(h) in (step, layer) and mf[h] == 0 and h != tail:
            trips[h] = 1
    phases = collections.OrderedDict()
    seen_layer = seen_head = seen_ffn = seen_tail = False
    for b in blocks:
        lp = b["loop"]
        if lp is None:
            continue  # kernel entry / exit (once per launch)
        chain, h = [], lp
        while h is not None:
            chain.append(h)
            h = parent.get(h)
        if head in chain:
            ph, seen_head = "attention heads (in_proj stream, core, out-proj slabs)", True
        elif ffn in chain:
            ph, seen_ffn = "FFN chunk stream (pipelined GEMM1 / GEMM2)", True
        elif tail in chain:
            ph, seen_tail = "step tail (log-softmax, posterior, draw)", True
        elif layer in chain:
            seen_layer = True
            ph = ("layer entry (row statistics, AdaLN transform, tables)" if not seen_head else
                  "LN2 + FFN entry" if not seen_ffn else "layer exit")
            if seen_ffn and ph == "layer exit":
                pass
        else:
            ph = ("step prologue (embedding gather)" if not seen_layer else
                  "vocabulary head (LN + 5 tiles) + row log-softmax" if not seen_tail else "step epilogue")
        # a new layer iteration resets nothing: the blocks appear once in the assembly
        w = weight(lp)
        c = phases.setdefault(ph, collections.Counter())
        for op, rest in b["ins"]:
            c[classify(op, rest)] += w
    total = collections.Counter()
    for c in phases.values():
        total.update(c)
    valu = [k for k in CLASSES if k.startswith("valu")]
    print("# dynamic instruction mix of stack_stream_k<false,2> per WAVEFRONT and reverse step (sampling=random), from the gfx950")
    print("# assembly of this tree's kernels_stack.hip weighted by the loop trip counts (tools/instruction_mix.py)")
    print(f"# MFMA check: {total['mfma']:.0f} per wave-step (algorithmic + padding: 32 x 266 + 236 x 59 + 145 = 22581)")
    assert abs(total["mfma"] - 22581) < 1, total["mfma"]
    hdr = f"{'phase':58s}" + "".join(f"{k.replace('valu_', 'v:'):>11s}" for k in CLASSES) + f"{'VALU/MFMA':>11s}"
    print(hdr)
    for ph, c in phases.items():
        v = sum(c[k] for k in valu)
        print(f"{ph:58s}" + "".join(f"{c[k]:11.0f}" for k in CLASSES) + (f"{v / c['mfma']:11.2f}" if c["mfma"] else f"{'-':>11s}"))
    v = sum(total[k] for k in valu)
    print(f"{'TOTAL':58s}" + "".join(f"{total[k]:11.0f}" for k in CLASSES) + f"{v / total['mfma']:11.2f}")
    print()
    print(f"VALU per MFMA {v / total['mfma']:.2f} (SQ counters, r03: 3.06); LDS per MFMA "
          f"{(total['lds_read'] + total['lds_write']) / total['mfma']:.2f} (1.16); SALU + s_nop per MFMA "
          f"{(total['salu'] + total['s_nop']) / total['mfma']:.2f} (0.43)")
    print("VALU by class (share of all VALU):")
    for k in valu:
        print(f"  {k:16s} {total[k]:9.0f}  {100 * total[k] / v:5.1f} %   {NOTE[k]}")
    arith = total["valu_fp"] + total["valu_pk"] + total["valu_trans"] + total["valu_cvt"]
    print(f"arithmetic (fp / packed / transcendental / casts): {arith:.0f} = {100 * arith / v:.1f} % of the VALU stream; "
          f"moves, lane ids and integer / address work: {v - arith:.0f} = {100 * (v - arith) / v:.1f} %")


if __name__ == "__main__":
    main()
