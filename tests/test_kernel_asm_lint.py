"""Static checks on the gfx950 code hipcc generates for the hand-pipelined kernels (no GPU needed: hipcc cross-compiles).

The pipelines issue their MFMAs, LDS reads and LDS-DMA through inline asm, which hipcc's hazard recogniser and waitcnt
insertion cannot see into.  Two failure classes found on hardware this round are checked here on the assembly:

* register spills: a reload inside a stream makes hipcc emit an ``s_waitcnt vmcnt(n)`` that also waits for the weight
  DMA it does not know about (2 300 cycles per head), and spill traffic showed up as 210 MB per launch;
* VALU write -> MFMA operand read without wait states (an MFMA issued right behind the ``v_cvt_pk`` of its B operand read
  the old register contents: NaNs).  ``s_nop 3`` (4 wait states) is the distance verified on the MI355X.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
VALU_TO_MFMA_WAIT_STATES = 4


def _compile(src: str, tmp_path) -> str:
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o",
                    str(out), os.path.join(ROOT, "layout_dm_amd", "csrc", src), "-Wno-unused-function"],
                   check=True, capture_output=True, text=True, timeout=900)
    return out.read_text()


def _kernels(asm: str):
    """name -> list of instruction strings, for every kernel (functions with an .amdhsa_kernel descriptor).  Instructions
    that come from inline asm (between hipcc's ;;#ASMSTART / ;;#ASMEND markers) carry the prefix "asm:"."""
    names = set(re.findall(r"\.amdhsa_kernel\s+(\S+)", asm))
    out, cur, in_asm = {}, None, False
    for line in asm.splitlines():
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\S+):\s*(;.*)?$", t)
        if m and m.group(1) in names:
            cur = out.setdefault(m.group(1), [])
            continue
        if t.startswith(".Lfunc_end"):
            cur = None
        if cur is None or not t or t[0] in ";." or t.endswith(":"):
            continue
        cur.append(("asm:" if in_asm else "") + t.split(";")[0].strip())
    return out


def _regs(tok: str):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


@pytest.mark.parametrize("src,kernel_substr", [("kernels_stack.hip", "stack_stream_k")])
def test_stream_kernels_no_spills_no_valu_mfma_hazard(tmp_path, src, kernel_substr):
    asm = _compile(src, tmp_path)
    kernels = {k: v for k, v in _kernels(asm).items() if kernel_substr in k}
    assert kernels, "no kernel found"
    # every variant without the probe instrumentation (TM = false: the first template argument mangles as Lb0) must be
    # free of scratch; the probe variants may spill a couple of registers around their timers
    sizes = dict(re.findall(r"\.amdhsa_kernel\s+(\S+)[\s\S]*?\.amdhsa_private_segment_fixed_size\s+(\d+)", asm))
    for name, instr in kernels.items():
        product = "ILb0E" in name
        loop = "ILb0ELi2E" in name  # HEAD == 2: the whole reverse loop in the workgroup
        if product and not loop:
            assert int(sizes[name]) == 0, f"{name}: {sizes[name]} bytes of scratch per lane"
            assert not [i for i in instr if i.startswith("scratch_")], f"{name}: scratch instructions"
        if loop:
            # a handful of loop-carried values are parked in scratch between phases (hipcc keeps what it hoists out of
            # the step loop alive across a body that fills the register file); what must not happen is a spill or a
            # reload INSIDE a hand-issued stream, where hipcc's s_waitcnt vmcnt(n) for it would also wait for the
            # weight DMA it cannot see
            assert int(sizes[name]) <= 64, f"{name}: {sizes[name]} bytes of scratch per lane"
            mfma = [i for i, t in enumerate(instr) if t.startswith("asm:v_mfma")]
            for i, t in enumerate(instr):
                if t.startswith("scratch_"):
                    near = min(abs(i - j) for j in mfma)
                    assert near > 24, f"{name}: '{t}' {near} instructions from a hand-issued MFMA"
        n_mfma = 0
        for i, t in enumerate(instr):
            # only MFMAs issued through inline asm: for its own (builtin) MFMAs hipcc places the wait states itself
            if not t.startswith("asm:v_mfma"):
                continue
            t = t[4:]
            n_mfma += 1
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            src_regs = _regs(ops[1]) | _regs(ops[2])
            ws = 0
            for j in range(i - 1, max(i - 10, -1), -1):
                p = instr[j][4:] if instr[j].startswith("asm:") else instr[j]
                op = p.split()[0]
                if op == "s_nop":
                    ws += int(p.split()[1]) + 1
                    continue
                if op.startswith(("ds_read", "global_load", "scratch_load", "buffer_load")):
                    if _regs(p.split(None, 1)[1].split(",")[0].strip()) & src_regs:
                        break  # the operand is (re)defined by a load: covered by the counted waits, not by this rule
                if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp", "v_accvgpr_write")):
                    if _regs(p.split(None, 1)[1].split(",")[0].strip()) & src_regs:
                        assert ws >= VALU_TO_MFMA_WAIT_STATES, \
                            f"{name}: '{p}' writes an operand of '{t}' only {ws} wait states ahead"
                        break
                # an intervening MFMA holds the matrix pipe for 8 passes: the checked MFMA cannot issue before it has left
                ws += 8 if op.startswith("v_mfma") else 1
                if ws >= VALU_TO_MFMA_WAIT_STATES:
                    break
        assert n_mfma > 150, f"{name}: only {n_mfma} inline-asm MFMAs found (parser broken?)"


def test_step_tail_kernels_use_no_scratch(tmp_path):
    """The other launch of every reverse step (posterior + draw + next step's embedding) and the result packaging: no
    scratch either — a scratch-using kernel between the scratch-free stack kernels slowed the graph-replayed two-lane loop
    (profiles/r02_call35_37_*: the same effect as observed with the wide fp32 GEMM tile)."""
    for src, names in (("kernels_post.hip", ("posterior_sample_k",)), ("kernels_decode.hip", ("decode_layouts_k",))):
        asm = _compile(src, tmp_path)
        for name in names:
            blocks = re.findall(r"\.amdhsa_kernel\s+(\S*%s\S*)(.*?)\.end_amdhsa_kernel" % name, asm, flags=re.S)
            assert blocks, name
            for sym, body in blocks:
                m = re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body)
                assert m and int(m.group(1)) == 0, (sym, m and m.group(1))


def test_exact_mode_kernels_resources(tmp_path):
    """The exact mode's fp32 kernels: no scratch, and the register / LDS footprints their schedules were measured at —
    the 128 x 128 GEMM tile at <= 102 VGPRs and 32 KB of LDS (5 workgroups per CU), its 64-row and 160-wide variants and the
    fp32 attention (K alone in LDS: 30 KB, <= 170 VGPRs: 3 workgroups per CU).  The GEMMs' operands arrive by LDS-DMA."""
    want = {
        "gemm_f32_tileILi2E": dict(src="kernels_gemm.hip", vgpr=102, lds=32 * 1024),
        "gemm_f32_tileILi1E": dict(src="kernels_gemm.hip", vgpr=85, lds=24 * 1024),
        "gemm_f32_128x160": dict(src="kernels_gemm.hip", vgpr=128, lds=36 * 1024),
        "attn32_direct_kILi29E": dict(src="kernels_attn.hip", vgpr=170, lds=0),  # (dynamic LDS: 128 x 59 floats at launch)
    }
    asm_of = {}
    for key, w in want.items():
        asm = asm_of.setdefault(w["src"], _compile(w["src"], tmp_path))
        blocks = re.findall(r"\.amdhsa_kernel\s+(\S*%s\S*)(.*?)\.end_amdhsa_kernel" % key, asm, flags=re.S)
        assert len(blocks) == 1, (key, [b[0] for b in blocks])
        sym, body = blocks[0]
        get = lambda field: int(re.search(r"\.amdhsa_%s\s+(\d+)" % field, body).group(1))
        assert get("private_segment_fixed_size") == 0, sym
        assert get("next_free_vgpr") <= w["vgpr"], (sym, get("next_free_vgpr"))
        assert get("group_segment_fixed_size") == w["lds"], (sym, get("group_segment_fixed_size"))
        if "gemm_f32" in key:
            ins = _kernels(asm)[sym]
            assert sum("global_load_lds_dwordx4" in i for i in ins) >= 3, sym
            assert not any(i.startswith("ds_write") for i in ins), sym  # no register staging left


def test_split_mode_kernels_resources(tmp_path):
    """The split mode's kernels (r04): the production GEMM instantiations (256 x 256 tiles on 8 waves: 128 accumulator
    registers per lane, <= 256 registers = two waves per SIMD, operands by LDS-DMA only, no scratch) and the fp16 x 3
    attention (<= 170 VGPRs: three workgroups per CU; no scratch: a `cond ? *p : zero` load once compiled into a flat load
    of a select between the global pointer and a STACK zero)."""
    asm = _compile("kernels_gemm16.hip", tmp_path)
    blocks = re.findall(r"\.amdhsa_kernel\s+(\S*gemm16x3_kILi256ELi256ELi32ELi2ELi2ELi4ELi[0-3]ELi0ELi0E\S*)(.*?)\.end_amdhsa_kernel", asm, flags=re.S)
    assert len(blocks) == 4, [b[0] for b in blocks]
    get = lambda body, field: int(re.search(r"\.amdhsa_%s\s+(\d+)" % field, body).group(1))
    for sym, body in blocks:
        assert get(body, "private_segment_fixed_size") == 0, sym
        total = get(body, "next_free_vgpr")  # (unified register file: arch + accumulation registers)
        assert total <= 256, (sym, total)
        ins = _kernels(asm)[sym]
        assert sum("global_load_lds_dwordx4" in i for i in ins) >= 8, sym
        assert sum("v_mfma_f32_32x32x16_f16" in i or "v_mfma_f32_32x32x16f16" in i for i in ins) >= 48, sym
    asm = _compile("kernels_attn16.hip", tmp_path)
    blocks = re.findall(r"\.amdhsa_kernel\s+(\S*attn16x3_k\S*)(.*?)\.end_amdhsa_kernel", asm, flags=re.S)
    assert len(blocks) == 1
    sym, body = blocks[0]
    assert get(body, "private_segment_fixed_size") == 0, sym
    assert get(body, "next_free_vgpr") <= 170, (sym, get(body, "next_free_vgpr"))
    ins = _kernels(asm)[sym]
    assert not any(i.startswith(("flat_load", "scratch_")) for i in ins), sym
    assert sum("v_mfma" in i for i in ins) == 96, sym  # 2 x (4 x 4 | 8 x 2) fragment pairs x 3 products


def test_lngemm_kernel_no_scratch_and_mfma_hazards(tmp_path):
    """kernels_lngemm.hip (r05): no scratch; no VALU write of an inline-asm MFMA operand inside the verified wait states; and the
    failure class r05 found on hardware — hipcc reading a tile accumulator right behind the asm MFMAs it cannot recognise (two
    builds lost the low-order products that way: logits error 5e-5 instead of 9e-7) — every non-MFMA instruction that READS the
    destination registers of an inline-asm MFMA must sit at least one whole MFMA (8 passes = 32 cycles) of wait states behind it."""
    asm = _compile("kernels_lngemm.hip", tmp_path)
    # the product instantiations (TM = false, ABL = 0: template arguments 3 and 4 mangle as Lb0ELi0; the phase-timer and measurement builds are dev only)
    NAME = re.compile(r"ELb0ELi0ELb([01])ELi([123])ELi([123])EEEvNS_10LnGemmArgsE$")   # ... PRE, NPM, NPP> (products per k16-step: tile loop, GEMM prologue)
    kernels = {k: v for k, v in _kernels(asm).items() if "lngemm16x3_k" in k and NAME.search(k)}
    # (ADA, OUT) in {(1, 0), (0, 1), (0, 0), (1, 2)} x {plain, with the GEMM prologue} x {three products (split), two (mixed: weights fp16 only)};
    # OUT = 2 (r06): in_proj writing hi / lo q / k / v panels; + the hybrid mode's three: linear2 in plain fp16 in front of the two-product in_proj,
    # linear1 in plain fp16 writing plain-fp16 panels (OUT = 3), linear2 + head in plain fp16, the head alone in plain fp16 (behind the fused fp16 FFN)
    assert len(kernels) == 20, list(_kernels(asm))
    sizes = dict(re.findall(r"\.amdhsa_kernel\s+(\S+)[\s\S]*?\.amdhsa_private_segment_fixed_size\s+(\d+)", asm))
    for name, instr in kernels.items():
        assert int(sizes[name]) == 0 and not [i for i in instr if i.startswith("scratch_")], name
        mf = [i for i, t in enumerate(instr) if t.startswith("asm:v_mfma")]
        m = NAME.search(name)
        pre, per, ppre = m.group(1) == "1", int(m.group(2)), int(m.group(3))   # products per k16-step: tile loop / GEMM prologue
        # tile body: 29 k16-steps x per products; the GEMM prologue adds three stage bodies (one per A register set) of 30 items x ppre
        assert len(mf) == per * 29 + (ppre * 90 if pre else 0), (name, len(mf))
        if pre:
            # the matrix pipe runs its queue in order: the last stage's final MFMAs are still in flight when the stage loop falls through, and
            # hipcc (which cannot see asm MFMAs) puts its v_accvgpr_reads of the accumulator tiles right there — the wait states must sit inside
            # the asm statement of the stage's last item, in ALL THREE stage bodies (r05 call 24: logits error 3e-2 without them)
            # the stage barrier's COUNTED wait (vmcnt(4)) is only right if the four youngest vector-memory operations in front of it are the
            # A loads of stage + 2 — plain loads hipcc schedules — and everything older is a DMA piece: nothing else may sit between them
            n_a = 2 if ppre == 1 else 4      # A loads per stage: hi + lo rows of two k16-steps, or (one product: plain fp16) hi only
            waits = [i for i, t in enumerate(instr) if t == f"asm:s_waitcnt vmcnt({n_a})"]
            assert len(waits) == 3, (name, len(waits))
            for w in waits:
                vm = [t for t in instr[max(0, w - 400):w] if t.split()[0].replace("asm:", "") in
                      ("global_load_dwordx4", "global_load_lds_dwordx4", "global_load_dwordx2", "global_load_dword", "global_store_dwordx4",
                       "global_store_dwordx2", "global_store_dword", "scratch_load_dword", "scratch_store_dword", "scratch_load_dwordx4",
                       "scratch_store_dwordx4")]
                assert [t.split()[0] for t in vm[-n_a:]] == ["global_load_dwordx4"] * n_a, (name, vm[-6:])
                assert vm[-n_a - 1].startswith("asm:global_load_lds_dwordx4"), (name, vm[-6:])
            for last in (mf[30 * ppre - 1], mf[60 * ppre - 1], mf[90 * ppre - 1]):
                tail = instr[last + 1:last + 8]
                assert tail.count("asm:s_nop 15") == 2, (name, tail)
                assert not any(t.startswith("v_accvgpr_read") for t in tail), (name, tail)
        for i in mf:
            ops = [o.strip() for o in instr[i][4:].split(None, 1)[1].split(",")]
            dst = _regs(ops[0])
            ws = 0
            for j in range(i + 1, min(i + 200, len(instr))):
                p = instr[j][4:] if instr[j].startswith("asm:") else instr[j]
                op = p.split()[0]
                if op == "s_nop":
                    ws += int(p.split()[1]) + 1
                    continue
                if op.startswith("v_mfma"):
                    o2 = [o.strip() for o in p.split(None, 1)[1].split(",")]
                    if _regs(o2[0]) & dst:
                        break              # the same accumulator continues (or is re-initialised): the matrix pipe orders that itself
                    ws += 8
                    continue
                if op == "s_branch":
                    break                  # control leaves this stretch of text: what follows textually is another block (the loop tail)
                if op in ("s_barrier", "s_waitcnt", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz",
                          "s_cbranch_execz", "s_cbranch_execnz"):
                    ws += 1
                    continue
                args = p.split(None, 1)[1].split(",") if " " in p else []
                reads = set()
                for a_ in args[1:] if not op.startswith(("ds_write", "global_store", "ds_read")) else args:
                    reads |= _regs(a_.strip().split(" ")[0])
                if reads & dst:
                    assert ws >= 32, f"{name}: '{p}' reads the destination of '{instr[i][4:]}' only {ws} wait states behind it"
                    break
                ws += 1
                if ws >= 64:
                    break


def test_attnout_kernel_register_discipline_and_hazards(tmp_path):
    """kernels_attnout.hip (r06): every MFMA is inline asm with a pinned accumulator, q fragments arrive by asm global loads hipcc does
    not track.  Checked on the ISA of the product instantiation (TM = false):
      * no scratch; 276 asm MFMAs per head iteration + 48 in the prologue (scores of head 0);
      * the head loop moves NOTHING between the register files (the first builds did: hipcc put the score tiles in AGPRs and shuffled
        out_proj tiles through VGPRs, ~450 v_accvgpr_read / _write per head);
      * the registers an asm global load writes are not touched before a counted s_waitcnt vmcnt(N) that covers the load (loads
        return in order: covered = at least N vector-memory operations were issued behind it);
      * no VALU write of an asm MFMA's operand inside the verified wait states (hipcc had sunk the fp16 casts of the attention output to
        ONE instruction in front of the MFMA that consumes them);
      * a VALU read of an asm MFMA's VGPR result sits at least a whole MFMA behind it."""
    asm = _compile("kernels_attnout.hip", tmp_path)
    ks = {k: v for k, v in _kernels(asm).items() if "attnout16x3_k" in k}
    names = sorted(k for k in ks if "ILb0E" in k)   # <TM = false, W2 = false | true, FFN = false>: three products in out_proj, or two (Wo fp16 only)
    assert len(names) == 3 and "ILb0ELb0ELb0E" in names[0] and "ILb0ELb1ELb0E" in names[1] and "ILb0ELb1ELb1E" in names[2], list(ks)
    for name in names[:2]:
        _attnout_checks(asm, name, ks[name], 276 if "ILb0ELb0E" in name else 216)
    # <false, true, true> (the hybrid mode): the block's plain-fp16 FFN behind the attention in the same launch — ldm_pipes.h FfnStream on the same 15
    # AGPR tiles (its second GEMM is a builtin MFMA): no scratch, 59 more MFMAs, and neither loop moves anything between the register files
    instr = ks[names[2]]
    sizes = dict(re.findall(r"\.amdhsa_kernel\s+(\S+)[\s\S]*?\.amdhsa_private_segment_fixed_size\s+(\d+)", asm))
    assert int(sizes[names[2]]) == 0 and not [i for i in instr if i.startswith("scratch_")]
    mf = [i for i, t in enumerate(instr) if "v_mfma_f32_32x32x16" in t]
    assert len(mf) == 48 + 216 + 59, len(mf)
    for lo, hi in ((mf[48], mf[48 + 215]), (mf[48 + 216], mf[-1])):
        assert not [t for t in instr[lo:hi + 1] if t.startswith(("v_accvgpr_read", "v_accvgpr_write", "v_accvgpr_mov"))]


def _attnout_checks(asm, name, instr, per_head):
    sizes = dict(re.findall(r"\.amdhsa_kernel\s+(\S+)[\s\S]*?\.amdhsa_private_segment_fixed_size\s+(\d+)", asm))
    assert int(sizes[name]) == 0 and not [i for i in instr if i.startswith("scratch_")]
    mf = [i for i, t in enumerate(instr) if t.startswith("asm:v_mfma")]
    assert len(mf) == 48 + per_head, len(mf)
    assert not [t for t in instr if t.startswith("v_mfma")], "a builtin MFMA: hipcc would choose its register file"
    # ---- the head loop: between the first and the last MFMA of the 276
    loop = instr[mf[48]:mf[-1] + 1]
    acc = [t for t in loop if t.startswith(("v_accvgpr_read", "v_accvgpr_write", "v_accvgpr_mov"))]
    assert not acc, f"{len(acc)} AGPR <-> VGPR moves inside the head loop, e.g. {acc[:3]}"
    # ---- asm loads (global_load_dwordx4 vdst, voff, s[base]) vs the counted waits
    VM = ("global_load", "global_store", "buffer_load", "buffer_store", "scratch_load", "scratch_store", "flat_load", "flat_store")
    pending, n_asm_loads = {}, 0
    for t in instr:
        p = t[4:] if t.startswith("asm:") else t
        op = p.split()[0]
        if op.startswith(VM):
            for r in pending:
                pending[r] += 1
        if t.startswith("asm:global_load_dwordx4"):
            n_asm_loads += 1
            for r in _regs(p.split(None, 1)[1].split(",")[0].strip()):
                pending[r] = 0
            continue
        m = re.search(r"vmcnt\((\d+)\)", p) if op == "s_waitcnt" else None
        if m:
            n = int(m.group(1))
            pending = {r: c for r, c in pending.items() if c < n}
            continue
        if op.startswith("s_") or " " not in p:
            continue
        touched = set()
        for a_ in p.split(None, 1)[1].split(","):
            touched |= _regs(a_.strip().split(" ")[0])
        bad = touched & set(pending)
        assert not bad, f"'{p}' touches {sorted(bad)[:4]} while the asm load that writes them may still be in flight"
    assert n_asm_loads == 16, n_asm_loads          # Q fragments of head 0 (prologue) and of head h + 1 (loop): 8 + 8
    # ---- VALU write -> asm MFMA operand read; asm MFMA VGPR result -> VALU read
    for i in mf:
        t = instr[i][4:]
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        src_regs = _regs(ops[1]) | _regs(ops[2])
        ws = 0
        for j in range(i - 1, max(i - 12, -1), -1):
            p = instr[j][4:] if instr[j].startswith("asm:") else instr[j]
            op = p.split()[0]
            if op == "s_nop":
                ws += int(p.split()[1]) + 1
                continue
            if op.startswith("v_") and not op.startswith(("v_mfma", "v_cmp")) and " " in p:
                if _regs(p.split(None, 1)[1].split(",")[0].strip()) & src_regs:
                    assert ws >= VALU_TO_MFMA_WAIT_STATES, f"'{p}' writes an operand of '{t}' only {ws} wait states ahead"
                    break
            ws += 8 if op.startswith("v_mfma") else 1
            if ws >= VALU_TO_MFMA_WAIT_STATES:
                break
        dst = _regs(ops[0])
        if not any(f == "v" for f, _ in dst):
            continue                                # the AGPR tiles are only read behind the loop's s_nop block
        ws = 0
        for j in range(i + 1, min(i + 60, len(instr))):
            p = instr[j][4:] if instr[j].startswith("asm:") else instr[j]
            op = p.split()[0]
            if op == "s_nop":
                ws += int(p.split()[1]) + 1
                continue
            if op.startswith("v_mfma"):
                if _regs(p.split(None, 1)[1].split(",")[0].strip()) & dst:
                    break                           # the accumulator chain continues: ordered by the matrix pipe
                ws += 8
                continue
            if op.startswith(("s_", "ds_", "global_")) or " " not in p:
                ws += 1
                continue
            reads = set()
            for a_ in p.split(None, 1)[1].split(",")[1:]:
                reads |= _regs(a_.strip().split(" ")[0])
            if reads & dst:
                assert ws >= 16, f"'{p}' reads the result of '{t}' only {ws} wait states behind it"
                break
            ws += 1
            if ws >= 32:
                break


def test_ffn16_rows_kernel_resources(tmp_path):
    """kernels_ffn16.hip (r06, the hybrid mode's fused plain-fp16 FFN): the fast mode's chunk stream (ldm_pipes.h FfnStream) as a row kernel — no
    scratch, all of a CU's LDS budget respected, 59 MFMAs per chunk iteration (29 + 30), and nothing moves between the register files inside the
    chunk loop (the 15 residual tiles live in AGPRs from the first load to the last store)."""
    asm = _compile("kernels_ffn16.hip", tmp_path)
    blocks = re.findall(r"\.amdhsa_kernel\s+(\S*ffn16_rows_k\S*)(.*?)\.end_amdhsa_kernel", asm, flags=re.S)
    assert len(blocks) == 1
    sym, body = blocks[0]
    get = lambda field: int(re.search(r"\.amdhsa_%s\s+(\d+)" % field, body).group(1))
    assert get("private_segment_fixed_size") == 0 and get("next_free_vgpr") <= 512
    ins = _kernels(asm)[sym]
    assert not [t for t in ins if t.startswith("scratch_")]
    mf = [i for i, t in enumerate(ins) if "v_mfma_f32_32x32x16" in t]
    assert len(mf) == 59, len(mf)
    loop = ins[mf[0]:mf[-1] + 1]
    assert not [t for t in loop if t.startswith(("v_accvgpr_read", "v_accvgpr_write", "v_accvgpr_mov"))]
    assert sum("global_load_lds_dwordx4" in t for t in loop) == 16      # one 64-KiB stage per iteration: 16 pieces per wave
