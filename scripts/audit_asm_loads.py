#!/usr/bin/env python3
"""Audit hand-issued LDS reads (and the ownership of M0) in the compiled kernels (guide section 5.7 item 1/4).

For every inline-asm `ds_read_b128 v[a:b], ...` (between ;;#ASMSTART / ;;#ASMEND) check that no
instruction READS or WRITES a register of v[a:b] before a later `s_waitcnt lgkmcnt(N)` that covers
the load (N small enough given the hand-issued reads issued after it).
For every inline-asm `global_store_dwordx4 vaddr, v[a:b], ...` check the hazard hipcc's recognizer cannot see
inside asm: a store of more than 64 bits needs 2 wait states before a VALU instruction overwrites its data
registers (gfx940+ "VMEM store more than 64 bits followed by a VALU write of vdata").
For every inline-asm VMEM instruction (global_* / buffer_*) with an SGPR operand check the other hazard hipcc cannot
see: a VALU write of an SGPR (v_readfirstlane / v_readlane, which is how a wave-uniform value reaches an "s"
constraint) needs 5 wait states before a VMEM instruction reads that SGPR.  Usage:
    hipcc ... -save-temps -c kernel.hip ;  python scripts/audit_asm_loads.py kernel-hip-amdgcn-*.s
"""
import re
import sys

# Every mnemonic an inline-asm statement of this library may contain, with the hazard classes it can take part in and
# which check below covers them (the compiler pads NOTHING whose producer or consumer sits inside an asm string;
# cdna_hip_programming.md section 5.7).  An asm mnemonic that is not listed FAILS the audit: a new instruction means a
# new row here first.
#   vmem-sgpr   VALU write of an SGPR (v_readfirstlane / v_readlane) -> VMEM reading it: 5 wait states      [checked]
#   store-data  VMEM store of > 64 bits -> VALU write of its data registers: 2 wait states                  [checked]
#   load-dest   asm-issued load: destination untouched until a covering s_waitcnt (lgkmcnt / vmcnt)         [checked]
#   mfma-d      MFMA result (VGPR or AGPR) -> asm instruction reading or writing it: the compiler's MFMA hazard
#               recognizer does not look inside asm; 18 wait states cover the 16-pass 32x32x2 / 32x32x16 forms [checked]
#   m0          M0 is written and read inside ONE statement (s_mov m0 / s_nop 0 / global_load_lds); nothing the
#               compiler emits may touch M0                                                                 [checked]
#   none        plain VALU on VGPR operands (VALU -> VALU needs no wait states), s_nop, s_waitcnt
ASM_MNEMONICS = {
    "s_mov_b32": ("m0",), "s_nop": ("none",), "s_waitcnt": ("none",),
    "global_load_lds_dwordx4": ("vmem-sgpr", "m0"),
    "global_store_dwordx4": ("vmem-sgpr", "store-data", "mfma-d"), "global_store_dword": ("vmem-sgpr", "mfma-d"),
    "global_load_dwordx4": ("vmem-sgpr", "load-dest"), "global_load_dword": ("vmem-sgpr", "load-dest"),
    "ds_read_b128": ("load-dest",), "ds_read_b32": ("load-dest",),
    "ds_write_b32": ("mfma-d",), "ds_write_b128": ("mfma-d",),
    "v_max_f32": ("mfma-d",), "v_min_u32": ("mfma-d",), "v_bfe_i32": ("mfma-d",), "v_fmac_f32": ("mfma-d",),
    "v_pk_max_i16": ("mfma-d",),
    # split-f16 kernel (mlp_forward_f16x2.hip): range tracking and the hand-written hi / lo split
    "v_max3_f32": ("mfma-d",), "v_cvt_pk_f16_f32": ("mfma-d",), "v_fma_mix_f32": ("mfma-d",),
    # reverse chain of the split-f16 kernel: the workgroup's per-plane |dY| maxima (LDS atomic, no return value)
    "ds_max_u32": ("mfma-d",),
}
MFMA_WAIT_STATES = 18


def regs(tok):
    tok = tok.lstrip("-").strip("|")          # source modifiers: -v3, |v9|, -|v9|
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def sregs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def audit(path):
    lines = open(path).read().splitlines()
    in_asm = False
    pending = []  # [regset, line_no, younger_asm_reads]
    stores = []   # [data regset, line_no, wait states seen since the store]
    sgpr_writes = []   # [sgpr set, line_no, wait states seen since the VALU write]
    mfma_d = []   # [vgpr set, agpr set, line_no, wait states since]: results of recent MFMAs
    vloads = []   # [dest regset, line_no, younger VMEM operations]: asm-issued global loads nobody has waited for yet
    n_vloads = 0
    n_vmem_s = 0
    problems = 0
    total = 0
    n_stores = 0
    for no, raw in enumerate(lines, 1):
        ln = raw.strip()
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if ln.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not ln or ln.startswith(";") or ln.startswith(".") or ln.endswith(":"):
            continue
        op, _, rest = ln.partition(" ")
        toks = [t.strip().rstrip(",") for t in re.split(r"[ ,]+", rest) if t.strip()]
        # ---- unknown asm mnemonics fail the build; MFMA results must not be touched by asm within the MFMA's shadow
        if in_asm:
            base_op = op[:-4] if op.endswith("_e32") or op.endswith("_e64") else op
            if base_op not in ASM_MNEMONICS:
                problems += 1
                print(f"{path}:{no}: `{ln[:70]}`: asm mnemonic `{op}` has no row in ASM_MNEMONICS (scripts/audit_asm_loads.py)")
            if mfma_d and not op.startswith("s_"):
                tv, ta = set(), set()
                for t in toks:
                    tv |= regs(t)
                    m_a = re.fullmatch(r"a\[(\d+):(\d+)\]", t) or re.fullmatch(r"a(\d+)", t)
                    if m_a:
                        ta |= set(range(int(m_a.group(1)), int(m_a.group(m_a.lastindex)) + 1))
                for d in mfma_d:
                    if (tv & d[0]) or (ta & d[1]):
                        problems += 1
                        print(f"{path}:{no}: `{ln[:70]}` touches the result of the MFMA at line {d[2]} after {d[3]} wait "
                              f"state(s) (needs {MFMA_WAIT_STATES}: the compiler pads nothing inside asm)")
        if op == "s_branch":     # an unconditional jump: what follows in the FILE does not follow in time
            mfma_d = []
        m_n = re.fullmatch(r"s_nop (\d+)", ln)
        for d in mfma_d:
            d[3] += int(m_n.group(1)) + 1 if m_n else (16 if op.startswith("v_mfma") else 1)
        mfma_d = [d for d in mfma_d if d[3] < MFMA_WAIT_STATES]
        if op.startswith("v_mfma") and toks:
            dv, da = regs(toks[0]), set()
            m_a = re.fullmatch(r"a\[(\d+):(\d+)\]", toks[0])
            if m_a:
                da = set(range(int(m_a.group(1)), int(m_a.group(2)) + 1))
            mfma_d.append([dv, da, no, 0])
        # ---- asm-issued global loads (mlp_layered.hip: B operands, ReLU masks): the compiler believes the destination valid
        # from the moment of issue; nothing may read or write it before a vmcnt wait that covers the load (VMEM returns
        # in order: vmcnt(N) leaves at most the N youngest operations outstanding)
        m_vm = re.search(r"vmcnt\((\d+)\)", ln)
        if op == "s_waitcnt" and m_vm:
            n_left = int(m_vm.group(1))
            vloads = [v for v in vloads if v[2] < n_left]
        elif vloads:
            is_vmem = op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_")
            if not (in_asm and op.startswith("global_load")):
                touched_v = set()
                for t in toks:
                    touched_v |= regs(t)
                if is_vmem and toks and not op.startswith("global_store") and not op.startswith("buffer_store") and not op.startswith("scratch_store"):
                    pass
                for v in vloads:
                    if touched_v & v[0]:
                        problems += 1
                        print(f"{path}:{no}: `{ln[:70]}` touches v{sorted(touched_v & v[0])[:4]} of the asm-issued global load at "
                              f"line {v[1]} before a covering s_waitcnt vmcnt")
                        v[0] -= touched_v
            if is_vmem:
                for v in vloads:
                    v[2] += 1
        if in_asm and op in ("global_load_dwordx4", "global_load_dword") and toks and not ln.rstrip().endswith("lds"):
            vloads.append([regs(toks[0]), no, 0])
            n_vloads += 1
        # ---- wide-store data hazard: 2 wait states before a VALU write of the stored registers
        if stores:
            if op.startswith("v_") and toks:
                for st in stores:
                    if regs(toks[0]) & st[0]:
                        problems += 1
                        print(f"{path}:{no}: `{ln[:70]}` overwrites the data of the asm-issued store at line {st[1]} "
                              f"after {st[2]} wait state(s) (needs 2)")
            m_nop = re.fullmatch(r"s_nop (\d+)", ln)
            for st in stores:
                st[2] += int(m_nop.group(1)) + 1 if m_nop else 1
            stores = [st for st in stores if st[2] < 2]
        # ---- VALU write of an SGPR -> asm VMEM read of it: 5 wait states
        if in_asm and (op.startswith("global_") or op.startswith("buffer_")):
            used = set()
            for t in toks:
                used |= sregs(t)
            if used:
                n_vmem_s += 1
            for w in sgpr_writes:
                if used & w[0]:
                    problems += 1
                    print(f"{path}:{no}: `{ln[:70]}` reads s{sorted(used & w[0])} {w[2]} wait state(s) after the VALU "
                          f"write at line {w[1]} (needs 5)")
        m_nop2 = re.fullmatch(r"s_nop (\d+)", ln)
        for w in sgpr_writes:
            w[2] += int(m_nop2.group(1)) + 1 if m_nop2 else 1
        sgpr_writes = [w for w in sgpr_writes if w[2] < 5]
        if op in ("v_readfirstlane_b32", "v_readlane_b32") and toks:
            sgpr_writes.append([sregs(toks[0]), no, 0])
        elif op.startswith("s_") and toks and not in_asm and op not in ("s_waitcnt", "s_nop", "s_barrier"):
            for w in sgpr_writes:      # an SALU result replaces the VALU-written value: SALU -> VMEM has no hazard
                w[0] -= sregs(toks[0])
        if in_asm and op == "global_store_dwordx4" and len(toks) >= 2:
            stores.append([regs(toks[1]), no, 0])
            n_stores += 1
        if not in_asm and re.search(r"\bm0\b", ln):
            # the LDS-DMA pieces overwrite M0 without restoring it: nothing the compiler emits may depend on it
            problems += 1
            print(f"{path}:{no}: `{ln[:70]}` uses m0 outside the inline asm that owns it")
        if op.startswith("ds_") and (op.startswith("ds_read_b128") or not in_asm or True):           # every LDS operation is a younger LGKM op for the pending reads
            for p in pending:
                p[2] += 1
            if in_asm and op in ("ds_read_b128", "ds_read_b32"):
                pending.append([regs(toks[0]), no, 0])
                total += 1
            continue
        m = re.search(r"lgkmcnt\((\d+)\)", ln)
        if op == "s_waitcnt" and m:
            n = int(m.group(1))
            # LDS returns in order: after the wait at most the n youngest operations are outstanding
            pending = [p for p in pending if p[2] < n]
            continue
        touched = set()
        for t in toks:
            touched |= regs(t)
        for p in pending:
            if touched & p[0]:
                problems += 1
                print(f"{path}:{no}: `{ln[:70]}` touches v{sorted(touched & p[0])} of the un-waited read at line {p[1]}")
    print(f"{path}: {n_stores} asm-issued wide stores, {n_vmem_s} asm-issued VMEM instructions with SGPR operands, "
          f"{n_vloads} asm-issued global loads checked")
    print(f"{path}: {total} hand-issued LDS reads, {problems} problems")
    return problems


if __name__ == "__main__":
    sys.exit(1 if sum(audit(p) for p in sys.argv[1:]) else 0)
