#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing for the one hazard the compiler cannot see inside `asm volatile` MFMAs
(tools/ubench/valu_mfma_hazard.hip, measured: no hardware interlock): a vector-ALU instruction that writes a VGPR / AGPR an MFMA
reads as srcA / srcB must not be the instruction right in front of that MFMA (one wait state -- any instruction -- suffices).
Also reports non-MFMA readers of an MFMA result closer than 11 wait states (8-pass) / 7 (4-pass) behind it, counting s_nop N as N + 1.

usage: mfma_hazard_lint.py file.s [kernel-name-substring]      exit code 1 if a hazard is found"""
import re
import sys


def regs(tok):
    """'v[4:7]' -> {('v',4),...}; 'a3' -> {('a',3)}; anything else -> {}"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_smfmac")


def dst_regs(op, ops):
    if not ops:
        return set()
    if op.startswith(("v_cmp", "v_cmpx")):
        return set()
    d = regs(ops[0])
    if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64", "v_mad_i64", "v_div_scale")) and len(ops) > 1:
        d |= regs(ops[0])
    return d


def lint(path, key=None):
    lines = open(path).read().splitlines()
    if key:
        start = [k for k, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(";")[0].rstrip().endswith(":")]
        if not start:
            raise SystemExit(f"kernel {key} not found in {path}")
        end = [k for k in range(start[0], len(lines)) if ".amdhsa_kernel" in lines[k]][0]
        lines = lines[start[0] + 1:end]
    ins = [(n, p) for n, l in enumerate(lines) for p in [parse(l)] if p]
    bad = []
    n_mfma = 0
    for k, (n, (op, ops)) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        n_mfma += 1
        src = regs(ops[1]) | regs(ops[2])
        srcc = regs(ops[3]) if len(ops) > 3 else set()
        if k > 0:
            pop, pops = ins[k - 1][1]
            if is_valu(pop) and dst_regs(pop, pops) & src:
                bad.append((n, f"{pop} {', '.join(pops)}  ->  {op} {', '.join(ops)}   (vector write of an MFMA source in the instruction before it)"))
            if is_valu(pop) and dst_regs(pop, pops) & srcc:
                bad.append((n, f"{pop} {', '.join(pops)}  ->  {op} {', '.join(ops)}   (vector write of the MFMA's srcC in the instruction before it)"))
        # result read by a non-MFMA instruction too early
        need = 11 if "32x32x16" in op else 7 if "16x16x32" in op else 0
        d = regs(ops[0])
        ws = 0
        for j in range(k + 1, min(k + 14, len(ins))):
            qop, qops = ins[j][1]
            if ws >= need:
                break
            if qop.startswith("v_mfma"):
                ws += 1
                continue
            if qop == "s_nop":
                ws += int(qops[0]) + 1
                continue
            uses = set()
            for o in (qops[1:] if is_valu(qop) or qop.startswith(("ds_read", "global_load", "scratch_load")) else qops):
                uses |= regs(o)
            if (is_valu(qop) and dst_regs(qop, qops) & d) or (uses & d):
                bad.append((ins[j][0], f"{op} {ops[0]} ... then after {ws} wait states: {qop} {', '.join(qops)}   (needs {need})"))
                break
            ws += 1
    return n_mfma, bad


if __name__ == "__main__":
    n, bad = lint(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    print(f"{n} MFMAs checked, {len(bad)} hazards")
    for ln, msg in bad[:40]:
        print(f"  line {ln}: {msg}")
    sys.exit(1 if bad else 0)
