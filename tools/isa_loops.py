#!/usr/bin/env python3
"""Static look at one kernel of a `hipcc -S` listing: its innermost loops (a backward branch to a label with no other loop inside),
with per-loop instruction counts by class.  Usage: isa_loops.py file.s <symbol-substring> [min_insts]"""
import re, sys, collections

def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "fma64"
    if op.startswith("v_mul_f64"): return "mul64"
    if op.startswith("v_add_f64"): return "add64"
    if re.match(r"v_(rsq|rcp|sqrt)_f64", op): return "trans64"
    if re.match(r"v_(ldexp|rndne|fract|trunc|floor|ceil|max|min|cmp\w*|cmpx\w*|cvt_i32|cvt_f64\w*|div_\w+|frexp\w*)_f64", op) or op.endswith("_f64"): return "other64"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "mov"
    if "dpp" in op or op.startswith("v_permlane") or op.startswith("ds_swizzle") or op.startswith("ds_bpermute"): return "xlane"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"): return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "?"

def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels, insts = {}, []          # label -> inst index ; insts: (op, text, dpp)
    for l in body:
        s = l.split(";")[0].strip()
        if not s: continue
        m = re.match(r"^(\.L\w+):", s)
        if m: labels[m.group(1)] = len(insts); continue
        if s.startswith("."): continue
        op = s.split()[0]
        insts.append((op + ("_dpp" if "dpp" in s or "row_" in s else ""), s))
    loops = []
    for i, (op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i: loops.append((labels[tgt], i))
    inner = [lp for lp in loops if not any(o != lp and o[0] >= lp[0] and o[1] <= lp[1] for o in loops)]
    print("%s: %d instructions, %d loops, %d innermost" % (sym, len(insts), len(loops), len(inner)))
    for a, b in sorted(inner):
        n = b - a + 1
        if n < min_insts: continue
        c = collections.Counter(classify(op) for op, _ in insts[a:b + 1])
        valu = sum(v for k, v in c.items() if k not in ("lds", "vmem", "wait", "branch", "smem", "salu", "?"))
        f64 = c["fma64"] + c["mul64"] + c["add64"] + c["trans64"]
        print("  loop @%d..%d  %4d insts  VALU %4d  fp64(fma/mul/add/trans) %d/%d/%d/%d = %.2f of VALU  other64 %d lane %d mov %d xlane %d cnd %d valu_other %d | lds %d vmem %d salu %d wait %d" % (
            a, b, n, valu, c["fma64"], c["mul64"], c["add64"], c["trans64"], f64 / max(valu, 1), c["other64"], c["lane"], c["mov"], c["xlane"], c["cndmask"], c["valu_other"], c["lds"], c["vmem"], c["salu"], c["wait"]))
    if len(sys.argv) > 4:
        a, b = map(int, sys.argv[4].split(","))
        for op, s in insts[a:b + 1]: print("    ", s)

main()
