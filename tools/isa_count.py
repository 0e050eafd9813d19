"""Instruction counts of the MC kernel's tile loops FROM THE ISA ("report instructions/point from the ISA,
not by hand").  Compiles cornell_moe_amd/csrc/kg_mc_dp8.hip to gfx950 assembly (hipcc -S, the product flags), finds the
innermost loops of kg_mc_kernel<8, 0, true, false> -- the headline instantiation -- and, for each loop that evaluates the
Matern kernel (one v_rsq_f64 per tile of 64 points), prints the VALU / FP64 / LDS instructions per tile, i.e. per point and
pass.  Runs on the CPU box (hipcc cross-compiles).   python tools/isa_count.py [> profiles/rNN_isa_counts.txt]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cornell_moe_amd import build as moe_build  # noqa: E402

KERNEL = "_ZN3moe2mc12kg_mc_kernelILi8ELi0ELb1ELb0EEEvNS_10KgMcParamsE"


def main():
    tmp = tempfile.mkdtemp(prefix="moe_isa_")
    asm = os.path.join(tmp, "kg_mc_dp8.s")
    flags = [f for f in moe_build.FLAGS if f != "-fPIC"]
    subprocess.check_call([moe_build._hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(moe_build.CSRC, "kg_mc_dp8.hip"),
                                                          "-o", asm], stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    meta = {}
    for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size"):
        for i, l in enumerate(lines):
            if ".name:" in l and KERNEL in l:
                for m in lines[i:i + 12]:
                    if key in m:
                        meta[key] = m.split(":")[1].strip()
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i

    def isinst(l):
        l = l.strip()
        return bool(l) and not l.startswith((".", ";", "_")) and not l.endswith(":")

    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    print("kernel kg_mc_kernel<8, 0, true, false> (gfx950): %d instructions, %s VGPRs, %s SGPRs, %s bytes of scratch" % (
        sum(1 for l in body if isinst(l)), meta.get("vgpr_count"), meta.get("sgpr_count"), meta.get("private_segment_fixed_size")))
    print("%-34s %6s %6s %6s %6s %6s %6s   per tile of 64 points: %s" % ("tile loop", "tiles", "insts", "VALU", "FP64", "trans", "LDS",
                                                                         "VALU  FP64  non-FP64  LDS"))
    for a, b in sorted(inner):
        seg = [l.strip().split()[0] for l in body[a:b + 1] if isinst(l)]
        c = collections.Counter(seg)
        rsq = sum(v for k, v in c.items() if k.startswith("v_rsq_f64"))
        ldexp = sum(v for k, v in c.items() if k.startswith("v_ldexp_f64"))
        tiles = max(rsq, ldexp)
        if tiles == 0 or (rsq and rsq != ldexp):  # (not a covariance tile loop)
            continue
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        f64 = sum(v for k, v in c.items() if k.startswith("v_") and "_f64" in k)
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        kind = ("Matern" if rsq else "sq.exp.") + (" value+gradient" if c.get("v_add_f64", 0) >= 8 * tiles else " value")
        # multi-trial value loops (eval_multi_loop): T Armijo trials share one set of coordinate loads -- fewer than 9 LDS reads per
        # covariance evaluation; 'tiles' then counts tile x trial pairs
        if "gradient" not in kind and lds < 9 * tiles:
            kind += " x T trials"
        print("%-34s %6d %6d %6d %6d %6d %6d   %28.2f %5.2f %8.2f %5.2f" % (
            "%s (asm lines %d-%d)" % (kind, a + start, b + start), tiles, len(seg), valu, f64, rsq, lds, valu / tiles, f64 / tiles,
            (valu - f64) / tiles, lds / tiles))
    print("\n(one VALU instruction processes 64 points: 'per tile' = per point and pass ('x T trials': per point and trial -- the 'tiles' column counts tile x trial pairs of one iteration).  SURVEY 8(d) credits a value pass with 3d + 32"
          " = 56 flops per point and a value+gradient pass with 5d + 34 = 74 at d = 8; an FP64 FMA slot is worth 2.)")


if __name__ == "__main__":
    main()
