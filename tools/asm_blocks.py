#!/usr/bin/env python3
"""Basic-block census of one kernel in a hipcc -S listing: per block the instruction mix (VOP3-encoded integer ops are
half rate on gfx950) and the backward branches that close loops.
   tools/asm_blocks.py build/asm/nhip_bnb.s 'csm_bnb_kernelILi2ELb1ELb1E' [min_instrs]"""
import re
import sys

VOP3 = ("v_mad_u32_u24", "v_alignbit", "v_alignbyte", "v_bfe", "v_perm", "v_add3", "v_lshl_add", "v_and_or", "v_or3",
        "v_lshl_or", "v_mad_", "v_bfi", "v_med3", "v_max3", "v_min3", "v_mul_lo", "v_mul_hi", "v_lshlrev_b64",
        "v_lshrrev_b64", "v_readlane", "v_writelane", "v_mbcnt", "v_add_co", "v_addc_co", "v_xad",
        "v_fma_", "v_div", "v_ldexp", "v_mad_u64", "v_add_lshl", "v_sub_co", "v_subb")


def main():
    path, name = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, order = {"entry": []}, "entry", ["entry"]
    label_line = {"entry": start}
    for i in range(start + 1, end + 1):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            label_line[cur] = i
            continue
        if not l or l.startswith((";", ".", "//")):
            continue
        blocks[cur].append(l.split(";")[0].strip())
    idx = {b: k for k, b in enumerate(order)}
    tot = {}
    for b in order:
        c = dict(n=len(blocks[b]), valu=0, vop3=0, salu=0, lds=0, vmem=0, dpp=0, wait=0, back="")
        for s in blocks[b]:
            op = s.split()[0]
            if op.startswith("v_"):
                c["valu"] += 1
                if op.startswith(VOP3) or op.endswith("_e64"):
                    c["vop3"] += 1
                if "dpp" in s or "row_" in s:
                    c["dpp"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif op.startswith("s_waitcnt"):
                c["wait"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                t = s.split()[-1]
                if t in idx and idx[t] <= idx[b]:
                    c["back"] += " ->%s(%d blocks)" % (t, idx[b] - idx[t] + 1)
        tot[b] = c
    print("kernel lines %d..%d, %d blocks, %d instructions" % (start, end, len(order), sum(c["n"] for c in tot.values())))
    for b in order:
        c = tot[b]
        if c["n"] >= floor or c["back"]:
            print("%-12s line %6d  n %4d valu %4d (vop3 %3d dpp %3d) salu %3d lds %3d vmem %3d wait %2d %s"
                  % (b, label_line[b] + 1, c["n"], c["valu"], c["vop3"], c["dpp"], c["salu"], c["lds"], c["vmem"], c["wait"], c["back"]))


if __name__ == "__main__":
    main()
