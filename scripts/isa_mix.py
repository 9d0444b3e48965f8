#!/usr/bin/env python3
"""Instruction mix of the traversal kernel's node step from the compiler's assembly, by issue class of a gfx950 SIMD
(scripts/microbench/valu_issue.hip: which wave64 vector instructions can issue on either of the two pipes at 2.3 cycles, which on
one pipe only at 4.2).
   hipcc <the Makefile's flags> -S --cuda-device-only -o wf.s ptr_kernels_wavefront.hip
   scripts/isa_mix.py wf.s [kernel name substring]      (default: the untimed-statistics-free single-level variant)
Prints the inner loop (the block between the node's four 16-byte loads and the loop's back edge) with every vector instruction
tagged, and the totals."""
import re, sys

BOTH = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32")


def classify(op, operands):
    if not op.startswith("v_"):
        return "scalar/memory/other"
    base = re.sub(r"_e(32|64)$", "", op)
    if base.endswith("_dpp") or base.endswith("_sdwa") or base.startswith("v_pk_"):
        return "one pipe"
    if base in BOTH and not re.search(r"\bs\[?\d|\bvcc\b|\bexec\b|\bs\d", operands.split(",", 1)[1] if "," in operands else ""):
        return "either pipe"
    return "one pipe"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "ptr_wf_extendILb0ELi16ELi2ELb1ELi3ELb0E"
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and want in l and l.rstrip().endswith(":") or (l.startswith("_ZN") and want in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # the node step: from the first group of four consecutive 16-byte loads to the next unconditional loop control before the leaf code
    first = next(i for i in range(len(body) - 3) if all("global_load_dwordx4" in body[i + k] for k in range(4)))
    # walk back to the loop header label
    head = max(i for i in range(first) if body[i].startswith(".LBB"))
    tail = next(i for i in range(first, len(body)) if "s_cbranch_scc" in body[i] or re.match(r"\.LBB\d+_\d+:.*Depth=1$", body[i].strip()) and i > first + 50)
    counts = {}
    ops = {}
    for l in body[head:tail]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            print(l)
            continue
        parts = s.split(None, 1)
        op, rest = parts[0], parts[1] if len(parts) > 1 else ""
        rest = rest.split(";")[0]
        c = classify(op, rest)
        counts[c] = counts.get(c, 0) + 1
        ops.setdefault(c, {}).setdefault(re.sub(r"_e(32|64)$", "", op), 0)
        ops[c][re.sub(r"_e(32|64)$", "", op)] += 1
        print(f"{l:<72s} ; [{c}]" if op.startswith("v_") else l)
    print("\n; ---- totals of the block above (every path through it, spill and pop branches included) ----")
    for c in sorted(counts):
        detail = ", ".join(f"{k} {v}" for k, v in sorted(ops[c].items(), key=lambda kv: -kv[1]))
        print(f"; {c}: {counts[c]}  ({detail})")


if __name__ == "__main__":
    main()
