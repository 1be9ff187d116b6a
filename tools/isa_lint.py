#!/usr/bin/env python3
"""ISA lint for the hand-scheduled LDS reads of K1.

lds_issue_b64 / lds_landed* (p25fe_kernels.hip) define VGPRs from `asm volatile("ds_read_b64 ...")` that the compiler
believes are ready while the LDS read is still in flight; correctness rests on the explicit `s_waitcnt lgkmcnt(n)` asm in
front of every use AND on the compiler never touching such a register in between (a copy, a coalescing move, a spill).
This script walks the generated gfx950 assembly of every k_frontend / k_chunk kernel and checks exactly that:

  * every inline-asm `ds_read_b64 v[a:b]` puts (a, b) on an in-flight list (LDS operations of a wave return in order);
  * an inline-asm `s_waitcnt lgkmcnt(n)` lands all but the newest n; a compiler-emitted `s_waitcnt` with lgkmcnt(n) does too;
  * any other instruction that names an in-flight register (as source OR destination) is a violation;
  * no `scratch_` instruction may appear inside the sub-tile loop (loop depth 2) of the default-tap kernels (the
    caller-supplied-tap variants, which read their coefficients from LDS, are reported as warnings).

usage: isa_lint.py <file.s>   (make -C p25rx_amd/csrc asm writes /tmp/p25fe_api-hip-amdgcn-amd-amdhsa-gfx950.s)
Exit code 0 = clean."""
import re
import sys


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return out


def lint(path):
    src = open(path).read()
    bad, warn, n_kernels, n_reads = [], [], 0, 0
    for m in re.finditer(r"^(_ZN4p25k(?:10k_frontend|7k_chunk)\w+):[^\n]*\n", src, re.M):
        name = m.group(1)
        end = src.index(".Lfunc_end", m.end())
        lines = src[m.end():end].split("\n")
        n_kernels += 1
        inflight = []                    # list of register sets, oldest first
        in_asm, depth2 = False, False
        for ln, l in enumerate(lines):
            s = l.strip()
            if s.startswith(".LBB") or s.startswith(";"):
                if "Depth=2" in l:
                    depth2 = True
                elif s.startswith(".LBB") and "Depth" not in l and "Loop" not in l:
                    pass
                if "Depth=1" in l and "Inner" not in l and "Parent" not in l and "in Loop" not in l:
                    depth2 = False
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            op = s.split()[0]
            if in_asm and op == "ds_read_b64":
                dst = regs_of(s.split(",")[0])
                inflight.append(dst)
                n_reads += 1
                continue
            if op == "s_waitcnt":
                k = re.search(r"lgkmcnt\((\d+)\)", s)
                if k:
                    n = int(k.group(1))
                    inflight = inflight[len(inflight) - n:] if n else []
                elif "lgkmcnt" not in s and "vmcnt" in s:
                    pass
                continue
            if op.startswith("scratch_") and depth2:
                msg = "%s: line %d: scratch access inside the sub-tile loop: %s" % (name, ln, s)
                default_taps = "Lb1E" in name                        # template argument CT = true
                (bad if default_taps else warn).append(msg)
            if op.startswith("ds_") and not in_asm:
                # a compiler-scheduled LDS op takes a slot in the same in-order queue: track its results as landed only by a wait
                continue
            used = regs_of(s)
            for grp in inflight:
                hit = used & grp
                if hit:
                    bad.append("%s: line %d: touches v%s while its LDS read is in flight: %s" % (name, ln, sorted(hit), s))
    return bad, warn, n_kernels, n_reads


if __name__ == "__main__":
    bad, warn, nk, nr = lint(sys.argv[1])
    for b in bad[:40]:
        print(b)
    for w in warn[:10]:
        print("warning:", w)
    print("isa_lint: %d kernels, %d hand-issued LDS reads, %d violations, %d warnings" % (nk, nr, len(bad), len(warn)))
    sys.exit(1 if bad or nk == 0 or nr == 0 else 0)
