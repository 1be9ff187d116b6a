#!/usr/bin/env python3
"""docs/K1_MODEL.md, section 2: T_K1 = max(T_traffic, T_arith) * (1 + eps(r)) against the measured occupancy sweeps
(profiles/r05_k1_occupancy_and_floors.txt, box B: product, abl1s = K1's traffic alone, abl6 = K1's arithmetic alone).
Prints predicted vs measured for every occupancy and both formats; exit status 1 if any point is off by more than 5 %."""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def eps(r):
    return 0.015 + 0.085 * min(max((r - 0.62) / 0.25, 0.0), 1.0)


def model(t_traffic, t_arith):
    hi, lo = max(t_traffic, t_arith), min(t_traffic, t_arith)
    return hi * (1.0 + eps(lo / hi))


def load(path=os.path.join(ROOT, "profiles", "r05_k1_occupancy_and_floors.txt"), box="== box B =="):
    txt = open(path).read()
    blk = txt[txt.index(box):]
    blk = blk[:blk.index("\n==", 5)] if "\n==" in blk[5:] else blk
    d = {}
    for m in re.finditer(r"^(\w+)\s+pad\s+\d+\s+waves/CU\s+(\d+)\s+(\w+)\s+K1 ([0-9.]+) ms", blk, re.M):
        d[(m.group(1), int(m.group(2)), m.group(3))] = float(m.group(4))
    return d


def main():
    d = load()
    worst = 0.0
    for fmt in ("cf32", "u8"):
        print("%-5s %3s %9s %9s %9s %7s %9s %7s" % (fmt, "k", "traffic", "arith", "model", "r", "measured", "err %"))
        for k in sorted({key[1] for key in d}):
            tt, ta, cur = d[("abl1s", k, fmt)], d[("abl6", k, fmt)], d[("cur", k, fmt)]
            p = model(tt, ta)
            err = (p / cur - 1.0) * 100.0
            worst = max(worst, abs(err))
            print("%-5s %3d %9.4f %9.4f %9.4f %7.3f %9.4f %+7.1f" % (fmt, k, tt, ta, p, min(tt, ta) / max(tt, ta), cur, err))
    print("worst |error| %.1f %%" % worst)
    return 1 if worst > 5.0 else 0


if __name__ == "__main__":
    sys.exit(main())
