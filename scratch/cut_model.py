#!/usr/bin/env python3
"""A model of how long one k_bounce launch takes for a given cut of the batch (tinsel_hip_plan_regions): 256 CUs x 3 resident workgroups,
workgroups dispatched in index order to the CU with the fewest resident ones, a CU's throughput shared by its resident workgroups
(1 / 2 / 3 resident: 0.42 / 0.78 / 1.0 of the CU -- the measured gain of the second and third wave per SIMD), a workgroup's work
proportional to its positions.  Prints the modelled efficiency (work / (CUs x makespan)) of the cut the library makes now.
Usage: cut_model.py [slots ...]"""
import heapq
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinsel_amd  # noqa: E402

THR = {0: 0.0, 1: 0.42, 2: 0.78, 3: 1.0}


def simulate(groups, cus=256, slots_per_cu=3):
    """groups: list of work amounts in dispatch order.  Event-driven processor sharing per CU."""
    res = [[] for _ in range(cus)]          # remaining work of the resident groups of each CU
    nxt = 0
    t = 0.0
    # initial fill: round-robin over CUs, one at a time
    for s in range(slots_per_cu):
        for c in range(cus):
            if nxt < len(groups):
                res[c].append(groups[nxt]); nxt += 1
    while True:
        # time to the next completion on every CU
        best = None
        for c in range(cus):
            k = len(res[c])
            if k:
                rate = THR[k]/k
                dt = min(res[c])/rate
                if best is None or dt < best:
                    best = dt
        if best is None:
            break
        t += best
        for c in range(cus):
            k = len(res[c])
            if k:
                rate = THR[k]/k
                res[c] = [w - best*rate for w in res[c]]
                done = [w for w in res[c] if w <= 1e-9]
                res[c] = [w for w in res[c] if w > 1e-9]
                for _ in done:
                    if nxt < len(groups):
                        res[c].append(groups[nxt]); nxt += 1
    return t


def groups_of(plan, slots):
    n, L, big, S = plan["num_regions"], plan["region_len"], plan["big_regions"], plan["short_len"]
    out = []
    pos = 0
    for g in range(n//4):
        ln = L if g*4 < big else S
        w = 0
        for k in range(4):
            w += max(0, min(ln, slots - pos)); pos += ln
        if w:
            out.append(float(w))
    return out


def efficiency(slots, plan, cus=256):
    g = groups_of(plan, slots)
    return sum(g)/(cus*simulate(g, cus)), len(g)


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [256*256*16, 512*512*4, 512*512*6, 512*512*8, 512*512*16, 1024*1024*2, 1024*1024*4, 1024*1024*8, 1024*1024*20,
                                                1920*1080, 1920*1080*4, 1920*1080*16, 1024*1024*64, 3840*2160*8]
    for s in sizes:
        p = tinsel_amd.plan_regions(s, 256, 1, True)
        e, ng = efficiency(s, p)
        u = dict(p); u["big_regions"] = (s + p["region_len"] - 1)//p["region_len"]
        print("%10d slots: %5d busy groups (L %d x %d, S %d x %d): modelled efficiency %.3f" % (s, ng, p["region_len"], p["big_regions"], p["short_len"], p["num_regions"] - p["big_regions"], e))


def make_plan(slots, L, big_groups, S):
    big = big_groups*4
    covered = big*L
    rest = max(0, slots - covered)
    small = (rest + S*4 - 1)//(S*4)*4
    return {"num_regions": big + small, "region_len": L, "big_regions": big, "short_len": S}


def search(slots, cus=256):
    """the best (long groups per CU, long share, short divide) of a small family, in the model"""
    best = None
    for per_cu in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
        for share in (1.0, 0.9, 0.85, 0.8, 0.75, 0.7, 0.6, 0.5):
            for div in (2, 3, 4, 6):
                if per_cu == 0:
                    continue
                L = int(slots*share/(per_cu*cus*4))//64*64
                if L < 64*div:
                    continue
                S = L//div//64*64
                p = make_plan(slots, L, per_cu*cus, S)
                if p["num_regions"] > 49152:
                    continue
                e, ng = efficiency(slots, p, cus)
                if best is None or e > best[0]:
                    best = (e, per_cu, share, div, L, S, ng)
    return best
