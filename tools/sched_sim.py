#!/usr/bin/env python3
"""List-scheduling model of a causal forward launch: workgroups reach the 8 XCDs round-robin (block id % 8) and the 32 compute units of an XCD
take them in id order as they free up.  Compares the two orders of fa_device.hpp:decode_block - heaviest query tile first within each
(batch, head) (round 3), tile index first across all heads, and the shipped order: groups of ceil(64 / tiles) heads, tile index first inside a group.  Cost of a workgroup = fixed + per-tile
time x (4 t + 4) for its tile index t (256 query rows, 64-key tiles); the constants are the round-2 fit (profiles/NOTEBOOK.md 4).  A model, not a
measurement: profiles/r4_causal_tile_order_ab.log is the measurement."""
import heapq


def sim(seq, order, b=4, h=32, fixed=3.3, per=1.85, launch=4.1):
    tiles, n_bh = seq // 256, b * h
    wgs = []
    for i in range(tiles * n_bh):
        xcd, slot = i & 7, i >> 3
        per_xcd = n_bh // 8
        if order == "head":
            t = tiles - 1 - slot % tiles
        elif order == "tile":
            t = tiles - 1 - slot // per_xcd
        else:                                              # "group": ceil(64 / tiles) heads at a time, tile index first inside the group
            g = min(per_xcd, max(1, -(-64 // tiles)))
            span = g * tiles
            grp, within = divmod(slot, span)
            here = min(g, per_xcd - grp * g)
            t = tiles - 1 - within // here
        wgs.append((xcd, fixed + per * (4 * t + 4)))
    end = 0.0
    for x in range(8):
        cus = [0.0] * 32
        heapq.heapify(cus)
        for xc, cost in wgs:
            if xc == x:
                heapq.heappush(cus, heapq.heappop(cus) + cost)
        end = max(end, max(cus))
    return end + launch


if __name__ == "__main__":
    for seq in (512, 1024, 2048, 4096, 8192, 16384):
        fl = 4 * 4 * 32 * seq * seq * 128 * 0.5
        a, c, g = sim(seq, "head"), sim(seq, "tile"), sim(seq, "group")
        print(f"seq {seq:6d}  head-major {a:8.1f} us {fl / a / 1e6:6.0f} TF | tile-major {c:8.1f} us {fl / c / 1e6:6.0f} TF ({c / a:.3f}) | grouped {g:8.1f} us {fl / g / 1e6:6.0f} TF ({g / a:.3f})")
