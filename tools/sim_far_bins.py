"""Lane use of the deferred far phase of the fitness search on configs[4] under different orders of the far list
(no GPU: the CPU checker gives the final transforms of a few seeds, the walk of nearest_far_tiles is replayed per query).

    python tools/sim_far_bins.py [seeds]

Per query the replay records the sequence of buckets the generator reads (their point counts); a wave of 64 queries
costs sum_k max_lane cost(bucket k of the lane) -- lanes meet once per bucket (ndt_fitness.hip.h nearest_far_tiles) --
with cost = C0 + C1 * ceil(pairs / 6) (scan_bucket reads six 16-byte pairs per turn).  Orders compared: the list as
the kernel fills it (scan order = window-cell order), by the distance in hand, by distance inside groups of 256 / 1024
list entries, by home voxel.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from ndt_slam_amd import synth  # noqa: E402
from oracle import ndt_oracle as orc  # noqa: E402

C0, C1 = 40.0, 30.0          # instructions per bucket visit / per turn of scan_bucket's loop (from the ISA, roughly)


def grid_of(m, leaf):
    inv = np.float32(1.0 / leaf)
    ix = np.floor(m[:, 0] * inv).astype(np.int64)
    iy = np.floor(m[:, 1] * inv).astype(np.int64)
    mnx, mny = ix.min(), iy.min()
    dx, dy = ix.max() - mnx + 1, iy.max() - mny + 1
    cell = (iy - mny) * dx + (ix - mnx)
    order = np.argsort(cell, kind="stable")
    start = np.zeros(dx * dy + 1, np.int64)
    np.add.at(start, cell + 1, 1)
    start = np.cumsum(start)
    return dict(mnx=mnx, mny=mny, dx=dx, dy=dy, start=start, pts=m[order].astype(np.float32), leaf=np.float32(leaf), inv=inv)


def bucket_min(G, cell, qx, qy, best):
    s, e = G["start"][cell], G["start"][cell + 1]
    if e > s:
        p = G["pts"][s:e]
        ex = qx - p[:, 0]
        ey = qy - p[:, 1]
        d = (ex * ex + ey * ey).min()
        if d < best:
            best = d
    return best, int(e - s)


def replay(G, qx, qy, nearest_first=False):
    """-> (needs far phase, best in hand, [bucket sizes read by the far walk])"""
    L = G["leaf"]
    cx0 = int(math.floor(np.float32(qx * G["inv"]))) - G["mnx"]
    cy0 = int(math.floor(np.float32(qy * G["inv"]))) - G["mny"]
    cx = min(max(cx0, 0), G["dx"] - 1)
    cy = min(max(cy0, 0), G["dy"] - 1)
    inside = cx == cx0 and cy == cy0
    slack = max(1e-3 * L, 2.5e-7 * (abs(qx) + abs(qy) + L))
    fx = qx - (cx + G["mnx"]) * L
    fy = qy - (cy + G["mny"]) * L
    wl, wr, wd, wu = max(fx - slack, 0.0), max(L - fx - slack, 0.0), max(fy - slack, 0.0), max(L - fy - slack, 0.0)
    if not inside:
        wl = wr = wd = wu = 0.0
    best = np.float32(np.inf)
    best, _ = bucket_min(G, cy * G["dx"] + cx, qx, qy, best)
    if not (min(wl, wr, wd, wu) ** 2 < best):
        return False, best, [], 0
    for ddy in (-1, 0, 1):                                  # ring 1 (a superset of what the kernel reads: same minimum)
        for ddx in (-1, 0, 1):
            x, y = cx + ddx, cy + ddy
            if (ddx or ddy) and 0 <= x < G["dx"] and 0 <= y < G["dy"]:
                best, _ = bucket_min(G, y * G["dx"] + x, qx, qy, best)
    if float(best) <= (float(L) * 0.999) ** 2:
        return False, best, [], 0
    hand = best
    reads = []
    rows = 0
    skip = None
    if nearest_first and not np.isfinite(best):
        cand = None
        for step in range(17):
            k = (step + 1) >> 1
            up = step != 0 and not (step & 1)
            by = 0.0 if k == 0 else (wu if up else wd) + (k - 1) * L
            yy = cy + k if up else cy - k
            if yy < 0 or yy >= G["dy"]:
                continue
            tx = cx >> 3
            lo, hi = max((tx - 1) * 8, 0), min((tx + 2) * 8, G["dx"])
            row = G["start"][yy * G["dx"] + lo: yy * G["dx"] + hi + 1]
            for j in np.nonzero(np.diff(row))[0]:
                x = lo + j
                dxv = x - cx
                if k <= 1 and abs(dxv) <= 1:
                    continue
                bx = wl + (-dxv - 1) * L if dxv < 0 else (wr + (dxv - 1) * L if dxv > 0 else 0.0)
                d = bx * bx + by * by
                if cand is None or d < cand[0]:
                    cand = (d, yy, x)
        if cand is not None:
            best, cnt = bucket_min(G, cand[1] * G["dx"] + cand[2], qx, qy, best)
            reads.append(cnt)
            skip = (cand[1], cand[2])
    for step in range(17):                                  # rows cy, cy-1, cy+1, ...
        k = (step + 1) >> 1
        up = step != 0 and not (step & 1)
        by = 0.0 if k == 0 else (wu if up else wd) + (k - 1) * L
        if not (by * by < best):
            continue
        yy = cy + k if up else cy - k
        if yy < 0 or yy >= G["dy"]:
            continue
        rows += 1
        tx = cx >> 3
        lo, hi = max((tx - 1) * 8, 0), min((tx + 2) * 8, G["dx"])      # the three tiles' columns
        row = G["start"][yy * G["dx"] + lo: yy * G["dx"] + hi + 1]
        occ = [lo + j for j in np.nonzero(np.diff(row))[0]]
        if k <= 1:
            occ = [x for x in occ if abs(x - cx) > 1]
        occ.sort(key=lambda x: (abs(x - cx), x > cx))
        closed_l = closed_r = False
        for x in occ:
            if not (by * by < best):
                break
            dxv = x - cx
            if (dxv < 0 and closed_l) or (dxv > 0 and closed_r):
                continue
            bx = wl + (-dxv - 1) * L if dxv < 0 else (wr + (dxv - 1) * L if dxv > 0 else 0.0)
            if skip == (yy, x):
                continue
            if bx * bx + by * by < best:
                best, cnt = bucket_min(G, yy * G["dx"] + x, qx, qy, best)
                reads.append(cnt)
            elif dxv < 0:
                closed_l = True
            else:
                closed_r = True
    return True, hand, reads, rows


def wave_cost(lanes):
    """lanes: list of read sequences -> (wave cost, summed lane cost)"""
    depth = max((len(r) for r in lanes), default=0)
    tot = 0.0
    mine = 0.0
    for k in range(depth):
        c = [C0 + C1 * math.ceil(((r[k] + 1) // 2) / 6) for r in lanes if len(r) > k]
        tot += max(c)
        mine += sum(c)
    return tot, mine


def use(seqs, order):
    w = m = 0.0
    for i in range(0, len(order), 64):
        a, b = wave_cost([seqs[j] for j in order[i:i + 64]])
        w += a
        m += b
    return (m / (64.0 * w) if w else 0.0), w


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    c5 = synth.CONFIGS["C5"]
    m5 = synth.make_map(c5["n_map"], c5["half"])
    fac = synth.ScanFactory(m5, c5["half"], c5["n_scan"])
    scan, truth, _ = fac.make(0)
    seeds = synth.hypothesis_seeds(truth, c5["seeds"])
    pick = seeds[:: len(seeds) // nseeds][:nseeds]
    prm = orc.default_params(resolution=c5["resolution"])
    omap = orc.Map(m5, prm)
    G = grid_of(m5, c5["resolution"])
    tot = {}
    for s, init in enumerate(pick):
        r = omap.align(scan, init)
        c, sn, tx, ty = (np.float32(r[k]) for k in ("T00", "T10", "T03", "T13"))
        q = np.stack([c * scan[:, 0] - sn * scan[:, 1] + tx, sn * scan[:, 0] + c * scan[:, 1] + ty], 1).astype(np.float32)
        # the list is filled in the order of the cell-ordered scan: order the queries by their voxel at this pose (proxy)
        vox = np.floor(q * G["inv"]).astype(np.int64)
        scan_order = np.lexsort((vox[:, 0], vox[:, 1]))
        seqs, hand, home, blind, nrows, bseqs, bhome, nfseqs, nfrows = [], [], [], [], [], [], [], [], []
        for i in scan_order:
            need, h, reads, rows = replay(G, float(q[i, 0]), float(q[i, 1]))
            if need and not np.isfinite(h):
                blind.append((len(reads), rows, sum(reads)))
                bseqs.append(reads)
                nf = replay(G, float(q[i, 0]), float(q[i, 1]), True)
                nfseqs.append(nf[2])
                nfrows.append(nf[3])
                bhome.append(int(vox[i, 1]) * 100000 + int(vox[i, 0]))
            if need and np.isfinite(h):
                nrows.append(rows)
                seqs.append(reads)
                hand.append(float(h))
                home.append(int(vox[i, 1]) * 100000 + int(vox[i, 0]))
        n = len(seqs)
        hand = np.array(hand)
        home = np.array(home)
        ident = np.arange(n)
        orders = {
            "as filled": ident,
            "by distance": np.argsort(hand, kind="stable"),
            "by distance in 256": np.concatenate([i + np.argsort(hand[i:i + 256], kind="stable") for i in range(0, n, 256)]) if n else ident,
            "by distance in 1024": np.concatenate([i + np.argsort(hand[i:i + 1024], kind="stable") for i in range(0, n, 1024)]) if n else ident,
            "by reads (ideal)": np.argsort([len(r) for r in seqs], kind="stable"),
            "by cost (ideal)": np.argsort([wave_cost([r])[0] for r in seqs], kind="stable"),
        }
        for nb in (4, 8):
            edges = np.quantile(hand, np.linspace(0, 1, nb + 1)[1:-1]) if n else []
            orders["%d bins of distance" % nb] = np.argsort(np.searchsorted(edges, hand), kind="stable")
        line = "seed %d: %d far queries, reads/query %.1f |" % (s, n, np.mean([len(r) for r in seqs]) if n else 0)
        for name, o in orders.items():
            u, w = use(seqs, list(o))
            a = tot.setdefault(name, [0.0, 0.0])
            a[0] += u * w
            a[1] += w
            line += " %s %.2f (%.0fk)" % (name, u, w / 1e3)
        print(line, flush=True)
        nbq = len(bseqs)
        borders = {"blind as filled": np.arange(nbq), "blind by home voxel": np.argsort(bhome, kind="stable"),
                   "blind by reads (ideal)": np.argsort([len(r) for r in bseqs], kind="stable"),
                   "blind by cost (ideal)": np.argsort([wave_cost([r])[0] for r in bseqs], kind="stable")}
        for name, o in borders.items():
            u, w = use(bseqs, list(o))
            a = tot.setdefault(name, [0.0, 0.0])
            a[0] += u * w
            a[1] += w
        if nbq:
            u, w = use(nfseqs, list(range(nbq)))
            a = tot.setdefault("blind, nearest box first", [0.0, 0.0])
            a[0] += u * w
            a[1] += w
            print("   blind, nearest box first: reads %.1f rows %.1f points %.0f" % (np.mean([len(r) for r in nfseqs]), np.mean(nfrows), np.mean([sum(r) for r in nfseqs])))
        hist = np.bincount([len(r) for r in seqs], minlength=6)
        print("   reads histogram", hist[:12], "rows opened/query %.1f" % np.mean(nrows), "| blind", len(blind),
              "reads %.1f rows %.1f points %.0f" % tuple(np.mean(blind, 0)) if blind else "", flush=True)
    print("\nlane use (cost-weighted) and wave cost relative to 'as filled':")
    base = tot["as filled"][1]
    for name, (uw, w) in tot.items():
        print("  %-22s use %.3f  cost %.3f" % (name, uw / w, w / base))


if __name__ == "__main__":
    main()
