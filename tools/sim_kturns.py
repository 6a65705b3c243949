"""Offline (no GPU): pair-loop turns of the match kernel under different arrangements of a scan's points.
For bench scans, K(point) = in-radius voxel centroids at the INIT pose (what set-up knows) and at the TRUTH pose (what the
later passes see).  A wave's pair loop runs max-over-lanes(K) turns per round of 64 points.
  current : sorted by window cell, unit (w, q) = points w*64 + lane + k*1024
  contig  : unit u = contiguous range of the sorted scan, dealt to (round, lane) by K at the init pose inside the unit
  wave640 : a wave's 10 rounds contiguous (640 points) and dealt by K
Usage: python tools/sim_kturns.py [n_scans]"""
import math, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndt_slam_amd import synth
from oracle import ndt_oracle as O

def main():
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    cfg = synth.CONFIGS["C3"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    prm = O.default_params(resolution=cfg["resolution"])
    M = O.Map(m, prm); info = M.info(); t = M.export()
    res = np.float32(cfg["resolution"]); inv = np.float32(1.0) / res
    gw, gh = info.div_x + 4, info.div_y + 4
    cent = np.full((gh, gw, 2), np.inf, np.float32)
    iy, ix = np.divmod(t["idx"], info.div_x)
    cent[iy + 2, ix + 2] = t["cent"]
    r2 = np.float32(float(res) * float(res))
    def K_of(scan, pose):
        c, s = np.float32(math.cos(np.float32(pose[2]))), np.float32(math.sin(np.float32(pose[2])))
        x = c * scan[:, 0] + (-s * scan[:, 1] + np.float32(pose[0])); y = s * scan[:, 0] + (c * scan[:, 1] + np.float32(pose[1]))
        vx = np.floor(x * inv).astype(np.int64) - info.min_bx; vy = np.floor(y * inv).astype(np.int64) - info.min_by
        K = np.zeros(len(scan), np.int64)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                cc = cent[np.clip(vy + dy + 2, 0, gh - 1), np.clip(vx + dx + 2, 0, gw - 1)]
                ex = x - cc[:, 0]; ey = y - cc[:, 1]
                K += ((ex * ex + ey * ey) < r2)
        return K, vy * 100000 + vx
    tot = {}
    def add(name, turns, useful): 
        a = tot.setdefault(name, [0, 0]); a[0] += turns; a[1] += useful
    for b in range(ns):
        scan, truth, init = sf.make(b)
        K0, cell0 = K_of(scan, init)
        order = np.argsort(cell0, kind="stable")
        K1, _ = K_of(scan, truth)
        n = len(scan)
        for tag, Kset, Kuse in (("init", K0, K0), ("truth", K0, K1)):
            Ks, Ku = Kset[order], Kuse[order]
            pad = (-n) % 1024
            Kp = np.concatenate([Ku, np.zeros(pad, np.int64)])
            # current: round (k, w) = 64 consecutive sorted points at k*1024 + w*64
            rounds = Kp.reshape(-1, 64)
            add("current/" + tag, rounds.max(1).sum(), Ku.sum())
            # contiguous units of ceil(n/64) points, dealt by K (stable) inside the unit
            for label, unit in (("contig160", 160), ("contig320", 320), ("wave640", 640), ("all", n)):
                turns = 0
                for a in range(0, n, unit):
                    seg = slice(a, min(a + unit, n))
                    o = np.argsort(Ks[seg], kind="stable")
                    ku = Ku[seg][o]
                    padu = (-len(ku)) % 64
                    turns += np.concatenate([ku, np.zeros(padu, np.int64)]).reshape(-1, 64).max(1).sum()
                add(label + "/" + tag, turns, Ku.sum())
    for k in sorted(tot):
        turns, useful = tot[k]
        print("%-20s turns per round %.3f (mean K %.3f, lane use %.2f)" % (k, turns / (ns * 157.0), useful / (ns * 10000.0), useful / (turns * 64.0)))

main()
