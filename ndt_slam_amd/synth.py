"""Synthetic worlds, scans and initial guesses (SURVEY.md 8d "Synthetic inputs").

The reference ships no data (its input log is a path on the author's machine,
ndt_mapping.launch:3), so every workload is generated: a 2-D world of noisy wall segments,
scans = map subsets seen from a known pose, initial guess = truth + odometry-sized error.
PRNG: numpy Philox (counter based), map_seed = 20211107, scan_seed = 1000 + i.
"""
import math

import numpy as np

MAP_SEED = 20211107
SCAN_SEED0 = 1000

# BASELINE.json configs -> (scan points, map points, world half extent [m], resolution [m])
CONFIGS = {
    "C1": dict(n_scan=360, n_map=5_000, half=24.0, resolution=0.3),
    "C2": dict(n_scan=10_000, n_map=1_000_000, half=128.0, resolution=0.5),
    "C3": dict(n_scan=10_000, n_map=1_000_000, half=128.0, resolution=0.5, batch=256),
    "C4": dict(n_scan=10_000, n_map=1_000_000, half=128.0, resolution=0.5, batch=2048),
    "C5": dict(n_scan=10_000, n_map=5_000_000, half=256.0, resolution=0.5, seeds=4096),
}


def _rng(seed):
    return np.random.Generator(np.random.Philox(seed))


def make_map(n_points, half_extent, seed=MAP_SEED, spacing=0.02, sigma=0.02):
    """Random wall segments (50 % axis aligned, 50 % any orientation), length U(5,50) m clipped
    to the world, sampled every `spacing` m with isotropic N(0, sigma^2) noise, appended until
    exactly n_points.  Returns float32 [n_points, 2]."""
    rng = _rng(seed)
    out = np.empty((n_points, 2), dtype=np.float64)
    filled = 0
    max_len = min(50.0, 1.5 * half_extent)
    min_len = min(5.0, 0.2 * half_extent)
    while filled < n_points:
        start = rng.uniform(-half_extent, half_extent, size=2)
        if rng.random() < 0.5:
            ang = rng.integers(0, 4) * (math.pi / 2)
        else:
            ang = rng.uniform(-math.pi, math.pi)
        length = rng.uniform(min_len, max_len)
        k = int(length / spacing) + 1
        t = np.arange(k, dtype=np.float64) * spacing
        pts = start[None, :] + t[:, None] * np.array([math.cos(ang), math.sin(ang)])[None, :]
        pts += rng.normal(0.0, sigma, size=pts.shape)
        keep = (np.abs(pts[:, 0]) <= half_extent) & (np.abs(pts[:, 1]) <= half_extent)
        pts = pts[keep]
        take = min(len(pts), n_points - filled)
        out[filled:filled + take] = pts[:take]
        filled += take
    return out.astype(np.float32)


class ScanFactory:
    """Scans = map subsets moved into a sensor frame at a known pose."""

    def __init__(self, map_xy, half_extent, n_scan, radius=30.0, sigma=0.01):
        from scipy.spatial import cKDTree
        self.map64 = np.asarray(map_xy, dtype=np.float64)
        self.tree = cKDTree(self.map64)
        self.half = half_extent
        self.n_scan = n_scan
        self.radius = min(radius, half_extent)
        self.sigma = sigma

    def truth_pose(self, index, seed0=SCAN_SEED0):
        rng = _rng(seed0 + index)
        xy = rng.uniform(-0.8 * self.half, 0.8 * self.half, size=2)
        yaw_deg = rng.uniform(-180.0, 180.0)
        if index % 8 == 5:   # stratum that exercises the asin/acos extraction (a9)
            yaw_deg = [90.0, -90.0, 180.0, -180.0][(index // 8) % 4] + rng.uniform(-0.5, 0.5)
            if yaw_deg > 180.0:
                yaw_deg -= 360.0
            if yaw_deg < -180.0:
                yaw_deg += 360.0
        return rng, np.array([xy[0], xy[1], math.radians(yaw_deg)])

    def make(self, index, seed0=SCAN_SEED0):
        """-> (scan float32 [n,2] in the sensor frame, truth (x,y,yaw rad), init (x,y,yaw rad))."""
        rng, truth = self.truth_pose(index, seed0)
        r = self.radius
        while True:
            idx = self.tree.query_ball_point(truth[:2], r)
            if len(idx) >= self.n_scan or r > 4 * self.half:
                break
            r *= 1.5
        idx = np.asarray(idx, dtype=np.int64)
        idx.sort()
        if len(idx) >= self.n_scan:
            sel = rng.choice(idx, size=self.n_scan, replace=False)
        else:  # tiny worlds only
            sel = rng.choice(idx, size=self.n_scan, replace=True)
        d = self.map64[sel] - truth[None, :2]
        c, s = math.cos(truth[2]), math.sin(truth[2])
        loc = np.stack([c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1]], axis=1)
        loc += rng.normal(0.0, self.sigma, size=loc.shape)
        init = truth + np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2),
                                 math.radians(rng.uniform(-3.0, 3.0))])
        return loc.astype(np.float32), truth, init

    def batch(self, first, count, seed0=SCAN_SEED0):
        """Concatenated batch: (scans [count*n,2] f32, offsets u64 [count+1], truths, inits)."""
        scans, truths, inits = [], [], []
        for i in range(first, first + count):
            sc, t, g = self.make(i, seed0)
            scans.append(sc); truths.append(t); inits.append(g)
        offsets = np.zeros(count + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(s) for s in scans])
        return (np.concatenate(scans, axis=0), offsets, np.stack(truths), np.stack(inits))


def hypothesis_seeds(truth, count=4096, seed=5, pitch=0.1, yaw_deg=10.0):
    """C5: seed poses on a sqrt(count)^2 lattice around truth, yaw = truth + U(-10,10) deg."""
    side = int(round(math.sqrt(count)))
    assert side * side == count
    rng = _rng(seed)
    ax = (np.arange(side) - (side - 1) / 2.0) * pitch
    gx, gy = np.meshgrid(ax, ax, indexing="xy")
    out = np.empty((count, 3))
    out[:, 0] = truth[0] + gx.ravel()
    out[:, 1] = truth[1] + gy.ravel()
    out[:, 2] = truth[2] + np.radians(rng.uniform(-yaw_deg, yaw_deg, size=count))
    return out


def submap_scans(n_scans, n_points, seed=21, mover_frac=0.025, jitter=0.003, room=(8.0, 6.0)):
    """Scans of one submap already registered in the map frame (input of Submap::makeMap):
    the walls of a room re-observed by every scan at the same bearings with `jitter` metres of noise, plus a
    small object that moves 0.6 m between scans (what the moving-object removal is there to drop)."""
    rng = _rng(seed)
    n_mover = max(1, int(n_points * mover_frac))
    n_wall = n_points - n_mover
    th = np.linspace(0.0, 2.0 * np.pi, n_wall, endpoint=False)
    d = np.maximum(np.abs(np.cos(th)), np.abs(np.sin(th)))
    walls = np.stack([room[0] * np.cos(th) / d, room[1] * np.sin(th) / d], axis=1)
    out = []
    for k in range(n_scans):
        mover = np.stack([rng.normal(-3.0 + 0.6 * k, 0.1, n_mover), rng.normal(0.5, 0.15, n_mover)], axis=1)
        out.append((np.concatenate([walls, mover]) + rng.normal(size=(n_points, 2)) * jitter).astype(np.float32))
    return out



def replay_records(n_frames=40, n_beams=541, fov_deg=270.0, step=0.3, seed=33, max_range=30.0, sigma=0.01,
                   odo_scale=1.015, odo_yaw_bias_deg=0.05):
    """Records of a synthetic drive for the replay harness (the reference's own log is not
    in the repository): a robot circles inside a 24 m x 16 m hall with two pillars while a cart crosses the
    hall; a 2-D lidar (n_beams over fov_deg, range noise sigma) is ray-cast against the walls; the odometry
    drifts in scale and heading.  -> (records for replay.write_log, true poses [n,3] with yaw in degrees)."""
    rng = _rng(seed)

    def box(cx, cy, w, h):
        x0, x1, y0, y1 = cx - w / 2, cx + w / 2, cy - h / 2, cy + h / 2
        return [((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1)), ((x0, y1), (x0, y0))]

    static = box(0.0, 0.0, 24.0, 16.0) + box(-4.0, 2.5, 1.0, 1.0) + box(5.0, -3.0, 1.5, 0.8)
    beams = np.radians(np.linspace(-fov_deg / 2, fov_deg / 2, n_beams))
    records, truth = [], []
    odo = np.zeros(3)
    prev = None
    for k in range(n_frames):
        a = 2 * np.pi * k * step / (2 * np.pi * 5.0)                 # circle of radius 5 m
        pose = np.array([5.0 * np.cos(a) - 5.0, 5.0 * np.sin(a), np.degrees(a) + 90.0])
        pose[2] = (pose[2] + 180.0) % 360.0 - 180.0
        segs = static + box(-9.0 + 0.45 * k, -5.5, 0.8, 0.5)          # the cart
        p0 = np.array([s[0] for s in segs]); p1 = np.array([s[1] for s in segs])
        th = np.radians(pose[2]) + beams
        d = np.stack([np.cos(th), np.sin(th)], 1)                     # [b,2]
        e = p1 - p0                                                   # [m,2]
        w = p0[None, :, :] - pose[None, None, :2]                     # [1,m,2]
        den = d[:, None, 0] * e[None, :, 1] - d[:, None, 1] * e[None, :, 0]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (w[:, :, 0] * e[None, :, 1] - w[:, :, 1] * e[None, :, 0]) / den
            u = (w[:, :, 0] * d[:, None, 1] - w[:, :, 1] * d[:, None, 0]) / den
        t = np.where((den != 0) & (t > 0.05) & (u >= 0) & (u <= 1), t, np.inf)
        r = t.min(1)
        ok = r < max_range
        r = r[ok] + rng.normal(0, sigma, ok.sum())
        pts = np.stack([r * np.cos(beams[ok]), r * np.sin(beams[ok])], 1)     # sensor frame
        if prev is not None:                                          # odometry: true motion with drift
            dth = np.radians(prev[2])
            dx, dy = pose[0] - prev[0], pose[1] - prev[1]
            fx = (np.cos(dth) * dx + np.sin(dth) * dy) * odo_scale
            fy = (-np.sin(dth) * dx + np.cos(dth) * dy) * odo_scale
            oth = np.radians(odo[2])
            odo[0] += np.cos(oth) * fx - np.sin(oth) * fy
            odo[1] += np.sin(oth) * fx + np.cos(oth) * fy
            odo[2] += ((pose[2] - prev[2] + 180.0) % 360.0 - 180.0) + odo_yaw_bias_deg
            odo[2] = (odo[2] + 180.0) % 360.0 - 180.0
        else:
            odo = pose.copy()
        prev = pose
        records.append(dict(stamp=k, x=odo[0], y=odo[1], th=odo[2], image="img%04d.png" % k, front=pts,
                            left=np.zeros((0, 2)), right=np.zeros((0, 2))))
        truth.append(pose)
    return records, np.array(truth)
