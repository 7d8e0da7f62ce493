"""Deterministic synthetic inputs for the ccicp2d / local_mapper hot path
(SURVEY.md section 8(d)): a 40 x 30 m room with four 2 x 2 m pillars, a model
cloud sampled along the walls, and 1081-beam scans taken from a loop
trajectory.  numpy only; no device code, no oracle.

Classes follow the reference's two-class ICP (ccicp2d/include/ccicp2d/icp.h:85-88):
pillar returns are "ground adjacent" (GA), wall returns are NGA, so both
class-constrained searches are exercised.
"""
import numpy as np

ROOM_W, ROOM_H = 40.0, 30.0
PILLARS = [(-8.0, -6.0), (7.0, -4.0), (-5.0, 8.0), (10.0, 6.0)]
PILLAR_SIZE = 2.0
N_BEAMS = 1081
BEAM_START_DEG, BEAM_STEP_DEG = -135.0, 0.25
MAX_BEAM_RANGE = 30.0
NOISE_SIGMA = 0.01


def world_segments():
    """Returns (seg[S,4] = x0,y0,x1,y1, is_ga[S]) -- walls first, then pillars."""
    hw, hh = ROOM_W / 2, ROOM_H / 2
    segs = [(-hw, -hh, hw, -hh), (hw, -hh, hw, hh), (hw, hh, -hw, hh), (-hw, hh, -hw, -hh)]
    ga = [False] * 4
    h = PILLAR_SIZE / 2
    for (cx, cy) in PILLARS:
        segs += [(cx - h, cy - h, cx + h, cy - h), (cx + h, cy - h, cx + h, cy + h),
                 (cx + h, cy + h, cx - h, cy + h), (cx - h, cy + h, cx - h, cy - h)]
        ga += [True] * 4
    return np.array(segs, dtype=np.float64), np.array(ga, dtype=bool)


def make_map(n_points=10000, seed=12345, all_nga=False):
    """Model cloud: points uniform along the segments (prob. ~ length) plus
    N(0, 0.01^2) per coordinate.  Returns (m_ga[nGA,2], m_nga[nNGA,2]) f64."""
    segs, is_ga = world_segments()
    rs = np.random.RandomState(seed)
    d = segs[:, 2:] - segs[:, :2]
    length = np.hypot(d[:, 0], d[:, 1])
    cum = np.concatenate([[0.0], np.cumsum(length)])
    u = rs.uniform(0.0, cum[-1], size=n_points)
    si = np.clip(np.searchsorted(cum, u, side="right") - 1, 0, len(segs) - 1)
    frac = (u - cum[si]) / length[si]
    pts = segs[si, :2] + d[si] * frac[:, None] + rs.normal(0.0, NOISE_SIGMA, size=(n_points, 2))
    cls = is_ga[si] & (not all_nga)
    return np.ascontiguousarray(pts[cls]), np.ascontiguousarray(pts[~cls])


def true_pose(k, n_loop):
    a = 2.0 * np.pi * k / float(n_loop)
    return 6.0 * np.cos(a), 4.0 * np.sin(a), a + np.pi / 2


def pose_to_Rt(x, y, th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s], [s, c]], dtype=np.float64), np.array([x, y], dtype=np.float64)


def make_scan(k, n_loop=256, seed_base=1000, n_beams=N_BEAMS, all_nga=False):
    """One scan from the k-th pose of the loop.  Returns (t_ga, t_nga, pose) with
    the points in the SENSOR frame as f64 xy; beams longer than 30 m are dropped."""
    segs, is_ga = world_segments()
    x, y, th = true_pose(k, n_loop)
    rs = np.random.RandomState(seed_base + k)
    ang = th + np.deg2rad(BEAM_START_DEG + BEAM_STEP_DEG * np.arange(n_beams))
    dx, dy = np.cos(ang), np.sin(ang)
    # ray (o + r*dir) against every segment (p + s*e), s in [0,1], r > 0
    px, py = segs[:, 0][None, :], segs[:, 1][None, :]
    ex, ey = (segs[:, 2] - segs[:, 0])[None, :], (segs[:, 3] - segs[:, 1])[None, :]
    den = dx[:, None] * ey - dy[:, None] * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        r = ((px - x) * ey - (py - y) * ex) / den
        s = ((px - x) * dy[:, None] - (py - y) * dx[:, None]) / den
    ok = (np.abs(den) > 1e-12) & (r > 1e-9) & (s >= 0.0) & (s <= 1.0)
    r = np.where(ok, r, np.inf)
    hit = np.argmin(r, axis=1)
    rng = r[np.arange(n_beams), hit] + rs.normal(0.0, NOISE_SIGMA, size=n_beams)
    keep = np.isfinite(rng) & (rng <= MAX_BEAM_RANGE) & (rng > 0.0)
    a_s = np.deg2rad(BEAM_START_DEG + BEAM_STEP_DEG * np.arange(n_beams))
    pts = np.stack([rng * np.cos(a_s), rng * np.sin(a_s)], axis=1)[keep]
    cls = is_ga[hit][keep] & (not all_nga)
    return np.ascontiguousarray(pts[cls]), np.ascontiguousarray(pts[~cls]), (x, y, th)


def init_pose(k, pose, seed_base=5000, dxy=0.3, dth=0.05):
    rs = np.random.RandomState(seed_base + k)
    e = rs.uniform(-1.0, 1.0, size=3)
    return pose[0] + dxy * e[0], pose[1] + dxy * e[1], pose[2] + dth * e[2]


class ScanBatch:
    """Scans concatenated the way the C-ABI batch entry points take them:
    pts[P,2] f64 (scan s = pts[scan_off[s]:scan_off[s+1]], its first
    scan_nga[s] points are class GA), R[S,4], t[S,2] initial poses."""

    def __init__(self, pts, scan_off, scan_nga, R, t, true_poses):
        self.pts, self.scan_off, self.scan_nga = pts, scan_off, scan_nga
        self.R, self.t, self.true_poses = R, t, true_poses

    @property
    def n_scans(self):
        return len(self.scan_nga)

    @property
    def n_points(self):
        return int(self.scan_off[-1])

    def scan(self, s):
        o, e, g = self.scan_off[s], self.scan_off[s + 1], self.scan_nga[s]
        return self.pts[o:o + g], self.pts[o + g:e]

    def shard(self, rank, world):
        """Contiguous split by scan index; returns a ScanBatch."""
        n = self.n_scans
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        o0, o1 = self.scan_off[lo], self.scan_off[hi]
        return ScanBatch(self.pts[o0:o1].copy(), (self.scan_off[lo:hi + 1] - o0).astype(np.int32),
                         self.scan_nga[lo:hi].copy(), self.R[lo:hi].copy(), self.t[lo:hi].copy(),
                         self.true_poses[lo:hi].copy())


def make_batch(n_scans, n_loop=None, first=0, all_nga=False, n_beams=N_BEAMS):
    n_loop = n_loop or max(n_scans, 1)
    pts, off, nga, Rs, ts, poses = [], [0], [], [], [], []
    for i in range(n_scans):
        k = first + i
        ga, ng, pose = make_scan(k, n_loop, all_nga=all_nga, n_beams=n_beams)
        pts += [ga, ng]
        off.append(off[-1] + len(ga) + len(ng))
        nga.append(len(ga))
        R, t = pose_to_Rt(*init_pose(k, pose))
        Rs.append(R.reshape(4))
        ts.append(t)
        poses.append(pose)
    pts = np.ascontiguousarray(np.concatenate(pts, axis=0)) if pts else np.zeros((0, 2))
    return ScanBatch(pts, np.array(off, dtype=np.int32), np.array(nga, dtype=np.int32),
                     np.ascontiguousarray(np.array(Rs)).reshape(n_scans, 4),
                     np.ascontiguousarray(np.array(ts)).reshape(n_scans, 2),
                     np.array(poses, dtype=np.float64).reshape(n_scans, 3))


# ------------------------------------------------------------------ 3-D clouds
# BASELINE config 3: a 64-ring spinning lidar (elevation -24.8..+2 deg, 2048
# azimuth steps = 131 072 rays) in the same world, walls extruded to 3 m above a
# ground plane 1.73 m below the sensor.
GROUND_Z = -1.73
WALL_HEIGHT = 3.0
RING_EL_DEG = (-24.8, 2.0)


def make_cloud3d(k, n_loop=50, rings=64, n_az=2048, seed_base=9000, max_range=100.0):
    """Point cloud (sensor frame, f32 [n,3]) of the k-th pose of the loop; returns (xyz, pose)."""
    segs, _ = world_segments()
    x, y, th = true_pose(k, n_loop)
    rs = np.random.RandomState(seed_base + k)
    el = np.deg2rad(np.linspace(RING_EL_DEG[0], RING_EL_DEG[1], rings))
    az = np.deg2rad(np.arange(n_az) * (360.0 / n_az))
    EL, AZ = np.meshgrid(el, az, indexing="ij")
    EL, AZ = EL.ravel(), AZ.ravel()
    ce, se = np.cos(EL), np.sin(EL)
    dx, dy = np.cos(AZ + th), np.sin(AZ + th)            # horizontal direction in the world
    px, py = segs[:, 0][None, :], segs[:, 1][None, :]
    ex, ey = (segs[:, 2] - segs[:, 0])[None, :], (segs[:, 3] - segs[:, 1])[None, :]
    den = dx[:, None] * ey - dy[:, None] * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = ((px - x) * ey - (py - y) * ex) / den      # horizontal distance to each wall line
        s = ((px - x) * dy[:, None] - (py - y) * dx[:, None]) / den
    r = rho / ce[:, None]
    z = r * se[:, None]
    ok = (np.abs(den) > 1e-12) & (rho > 1e-9) & (s >= 0) & (s <= 1) & (z >= GROUND_Z) & (z <= GROUND_Z + WALL_HEIGHT)
    r_wall = np.where(ok, r, np.inf).min(axis=1)
    with np.errstate(divide="ignore"):
        r_ground = np.where(se < 0, GROUND_Z / se, np.inf)
    rng = np.minimum(r_wall, r_ground) + rs.normal(0.0, NOISE_SIGMA, size=len(EL))
    keep = np.isfinite(rng) & (rng > 0.5) & (rng < max_range)
    rng, ce, se, AZ = rng[keep], ce[keep], se[keep], AZ[keep]
    xyz = np.stack([rng * ce * np.cos(AZ), rng * ce * np.sin(AZ), rng * se], axis=1).astype(np.float32)
    return np.ascontiguousarray(xyz), (x, y, th)
