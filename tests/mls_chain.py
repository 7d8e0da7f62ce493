"""MLS::addToMap(cloud, pose) in rolling mode, cloud after cloud (mls.cpp:34-150 with setPose :408-479), put together from
oracle pieces: what the C++ adapter (include/slam_amd/mls.hpp) and the local_mapper node over it must hold after every
cloud.  Test infrastructure."""
import numpy as np

import oracle_lib as O


def roll(plane, dx, dy, fill=0):
    """Window after Grid::shiftOrigin(dx,dy) + the clears of mls.cpp:433-477: cell (i,j) shows old (i+dx, j+dy)."""
    sy, sx = plane.shape
    out = np.full_like(plane, fill)
    xs = np.arange(sx) + dx
    ys = np.arange(sy) + dy
    vx = (xs >= 0) & (xs < sx)
    vy = (ys >= 0) & (ys < sy)
    out[np.ix_(vy, vx)] = plane[np.ix_(ys[vy], xs[vx])]
    return out


def oracle_local_map(clouds, poses, size=200, res=0.2, min_cluster_points=20):
    """clouds: f32 [n, 3] each, sensor frame; poses: x y z qx qy qz qw each.  Returns one snapshot per cloud:
    (window centre x, y, occupancy int8 [size * size], drv points, ground points, global_cloud before any filter)."""
    gp = O.grid_params(size, size, res, min_cluster_points=min_cluster_points, rolling=1)
    num, drv, eocc = np.zeros(size * size), np.full(size * size, -1, np.int8), np.full(size * size, -1, np.int8)
    cx = cy = 0.0
    gc = np.zeros((0, 3), np.float32)
    snaps = []
    for k in range(len(clouds)):
        px, py = poses[k][0], poses[k][1]
        dx, dy = int(np.round((px - cx) / res)), int(np.round((py - cy) / res))     # mls.cpp:419-424
        if dx or dy:
            num = roll(num.reshape(size, size), dx, dy).reshape(-1)
            drv = roll(drv.reshape(size, size), dx, dy, -1).reshape(-1)
            eocc = roll(eocc.reshape(size, size), dx, dy, -1).reshape(-1)
            cx += dx * res
            cy += dy * res
            gc = gc + np.array([-(dx * res), -(dy * res), 0], np.float32)             # :433-454
            crop = np.float32(size * res / 2)
            gc = gc[(gc[:, 0] >= -crop) & (gc[:, 0] <= crop) & (gc[:, 1] >= -crop) & (gc[:, 1] <= crop)]
        # tf's matrix from the quaternion the program was given (not from the angle): the same doubles
        q = poses[k][3:]
        dd = sum(v * v for v in q)
        s2 = 2.0 / dd
        xs, ys, zs = q[0] * s2, q[1] * s2, q[2] * s2
        wx, wy, wz, xx, xy, xz, yy, yz, zz = q[3] * xs, q[3] * ys, q[3] * zs, q[0] * xs, q[0] * ys, q[0] * zs, q[1] * ys, q[1] * zs, q[2] * zs
        Rm = np.array([[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]])
        P = clouds[k].astype(np.float64)
        T = np.stack([Rm[0, 0] * P[:, 0] + Rm[0, 1] * P[:, 1] + Rm[0, 2] * P[:, 2] + (cx - px),
                      Rm[1, 0] * P[:, 0] + Rm[1, 1] * P[:, 1] + Rm[1, 2] * P[:, 2] + (cy - py),
                      Rm[2, 0] * P[:, 0] + Rm[2, 1] * P[:, 1] + Rm[2, 2] * P[:, 2] + 0.0], 1).astype(np.float32)
        lab, *_ = O.gseg_segment(T)
        o, g = T[lab == O.GSEG_OBSTACLE], T[lab == O.GSEG_GROUND]
        O.grid_add_scan_inorder(gp, np.concatenate([o, np.zeros((len(o), 1), np.float32)], 1),
                                np.concatenate([g, np.zeros((len(g), 1), np.float32)], 1), num, drv, eocc)
        gc = np.concatenate([gc, o])
        snaps.append((cx, cy, eocc.copy(), len(o), len(g), gc.copy()))
    return snaps
