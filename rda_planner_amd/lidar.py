"""
Lidar front end of the reference's lidar demo: `scan_box` of example/lidar_nav/lidar_path_track.py:20-60 -
range scan -> points in the robot frame -> DBSCAN(eps=2.0, min_samples=6) -> one minimum-area rectangle per cluster ->
4-vertex `Rpositive` obstacles in the world frame for `MPC.control`.

The reference takes DBSCAN from scikit-learn and the rectangle from OpenCV (`cv2.minAreaRect` + `cv2.boxPoints`); OpenCV
is not available here, so both steps are restated in numpy:
  * `dbscan` reproduces scikit-learn's labelling exactly (clusters numbered by their first core point in index order,
    a border point goes to the first cluster that reaches it, noise = -1); tests pin it against sklearn.cluster.DBSCAN.
  * `min_area_rect` is the rotating-calipers rectangle over the convex hull (exact in float64; OpenCV works in float32
    and may return the corners in another order / starting corner - `MPC` re-orders vertices anyway, mpc.py:476-490).
    A cluster of collinear points gives a zero-width rectangle in OpenCV, which the reference's half-space conversion
    turns into an infinite line; here such a rectangle is given `min_width` (1 cm) so that it stays a bounded obstacle.
This is host code, like in the reference (a few hundred points per scan); the solver behind `MPC.control` is the GPU path.
"""
from collections import namedtuple

import numpy as np

Obstacle = namedtuple("obstacle", "center radius vertex cone_type velocity")


def scan_points(scan_data):
    """hits of a scan (`ranges`, `angle_min`, `angle_max`, `range_max`) as an (n, 2) array in the sensor frame;
    beams at (range_max - 0.01) or beyond are misses (lidar_path_track.py:27-35)"""
    ranges = np.asarray(scan_data["ranges"], float)
    angles = np.linspace(scan_data["angle_min"], scan_data["angle_max"], len(ranges))
    hit = ranges < scan_data["range_max"] - 0.01
    return np.stack((ranges[hit] * np.cos(angles[hit]), ranges[hit] * np.sin(angles[hit])), axis=1)


def dbscan(points, eps=2.0, min_samples=6):
    """labels of scikit-learn's DBSCAN (euclidean, the point itself counts as a neighbour)"""
    X = np.asarray(points, float)
    n = len(X)
    labels = np.full(n, -1, dtype=np.int64)
    if n == 0:
        return labels
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(axis=2)
    near = d2 <= eps * eps
    core = near.sum(axis=1) >= min_samples
    cluster = 0
    for i in range(n):
        if labels[i] != -1 or not core[i]:
            continue
        labels[i] = cluster                       # depth-first expansion, like sklearn's dbscan_inner
        stack = [i]
        while stack:
            j = stack.pop()
            if core[j]:
                for k in np.flatnonzero(near[j]):
                    if labels[k] == -1:
                        labels[k] = cluster
                        stack.append(k)
        cluster += 1
    return labels


def _cross(a, b):
    return a[0] * b[1] - a[1] * b[0]


def convex_hull(points):
    """counter-clockwise hull (Andrew's monotone chain), collinear points dropped"""
    P = np.unique(np.asarray(points, float), axis=0)
    if len(P) <= 2:
        return P
    P = P[np.lexsort((P[:, 1], P[:, 0]))]

    def half(seq):
        out = []
        for p in seq:
            while len(out) >= 2 and _cross(out[-1] - out[-2], p - out[-2]) <= 0:
                out.pop()
            out.append(p)
        return out
    lower, upper = half(P), half(P[::-1])
    return np.array(lower[:-1] + upper[:-1])


def min_area_rect(points, min_width=0.01):
    """(4, 2) corners, counter-clockwise, of the minimum-area enclosing rectangle (one side collinear with a hull edge)"""
    H = convex_hull(points)
    if len(H) == 1:
        c, u, lo, hi, wlo, whi = H[0], np.array([1.0, 0.0]), 0.0, 0.0, 0.0, 0.0
    else:
        best = None
        m = len(H)
        for i in range(m if m > 2 else 1):
            e = H[(i + 1) % m] - H[i]
            u = e / np.linalg.norm(e)
            v = np.array([-u[1], u[0]])
            a, b = H @ u, H @ v
            area = (a.max() - a.min()) * (b.max() - b.min())
            if best is None or area < best[0]:
                best = (area, u, a.min(), a.max(), b.min(), b.max())
        _, u, lo, hi, wlo, whi = best
        c = np.zeros(2)
    v = np.array([-u[1], u[0]])
    if hi - lo < min_width:
        mid = 0.5 * (lo + hi); lo, hi = mid - 0.5 * min_width, mid + 0.5 * min_width
    if whi - wlo < min_width:
        mid = 0.5 * (wlo + whi); wlo, whi = mid - 0.5 * min_width, mid + 0.5 * min_width
    return np.array([c + lo * u + wlo * v, c + hi * u + wlo * v, c + hi * u + whi * v, c + lo * u + whi * v])


def scan_box(state, scan_data, eps=2.0, min_samples=6):
    """the reference's `scan_box`: obstacles (4-vertex boxes in the world frame) seen by one scan from `state`"""
    pts = scan_points(scan_data)
    if len(pts) < 4:
        return []
    labels = dbscan(pts, eps, min_samples)
    state = np.asarray(state, float)
    rot = state[2, 0]
    R = np.array([[np.cos(rot), -np.sin(rot)], [np.sin(rot), np.cos(rot)]])
    out = []
    for label in np.unique(labels):
        if label == -1:
            continue
        box = min_area_rect(pts[labels == label])
        out.append(Obstacle(None, None, state[0:2] + R @ box.T, "Rpositive", np.zeros((2, 1))))
    return out
