"""Non-uniform sampling protocols of the reference's data loader, vectorised.

utils/DataSet.py:364-646 selects points one at a time in Python loops (one `RandomState.random_sample()` per point,
repeated passes over the cloud until `numPoints` are collected): ~1 s for a 100k-point room. The functions here make the
same decisions with array operations and consume the `numpy.random.RandomState` exactly as the loops do -- same outputs,
same generator state afterwards -- so a loader can swap them in without changing a training run. Host-side NumPy only:
this module is a producer of inputs for the GPU path, not part of it.

Protocols (DataSet.py:666-671): 1 split, 2 gradient, 3 lambert, 4 occlusion; uniform sampling is `RandomState.choice`.
"""
import numpy as np


def _take(points, features, labels, order):
    return points[order], (None if features is None else features[order]), (None if labels is None else labels[order])


def _collect(prob, num_points, rs):
    """Indices accepted by `for i: if rs.random_sample() < prob[i]` repeated until num_points are collected (one pass if
    num_points == 0); leaves `rs` in the state the per-point loop would leave it in."""
    n = prob.shape[0]
    out = []
    have = 0
    while True:
        state = rs.get_state()
        r = rs.random_sample(n)
        acc = np.nonzero(r < prob)[0]
        if num_points > 0 and have + acc.shape[0] >= num_points:
            acc = acc[:num_points - have]
            rs.set_state(state)
            rs.random_sample(int(acc[-1]) + 1)  # the loop stops drawing right after the last accepted point
            out.append(acc)
            break
        out.append(acc)
        have += acc.shape[0]
        if num_points == 0:
            break
        if n == 0 or (have == 0 and not np.any(prob > 0)):
            raise RuntimeError("no point can be selected")
    return np.concatenate(out) if out else np.zeros(0, np.int64)


def _cycle(mask, num_points):
    """Deterministic variant (occlusion): indices with mask set, cycling over the cloud until num_points."""
    idx = np.nonzero(mask)[0]
    if num_points == 0:
        return idx
    if idx.shape[0] == 0:
        raise RuntimeError("no point can be selected")
    reps = -(-num_points // idx.shape[0])
    return np.tile(idx, reps)[:num_points]


def _bbox(points):
    cmax = np.amax(points, axis=0)
    cmin = np.amin(points, axis=0)
    size = cmax - cmin
    return cmin, size, int(np.argmax(size))


def sample_split(rs, points, features=None, labels=None, num_points=0, low_probability=0.25):
    """_non_uniform_sampling_split_ (DataSet.py:364-428): probability 1 in the upper half of the longest box axis,
    `low_probability` in the lower half."""
    cmin, size, ax = _bbox(points)
    pos = (points[:, ax] - cmin[ax]) / size[ax]
    prob = np.where(pos > 0.5, 1.0, low_probability)
    return _take(points, features, labels, _collect(prob, num_points, rs))


def sample_gradient(rs, points, features=None, labels=None, num_points=0):
    """_non_uniform_sampling_gradient_ (DataSet.py:431-492): sqrt(clip((x - 0.2 L) / (0.6 L), 0.01, 1)) along the longest
    axis."""
    cmin, size, ax = _bbox(points)
    p = (points[:, ax] - cmin[ax] - size[ax] * 0.2) / (size[ax] * 0.6)
    prob = np.power(np.clip(p, 0.01, 1.0), 1.0 / 2.0)
    return _take(points, features, labels, _collect(prob, num_points, rs))


def sample_lambert(rs, view_dir, points, normals, features=None, labels=None, num_points=0):
    """_non_uniform_sampling_lambert_ (DataSet.py:495-548): sqrt(clip(view . normal, 0, 1))."""
    d = normals @ view_dir
    prob = np.power(np.clip(d, 0.0, 1.0), 0.5)
    return _take(points, features, labels, _collect(prob, num_points, rs))


def sample_occlusion(view_dir, points, normals, features=None, labels=None, num_points=0, screen_resolution=128):
    """_non_uniform_sampling_occlusion_ (DataSet.py:551-646): orthographic z-buffer of the points facing the camera, a
    point is kept if it lies within 0.01 of the nearest depth of its pixel. No random numbers."""
    x_vec = np.cross(view_dir, np.array([0.0, 1.0, 0.0]))
    x_vec = x_vec / np.linalg.norm(x_vec)
    y_vec = np.cross(x_vec, view_dir)
    y_vec = y_vec / np.linalg.norm(y_vec)
    cmax = np.amax(points, axis=0)
    cmin = np.amin(points, axis=0)
    diagonal = np.linalg.norm(cmax - cmin) * 0.5
    center = (cmax + cmin) * 0.5
    size = screen_resolution
    pixel = diagonal / (float(size) * 0.5)
    screen_pos = center - view_dir * diagonal - x_vec * diagonal - y_vec * diagonal
    n = points.shape[0]
    facing = (normals @ view_dir) < 0.0
    t = points - screen_pos
    tx, ty = t @ x_vec, t @ y_vec
    tz = (t @ view_dir) / (diagonal * 2.0)
    px = np.full(n, -1, np.int64)
    py = np.full(n, -1, np.int64)
    zv = np.ones(n)
    px[facing] = np.floor(tx[facing] / pixel).astype(np.int64)
    py[facing] = np.floor(ty[facing] / pixel).astype(np.int64)
    zv[facing] = tz[facing]
    if np.any(px[facing] >= size) or np.any(py[facing] >= size) or np.any(px[facing] < -size) or np.any(py[facing] < -size):
        raise IndexError("projected pixel outside the screen")  # the reference's list indexing raises here as well
    # z-buffer = per-pixel minimum over the facing points (empty = -1); negative ids wrap like Python indexing
    zbuf = np.full((size, size), np.inf)
    np.minimum.at(zbuf, (px[facing] % size, py[facing] % size), zv[facing])
    zbuf[np.isinf(zbuf)] = -1.0
    keep = (zv - zbuf[px % size, py % size]) < 0.01
    return _take(points, features, labels, _cycle(keep, num_points))


def random_view(rs):
    """The view direction the batcher draws for protocols 3 and 4 (DataSet.py:790-796)."""
    v = (rs.rand(3) * 2.0) - 1.0
    return v / np.linalg.norm(v)
