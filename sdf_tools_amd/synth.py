"""Synthetic occupancy inputs of BASELINE.md section 4 / SURVEY.md 8(d).

Counter-based generators: every voxel's value is a pure function of
(seed, linear index), so any slab of a grid can be produced independently on
any rank and the host (numpy) and device (torch) versions agree bit for bit.
Layout everywhere: ``index = x*ny*nz + y*nz + z`` (z fastest), the reference's
VoxelGrid layout (src/sdf_tools/utils_3d.py:71-73).
"""
import numpy as np

_M64 = (1 << 64) - 1


def _splitmix64_np(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def bernoulli_mask(shape, p=0.5, seed=1, x_range=None):
    """uint8 mask [nx, ny, nz] (1 = filled) with i.i.d. Bernoulli(p) occupancy.

    x_range=(x0, x1) returns only the slab x0 <= x < x1 of the full grid (same
    values the full grid would hold there)."""
    nx, ny, nz = (int(s) for s in shape)
    x0, x1 = (0, nx) if x_range is None else x_range
    plane = ny * nz
    thr = np.uint64(min(int(p * 2.0 ** 64), _M64))
    out = np.empty((x1 - x0, ny, nz), dtype=np.uint8)
    with np.errstate(over="ignore"):
        key = _splitmix64_np(np.uint64(seed & _M64))
        step = max(1, (1 << 22) // max(plane, 1))
        for xs in range(x0, x1, step):
            xe = min(x1, xs + step)
            idx = np.arange(xs * plane, xe * plane, dtype=np.uint64)
            h = _splitmix64_np(idx ^ key)
            out[xs - x0:xe - x0] = (h < thr).reshape(xe - xs, ny, nz)
    return out


def bernoulli_mask_torch(shape, p=0.5, seed=1, x_range=None, device="cuda"):
    """Device-side twin of :func:`bernoulli_mask` (bit-identical), so bench inputs
    are generated directly in HBM."""
    import torch

    nx, ny, nz = (int(s) for s in shape)
    x0, x1 = (0, nx) if x_range is None else x_range
    plane = ny * nz

    def s64(v):  # python int (mod 2^64) -> signed int64 value
        v &= _M64
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(t, k):  # logical shift right on int64
        return (t >> k) & ((1 << (64 - k)) - 1)

    def mix(t):
        t = t + s64(0x9E3779B97F4A7C15)
        t = (t ^ lsr(t, 30)) * s64(0xBF58476D1CE4E5B9)
        t = (t ^ lsr(t, 27)) * s64(0x94D049BB133111EB)
        return t ^ lsr(t, 31)

    with np.errstate(over="ignore"):
        key = int(_splitmix64_np(np.uint64(seed & _M64)))
    thr = min(int(p * 2.0 ** 64), _M64)
    out = torch.empty((x1 - x0, ny, nz), dtype=torch.uint8, device=device)
    step = max(1, (1 << 24) // max(plane, 1))
    for xs in range(x0, x1, step):
        xe = min(x1, xs + step)
        idx = torch.arange(xs * plane, xe * plane, dtype=torch.int64, device=device)
        h = mix(idx ^ s64(key))
        # unsigned compare h < thr  <=>  (h ^ MIN) < (thr ^ MIN) signed
        lt = (h ^ s64(1 << 63)) < s64(thr ^ (1 << 63))
        out[xs - x0:xe - x0] = lt.reshape(xe - xs, ny, nz).to(torch.uint8)
    return out


def spheres_mask(shape, n_spheres=12, r_range=(3, 9), seed=0):
    """Stress scene of SURVEY.md 8(d): random solid spheres in an otherwise free grid."""
    rng = np.random.RandomState(seed)
    nx, ny, nz = shape
    X, Y, Z = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    m = np.zeros(shape, dtype=np.uint8)
    for _ in range(n_spheres):
        c = rng.uniform(0, 1, 3) * np.array(shape)
        r = rng.uniform(*r_range)
        m |= (((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) <= r * r).astype(np.uint8)
    return m


def two_box_points(n_points, seed=0, lo=(0.0, 0.0, 0.0), scale=1.0):
    """Point-cloud pattern of scripts/3d_sdf_demo_rviz.py:15-19 (two boxes of uniform
    points), scaled; used by the streaming config."""
    rng = np.random.RandomState(seed)
    half = n_points // 2
    a = rng.uniform([0.2, 0.2, 0.0], [0.5, 0.4, 0.3], size=(half, 3))
    b = rng.uniform([0.6, 0.5, 0.0], [0.8, 0.7, 0.5], size=(n_points - half, 3))
    return (np.concatenate([a, b]) * scale + np.asarray(lo)).astype(np.float32)


def room_mask_torch(shape, device="cuda", floor=True, xwall=True, ywall=True):
    """A structured scene of the kind CollisionMapGrid callers hold (solid obstacles, most of the volume free): floor and
    two walls 2 % of the grid thick, a table top on four legs, a shelf.  Far-field everywhere, with thin, thick and
    perpendicular solids (bench.py's `structured_room` leg, tools/scene_bench.py)."""
    import torch
    nx, ny, nz = shape
    m = torch.zeros(shape, dtype=torch.uint8, device=device)
    fx, fy, fz = (lambda v: int(v * nx)), (lambda v: int(v * ny)), (lambda v: int(v * nz))
    if floor:
        m[:, :, :max(fz(0.02), 1)] = 1
    if xwall:
        m[:max(fx(0.02), 1), :, :] = 1
    if ywall:
        m[:, :max(fy(0.02), 1), :] = 1
    m[fx(0.3):fx(0.7), fy(0.3):fy(0.6), fz(0.35):fz(0.38)] = 1
    for (x, y) in ((0.31, 0.31), (0.68, 0.31), (0.31, 0.58), (0.68, 0.58)):
        m[fx(x):fx(x) + max(fx(0.02), 1), fy(y):fy(y) + max(fy(0.02), 1), :fz(0.35)] = 1
    m[fx(0.8):fx(0.98), fy(0.1):fy(0.9), fz(0.5):fz(0.55)] = 1
    return m


def tutorial_boxes_mask_torch(shape, device="cuda", solid=True):
    """The two boxes of the reference's tutorial (src/sdf_tools_tutorial.cpp:23-59) scaled to the grid, solid or as shells."""
    import torch
    nx, ny, nz = shape
    m = torch.zeros(shape, dtype=torch.uint8, device=device)
    for (x0, x1, y0, y1, z0, z1) in ((0.5, 0.7, 0.5, 0.6, 0.0, 0.5), (0.5, 0.75, 0.2, 0.4, 0.25, 0.5)):
        a = [int(x0 * nx), int(x1 * nx), int(y0 * ny), int(y1 * ny), int(z0 * nz), int(z1 * nz)]
        m[a[0]:a[1], a[2]:a[3], a[4]:a[5]] = 1
        if not solid:
            m[a[0] + 1:a[1] - 1, a[2] + 1:a[3] - 1, a[4] + 1:a[5] - 1] = 0
    return m


def solid_spheres_mask_torch(shape, device="cuda"):
    """Three solid spheres (radii 12 %, 20 %, 8 % of the grid): thick solids whose interiors need the far-field kernel's second pass."""
    import torch
    nx, ny, nz = shape
    X = torch.arange(nx, device=device, dtype=torch.float32)[:, None, None]
    Y = torch.arange(ny, device=device, dtype=torch.float32)[None, :, None]
    Z = torch.arange(nz, device=device, dtype=torch.float32)[None, None, :]
    m = torch.zeros(shape, dtype=torch.bool, device=device)
    n = float(min(shape))
    for (cx, cy, cz, r) in ((0.3, 0.3, 0.3, 0.12), (0.7, 0.6, 0.4, 0.2), (0.5, 0.8, 0.8, 0.08)):
        m |= (X - cx * nx) ** 2 + (Y - cy * ny) ** 2 + (Z - cz * nz) ** 2 <= (r * n) ** 2
    return m.to(torch.uint8)
