"""numpy helpers with the conventions of the reference's src/sdf_tools/utils_3d.py:5-108.

Input ``env`` is indexed ``[y, x, z]`` (utils_3d.py:57-60), a voxel is filled where ``env == 1``
(:65), out-of-bounds cell is COLLISION_CELL(-10000) (:21), no virtual border, unknown is free (:68).
Outputs are ``[y, x, z]`` float32 (:74-75) and, for the gradient, ``[y, x, z, 3]`` from
GetGradient(..., enable_edge_gradients=True) (:77-90).  Unlike the reference, no Python loop runs
per voxel: occupancy goes in as one array and the field comes back as one array.
"""
import numpy as np

from ._bindings import load_pysdf_tools


def _origin(origin_point):
    return [[1.0, 0.0, 0.0, origin_point[0]], [0.0, 1.0, 0.0, origin_point[1]],
            [0.0, 0.0, 1.0, origin_point[2]], [0.0, 0.0, 0.0, 1.0]]


def _grid_from_env(env, res, origin_point, frame="world"):
    m = load_pysdf_tools()
    env = np.asarray(env)
    y_shape, x_shape, z_shape = env.shape
    oob_value = m.COLLISION_CELL(-10000)
    grid = m.CollisionMapGrid(m.Isometry3d(_origin(origin_point)), frame, res, x_shape, y_shape, z_shape, oob_value)
    # cells default to the OOB cell (-10000, free); filled cells become occupancy 1 (utils_3d.py:62-67)
    occ = np.where(np.transpose(env, [1, 0, 2]) == 1, np.float32(1.0), np.float32(-10000.0))
    grid.SetOccupancyFromNumpy(np.ascontiguousarray(occ, dtype=np.float32))
    return grid, oob_value


def compute_sdf(env, res, origin_point):
    """:return: pysdf_tools.SignedDistanceField (utils_3d.py:5-36)"""
    grid, oob_value = _grid_from_env(env, res, origin_point)
    return grid.ExtractSignedDistanceField(oob_value.occupancy, False, False)[0]


def compute_sdf_and_gradient(env, res, origin_point):
    """:return: (sdf [y,x,z] float32, gradient [y,x,z,3] float32) (utils_3d.py:39-97)"""
    sdf = compute_sdf(env, res, origin_point)
    np_sdf = np.transpose(sdf.GetRawDataNumpy(), [1, 0, 2]).astype(np.float32)
    np_gradient = np.transpose(sdf.GetFullGradientNumpy(True), [1, 0, 2, 3]).astype(np.float32)
    return np_sdf, np_gradient


def get_gradient(sdf, dtype=np.float64):
    """[x, y, z, 3] gradient of a SignedDistanceField (utils_3d.py:100-108)."""
    return sdf.GetFullGradientNumpy(True).astype(dtype)
