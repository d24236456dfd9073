"""2-D helpers with the conventions of the reference's src/sdf_tools/utils_2d.py:6-58: ``grid_world``
is ``[y, x]`` of 0/1, the grid has one cell in z, outputs are ``[y, x]`` and ``[y, x, 2]`` float32."""
import numpy as np

from . import utils_3d


def compute_sdf_and_gradient(grid_world, sdf_resolution, sdf_origin, frame="world"):
    grid_world = np.asarray(grid_world)
    env = grid_world[:, :, None]
    grid, oob_value = utils_3d._grid_from_env(env, sdf_resolution, [sdf_origin[0], sdf_origin[1], 0.0], frame)
    sdf = grid.ExtractSignedDistanceField(oob_value.occupancy, False, False)[0]
    np_sdf = np.transpose(sdf.GetRawDataNumpy()[:, :, 0], [1, 0]).astype(np.float32)
    grad = sdf.GetFullGradientNumpy(True)[:, :, 0, 0:2]          # drop the z gradient (utils_2d.py:52)
    return np_sdf, np.transpose(grad, [1, 0, 2]).astype(np.float32)
