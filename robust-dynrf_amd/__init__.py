"""rodynrf (robust-dynrf_amd): the MI355X-native ray-batch hot path of RoDynRF.

Hand-written HIP kernels for gfx950 behind a C ABI (include/rodynrf.h, librodynrf.so), exposed
through the reference's own call surface:

    TensorVMSplit, TensorVMSplit_TimeEmbedding     (models/tensoRF.py: forward, compute_densityfeature /
                                                    compute_appfeature / compute_blendingfeature / warp_coordinate,
                                                    get_forward_backward_scene_flow, density_L1, TV_loss_*, ...)
    sampleXYZ, raw2outputs, OctreeRender_trilinear_fast, induce_flow, render_3d_point   (renderer.py)
    generate_rays                                  (train.py ray-generation block)

Importing this package loads librodynrf.so and raises if it is missing: there is no fallback.
"""
from . import _lib
from .fields import TensorVMSplit, TensorVMSplit_TimeEmbedding, TensorBase
from .renderer import (sampleXYZ, raw2outputs, OctreeRender_trilinear_fast, sample_rays, render_rays, render_chunks,
                       induce_flow, induce_flow_single, render_3d_point, render_single_3d_point,
                       eff_distloss, flatten_eff_distloss, render_frame, psnr)
from .ray_utils import generate_rays, ids2pixel, pose_to_mtx
from .regularizers import TVLoss
from .losses import LossTerms
from ._lib import RdrfError

__all__ = ["render_frame", "psnr", "TVLoss", "pose_to_mtx", "eff_distloss", "flatten_eff_distloss", "induce_flow", "induce_flow_single", "render_3d_point", "render_single_3d_point",
           "TensorVMSplit", "TensorVMSplit_TimeEmbedding", "TensorBase", "sampleXYZ", "raw2outputs",
           "OctreeRender_trilinear_fast", "sample_rays", "render_rays", "render_chunks", "generate_rays", "ids2pixel", "LossTerms",
           "RdrfError"]
