"""Drop-in modules for the two un-vendored CUDA dependencies of the reference's renderer, on top of the HIP rasterizer:

    siu3r_amd.compat.diff_gaussian_rasterization   ->  `from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
                                                       (reference src/models/cuda_splatting.py:9-12, call site :82-118)
    siu3r_amd.compat.gsplat                        ->  `from gsplat import rasterization` (reference src/models/gaussian_renderer.py:7, call site :92-106)

A reference checkout picks them up with two import lines changed (seams 2 and 3).  Inference only: no autograd."""
