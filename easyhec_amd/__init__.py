"""easyhec_amd -- MI355X-native differentiable silhouette rasterizer for EasyHeC's pose-optimisation hot path.

    import easyhec_amd.dr as dr                       # RasterizeCudaContext / rasterize / interpolate / antialias
    from easyhec_amd.renderer import NVDiffrastRenderer
    from easyhec_amd.rb_solver import RBSolver
    from easyhec_amd.trainer import RBSolverTrainer

The compute path is libehr_hip.so (hand-written HIP for gfx950, C ABI in include/ehr.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
