"""The few configuration fields the hot path reads, with the reference's names and defaults
(/root/reference/easyhec/config/defaults.py:14-16, :60, :66-80, :138, :150-153; configs/xarm7/example.yaml:42-46).
The reference's yacs tree, CLI and path catalog are out of scope."""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

__all__ = ["RBSolverCfg", "SolverCfg", "ModelCfg", "Cfg", "XARM7_K_1280x720", "FRANKA_K_1920x1080"]

# defaults.py:14-16 (sim_mask_data.K, 1280x720 RealSense intrinsics)
XARM7_K_1280x720 = [[906.80517578125, 0.0, 650.1978759765625],
                    [0.0, 906.6802978515625, 367.71429443359375],
                    [0.0, 0.0, 1.0]]
# defaults_franka.py:100-102
FRANKA_K_1920x1080 = [[1352.2, 0.0, 963.3], [0.0, 1352.4, 529.4], [0.0, 0.0, 1.0]]


@dataclass
class RBSolverCfg:
    init_Tc_c2b: Sequence = field(default_factory=list)   # 4x4, eye-to-hand
    mesh_paths: List[str] = field(default_factory=list)   # one mesh per link (STL / DAE / PLY)
    H: int = 720
    W: int = 1280
    use_fused: bool = True     # fused mask-loss kernel (default) or the three drop-in ops exactly as the reference
    # use_fused=False only: issue EXACTLY the calls of the reference's own files (nvdiffrast_renderer.py:33-47 inside the loop
    # of rb_solver.py:58-71) -- what a maintainer gets who swaps `import nvdiffrast.torch as dr` and changes nothing else:
    # K_to_projection, ones(verts.shape) and transform_pos per call, three colour channels, rast_db written (grad_db default),
    # rast not detached, no topology argument, one flip / stack / clamp per frame as written.  False = this repo's
    # optimised mirror of the same schedule (renderer.NVDiffrastRenderer: cached constants, one channel, batched glue)
    reference_schedule: bool = False
    # use_fused=False only: the three ops called ONCE per step over all (view, link) images -- nvdiffrast's range mode: one
    # concatenated vertex / triangle array, a (start, count) range per image -- instead of once per image.  Same ops, same
    # arithmetic per image; what changes is that a call's latency chain is paid once for B x L images instead of B x L times
    batched_ops: bool = False
    # use_fused=False, per-image schedule only: frames' render chains on this many HIP streams, each with its own rasterizer
    # context (renderer.link_lanes), so that the independent chains overlap; 0 / 1 = everything on the step's stream;
    # -1 = automatic: 2 when RBSolverTrainer(graph=True) replays the step from a graph (parallel branches), 1 otherwise
    render_lanes: int = -1


@dataclass
class ModelCfg:
    rbsolver: RBSolverCfg = field(default_factory=RBSolverCfg)
    device: str = "cuda"


@dataclass
class SolverCfg:
    optimizer: str = "Adam"            # solver/build.py:24-27
    max_lr: float = 0.003              # example.yaml:44
    weight_decay: float = 0.0005       # defaults.py:138 (L2 added to the gradient by torch.optim.Adam)
    num_epochs: int = 1000             # == Adam iterations: one batch per epoch (example.yaml:43)
    do_grad_clip: bool = False         # defaults.py:153
    grad_clip_type: str = "norm"
    grad_clip: float = 1.0
    log_interval: int = 100
    save_freq: int = 20
    batch_size: int = 100


@dataclass
class Cfg:
    model: ModelCfg = field(default_factory=ModelCfg)
    solver: SolverCfg = field(default_factory=SolverCfg)
    output_dir: Optional[str] = None
    dbg: bool = False
