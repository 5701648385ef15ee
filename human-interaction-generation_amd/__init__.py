"""MI355X-native hot path of line/Human-Interaction-Generation: the text-conditioned motion
diffusion denoiser (MotionTransformer), GaussianDiffusion and DDPMTrainer with the reference's
Python API, running on hand-written gfx950 HIP kernels behind the C ABI in include/hig.h.

The directory name carries a hyphen (it mirrors the upstream repository name), so it is imported
through the `hig_amd` shim package at the repository root:  `import hig_amd`.
Putting this directory itself on sys.path gives the reference's own import layout
(`from models.transformer import MotionTransformer`, `from trainers.ddpm_trainer import DDPMTrainer`).
"""
from . import _lib  # noqa: F401
from .models import (GaussianDiffusion, MotionConsistencyEvalModel, MotionEncoder,  # noqa: F401
                     MotionInteractionTransformer, MotionTransformer)
from .trainers import DDPMMulTrainer, DDPMTrainer  # noqa: F401

__all__ = ["MotionTransformer", "MotionInteractionTransformer", "MotionEncoder", "MotionConsistencyEvalModel",
           "GaussianDiffusion", "DDPMTrainer", "DDPMMulTrainer"]
