"""Mirror of codes/models/__init__.py (single- and two-person denoiser, diffusion, evaluation classifiers)."""
from .evaluation_models import MotionConsistencyEvalModel, MotionEncoder
from .gaussian_diffusion import GaussianDiffusion
from .interaction_transformer import MotionInteractionTransformer
from .transformer import MotionTransformer

__all__ = ["MotionTransformer", "MotionInteractionTransformer", "MotionEncoder", "MotionConsistencyEvalModel",
           "GaussianDiffusion"]
