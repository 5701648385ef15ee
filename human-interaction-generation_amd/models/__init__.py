"""Mirror of codes/models/__init__.py for the hot path (single- and two-person denoiser + diffusion)."""
from .gaussian_diffusion import GaussianDiffusion
from .interaction_transformer import MotionInteractionTransformer
from .transformer import MotionTransformer

__all__ = ["MotionTransformer", "MotionInteractionTransformer", "GaussianDiffusion"]
