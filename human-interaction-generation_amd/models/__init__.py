"""Mirror of codes/models/__init__.py for the hot path (single-person denoiser + diffusion)."""
from .gaussian_diffusion import GaussianDiffusion
from .transformer import MotionTransformer

__all__ = ["MotionTransformer", "GaussianDiffusion"]
