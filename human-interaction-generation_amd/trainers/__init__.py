"""Mirror of codes/trainers/__init__.py for the hot path."""
from .ddpm_trainer import DDPMTrainer

__all__ = ["DDPMTrainer"]
