"""Mirror of codes/trainers/__init__.py for the hot path."""
from .ddpm_trainer import DDPMTrainer
from .mul_ddpm_trainer import DDPMMulTrainer

__all__ = ["DDPMTrainer", "DDPMMulTrainer"]
