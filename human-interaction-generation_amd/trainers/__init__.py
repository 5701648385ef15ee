"""Trainers of the hot path: the single-person `DDPMTrainer` and the two-person `DDPMMulTrainer`
(same names the reference's `codes/trainers` package exports)."""
from . import ddpm_trainer as _single
from . import mul_ddpm_trainer as _pair

DDPMTrainer = _single.DDPMTrainer
DDPMMulTrainer = _pair.DDPMMulTrainer

__all__ = ["DDPMTrainer", "DDPMMulTrainer"]
