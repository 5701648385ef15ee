"""Mirror of codes/datasets/__init__.py for the input side of the hot path (two-person training set)."""
from .dataloader import build_dataloader
from .mul_dataset import DeviceMotionBank, Text2MotionMulDataset

__all__ = ["Text2MotionMulDataset", "DeviceMotionBank", "build_dataloader"]
