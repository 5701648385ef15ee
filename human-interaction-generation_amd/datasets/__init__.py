"""Mirror of codes/datasets/__init__.py for the two sides of the hot path: the two-person training set in front of
it, the evaluator's feature extraction behind it."""
from .dataloader import build_dataloader
from .evaluator import EvaluatorModelWrapper, evaluate_fid, evaluate_matching_score
from .mul_dataset import DeviceMotionBank, Text2MotionMulDataset

__all__ = ["Text2MotionMulDataset", "DeviceMotionBank", "build_dataloader", "EvaluatorModelWrapper",
           "evaluate_matching_score", "evaluate_fid"]
