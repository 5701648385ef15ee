"""Evaluator feature extraction after the sampling path: the wrapper the reference's
evaluation drives (`EvaluatorModelWrapper`, codes/datasets/evaluator.py:431-493) and the two
scoring passes built on it (`evaluate_matching_score` / `evaluate_fid`,
codes/tools/evaluation.py:56-135), with both classifiers running through `hig_eval_encoder_fwd`.

Not built (out of the tier's scope): the generated-motion dataset classes of evaluator.py:24-428
(they only call `trainer.generate` in a loop), confusion-matrix plots, log files.  The wrapper's
`get_co_embeddings` is dead code in the reference (it reads attributes the class never sets,
evaluator.py:466-482) and has no counterpart here.
"""
from collections import OrderedDict

import numpy as np
import torch

from ..models import MotionConsistencyEvalModel, MotionEncoder
from ..utils.metrics import calculate_activation_statistics, calculate_frechet_distance

EVAL_MODEL_PATH = 'checkpoints/ntu_mul/eval_model/model/best_eval_model.pth'
CONSISTENCY_MODEL_PATH = 'checkpoints/ntu_mul/consistency_eval_model/model/best_eval_model.pth'


def build_models(opt, load=True):
    """evaluator.py:413-428: both classifiers see dim_pose - 4 features (the init-pose flags are cut off by the
    wrapper) and `opt.max_motion_length` positions.  load=False leaves them at their initial values (tests,
    benchmarks: no checkpoints travel with the build)."""
    kw = dict(input_feats=opt.dim_pose - 4, num_frames=opt.max_motion_length, num_layers=opt.num_layers,
              latent_dim=opt.latent_dim)
    motion_enc = MotionEncoder(**kw)
    consistency_eval_model = MotionConsistencyEvalModel(**kw)
    if load:
        motion_enc.load_state_dict(torch.load(EVAL_MODEL_PATH, map_location="cpu"))
        consistency_eval_model.load_state_dict(torch.load(CONSISTENCY_MODEL_PATH, map_location="cpu"))
    return motion_enc, consistency_eval_model


class EvaluatorModelWrapper(object):
    """Same constructor side effects on `opt` and the same `get_motion_embeddings` as the reference class."""

    def __init__(self, opt, models=None):
        if opt.dataset_name == 't2m' or opt.dataset_name == 'ntu_mul':
            opt.dim_pose = 263
        elif opt.dataset_name == 'kit':
            opt.dim_pose = 251
        else:
            raise KeyError('Dataset not Recognized!!!')
        opt.dim_word = 300
        opt.max_motion_length = 196
        opt.dim_motion_hidden = 1024
        opt.max_text_len = 20
        opt.dim_text_hidden = 512
        opt.dim_coemb_hidden = 512

        self.encoder, self.consistency_eval_model = models if models is not None else build_models(opt)
        self.opt = opt
        self.device = opt.device
        self.encoder.to(opt.device)
        self.consistency_eval_model.to(opt.device)
        self.encoder.eval()
        self.consistency_eval_model.eval()

    def get_motion_embeddings(self, motions1, motions2, m_lens, keep_dim=False):
        """-> (class logits (B, 26), pooled feature (B, d), consistency logits (B, 2)); the last 4 features of every
        row (the init-pose flags) are dropped unless keep_dim (evaluator.py:484-493)."""
        with torch.no_grad():
            motions1 = motions1.detach().to(self.device).float()
            motions2 = motions2.detach().to(self.device).float()
            if not keep_dim:
                motions1, motions2 = motions1[:, :, :-4], motions2[:, :, :-4]
            fin_motion_embedding, motion_embedding = self.encoder(motions1, motions2, length=m_lens)
            consitency_embedding = self.consistency_eval_model(motions1, motions2, length=m_lens)
        return fin_motion_embedding, motion_embedding, consitency_embedding


def evaluate_matching_score(eval_wrapper, motion_loaders):
    """tools/evaluation.py:56-112 without the plots: per loader, classification accuracy of the generated pairs,
    the share the consistency model calls consistent (class 0), and the stacked activations FID works on.
    Batches are (class_id, _, _, motions1, motions2, m_lens) like the reference's evaluation datasets yield."""
    acc_dict, consistency_dict = OrderedDict(), OrderedDict()
    activation_dict, activation2_dict = OrderedDict(), OrderedDict()
    for name, loader in motion_loaders.items():
        hits, consistent, feats, logits = [], [], [], []
        for batch in loader:
            class_id, _, _, motions1, motions2, m_lens = batch
            fin, feat, cons = eval_wrapper.get_motion_embeddings(motions1=motions1, motions2=motions2, m_lens=m_lens)
            pred = fin.max(dim=1).indices.cpu().numpy()
            hits.extend(pred == np.asarray(class_id))
            consistent.extend(cons.max(dim=1).indices.cpu().numpy() == 0)
            feats.append(feat.cpu().numpy())
            logits.append(fin.cpu().numpy())
        activation_dict[name] = np.concatenate(feats, axis=0)
        activation2_dict[name] = np.concatenate(logits, axis=0)
        acc_dict[name] = sum(hits) / len(hits)
        consistency_dict[name] = sum(consistent) / len(consistent)
    return acc_dict, activation_dict, activation2_dict, consistency_dict


def evaluate_fid(activation_dict, reference_name='ground truth'):
    """tools/evaluation.py:115-135: FID of every activation set against the 'ground truth' one."""
    gt_mu, gt_cov = calculate_activation_statistics(activation_dict[reference_name])
    out = OrderedDict()
    for name, act in activation_dict.items():
        mu, cov = calculate_activation_statistics(act)
        out[name] = calculate_frechet_distance(gt_mu, gt_cov, mu, cov)
    return out
