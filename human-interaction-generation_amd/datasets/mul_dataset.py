"""Two-person input pipeline: the reference's `Text2MotionMulDataset`
(codes/datasets/mul_dataset.py:36-253) re-done for the MI355X.

Two layers, same semantics:
  * `Text2MotionMulDataset` -- same constructor / `__getitem__` / `inv_transform` contract as the
    reference class, on the host with numpy (what a stock DataLoader consumes; parity-pinned
    bit-for-bit by tests/golden/g11_dataset.npz).
  * `DeviceMotionBank` -- the MI355X path: every motion of the split lives in ONE device buffer
    (the whole NTU two-person split is a few GB of the 288 GB), and a batch is built by a single
    gather + Z-normalise kernel (`hig_gather_frames`) straight into the (2B, 91, F) layout the
    denoiser reads: no DataLoader workers, no per-batch H2D copy of motion data.  Frame indices,
    the random shift and the caption choice are drawn on the host in the SAME `random` call order
    as the reference's `__getitem__`, so both layers produce identical batches for one seed.

On-disk format (reference :73-127): `<motion_dir>/<name>.npy` (2, frames + 1, F) whose LAST row
is the init pose; `<text_dir>/<name>.txt` lines `cap1_cap2#tokens#from#to`; split file of names.
"""
import codecs as cs
import json
import random
from os.path import join as pjoin

import numpy as np
import torch
from torch.utils import data

from .. import _lib

NUM_FRAMES = 90   # motion frames per sample (reference :185) -> 91 tokens with the init-pose row


def build_caption_table(enumerator):
    """cap2key / cap2classid of the reference (:27-33) from an `{action_id: [caption, ...]}` table
    (the reference imports its NTU table from data/NTURGBD_multi/language_labels.py)."""
    caps, cap2classid = [], {}
    for class_id, key in enumerate(enumerator):
        caps.extend(enumerator[key])
        cap2classid[enumerator[key][0]] = class_id
    return {c: i for i, c in enumerate(caps)}, cap2classid


def adjust_std(std, joints_num, feat_bias, dataset_name):
    """Feature-group re-weighting of the Z-norm scale (reference :138-157), in place."""
    j = joints_num
    std[0:1] = std[0:1] / feat_bias
    std[1:3] = std[1:3] / feat_bias
    std[3:4] = std[3:4] / feat_bias
    std[4: 4 + (j - 1) * 3] = std[4: 4 + (j - 1) * 3] / 1.0
    std[4 + (j - 1) * 3: 4 + (j - 1) * 9] = std[4 + (j - 1) * 3: 4 + (j - 1) * 9] / 1.0
    std[4 + (j - 1) * 9: 4 + (j - 1) * 9 + j * 3] = std[4 + (j - 1) * 9: 4 + (j - 1) * 9 + j * 3] / 1.0
    fc = 4 + (j - 1) * 9 + j * 3
    if dataset_name != 'ntu_mul':
        std[fc:] = std[fc:] / feat_bias
    else:
        std[fc: fc + 4] = std[fc: fc + 4].mean() / feat_bias
    return std


def frame_indices(nframes, rng=random):
    """Row indices into a (nframes + 1, F) motion for one sample (reference :185-202): the init-pose
    row first, then 90 frames -- all of them padded with the last one when the clip is short, or a
    randomly shifted window."""
    if NUM_FRAMES > nframes:
        ntoadd = max(0, NUM_FRAMES - nframes)
        padding = (nframes - 1) * np.ones(ntoadd, dtype=int)
        return np.concatenate(([nframes], np.arange(0, nframes), padding))
    lastone = NUM_FRAMES - 1
    shift_max = nframes - lastone - 1
    shift = rng.randint(0, max(0, shift_max - 1))
    return np.concatenate(([nframes], shift + np.arange(0, lastone + 1, 1)))


class Text2MotionMulDataset(data.Dataset):

    def __init__(self, opt, mean, std, split_file, times=1, w_vectorizer=None, eval_mode=False, label_path=None,
                 train_eval=False, caption_table=None):
        self.opt = opt
        self.max_length = 20
        self.times = times
        self.cap_id = opt.cap_id
        self.cap_same = opt.cap_same
        self.train_eval = train_eval
        self.w_vectorizer = w_vectorizer
        self.eval_mode = eval_mode
        self.cap2key, self.cap2classid = build_caption_table(caption_table) if caption_table else ({}, {})
        if (self.cap_id or eval_mode or train_eval) and not caption_table:
            raise ValueError("cap_id / eval modes map captions to class ids: pass caption_table="
                             "{action_id: [captions]} (the reference's ntu_action_multi_enumerator)")
        min_motion_len = {'t2m': 40, 'kit': 24, 'ntu_mul': 20, 'multi_pose': 20}.get(opt.dataset_name, 24)
        self.with_label = label_path is not None
        if self.with_label:
            with open(label_path) as f:
                self.label = json.load(f)

        with cs.open(split_file, 'r') as f:
            id_list = [line.strip() for line in f.readlines()]
        data_dict, new_name_list, length_list = {}, [], []
        for name in id_list:
            try:
                motion = np.load(pjoin(opt.motion_dir, name + '.npy'))
                current_motion_len = len(motion) if motion.ndim == 2 else len(motion[1])
                if current_motion_len < min_motion_len or current_motion_len >= 200:
                    continue
                text_data, flag = [], False
                with cs.open(pjoin(opt.text_dir, name + '.txt')) as f:
                    for line in f.readlines():
                        line_split = line.strip().split('#')
                        captions = line_split[0].split('_')
                        if len(captions) == 1:
                            captions.extend(captions)
                        f_tag, to_tag = float(line_split[2]), float(line_split[3])
                        f_tag = 0.0 if np.isnan(f_tag) else f_tag
                        to_tag = 0.0 if np.isnan(to_tag) else to_tag
                        text_dict = {'captions': captions, 'tokens': line_split[1].split(' ')}
                        if f_tag == 0.0 and to_tag == 0.0:
                            flag = True
                            text_data.append(text_dict)
                        else:   # HumanML3D-style sub-clips (:103-115)
                            n_motion = motion[int(f_tag * 20): int(to_tag * 20)]
                            if len(n_motion) < min_motion_len or len(n_motion) >= 200:
                                continue
                            new_name = random.choice('ABCDEFGHIJKLMNOPQRSTUVW') + '_' + name
                            while new_name in data_dict:
                                new_name = random.choice('ABCDEFGHIJKLMNOPQRSTUVW') + '_' + name
                            data_dict[new_name] = {'motion': n_motion, 'length': len(n_motion), 'text': [text_dict]}
                            new_name_list.append(new_name)
                            length_list.append(len(n_motion))
                if flag:
                    data_dict[name] = {'motion': motion, 'length': current_motion_len, 'text': text_data}
                    new_name_list.append(name)
                    length_list.append(current_motion_len)
            except (IOError, OSError, ValueError, IndexError):
                pass   # a listed clip without its files is skipped, as in the reference (:125-127)
        name_list, length_list = zip(*sorted(zip(new_name_list, length_list), key=lambda x: x[1]))

        if opt.limit_data_num != -1:
            np.random.seed(0)
            all_indices = np.arange(len(name_list))
            np.random.shuffle(all_indices)
            name_list = [name_list[ind] for ind in all_indices[:opt.limit_data_num]]
            length_list = [length_list[ind] for ind in all_indices[:opt.limit_data_num]]
            data_dict = {key: data_dict[key] for key in name_list}

        if opt.is_train:
            adjust_std(std, opt.joints_num, opt.feat_bias, opt.dataset_name)
            np.save(pjoin(opt.meta_dir, 'mean.npy'), mean)
            np.save(pjoin(opt.meta_dir, 'std.npy'), std)

        self.mean, self.std = mean[:-4], std[:-4]
        self.init_mean, self.init_std = mean[-4:], std[-4:]
        self.length_arr = np.array(length_list)
        self.data_dict = data_dict
        self.name_list = name_list

    def inv_transform(self, data):
        return data * self.std + self.mean

    def real_len(self):
        return len(self.data_dict)

    def __len__(self):
        return self.real_len() * self.times

    def _draw(self, item):
        """The random decisions of one `__getitem__`, in the reference's call order."""
        idx = item % self.real_len()
        file_id = self.name_list[idx]
        d = self.data_dict[file_id]
        frame_ix = frame_indices(d['motion'].shape[1] - 1)
        text_data = random.choice(d['text'])
        caption1, caption2 = text_data['captions']
        if self.cap_id:
            caption1, caption2 = [self.cap2key[caption1]], [self.cap2key[caption2]]
        elif self.cap_same:
            caption2 = caption1
        swap = bool(self.with_label and not (self.eval_mode or self.train_eval) and self.label[file_id] != 0)
        return file_id, d, frame_ix, caption1, caption2, swap

    def __getitem__(self, item):
        file_id, d, frame_ix, caption1, caption2, swap = self._draw(item)
        motion = d['motion']
        motion1, motion2 = motion[0][frame_ix], motion[1][frame_ix]
        motion1[1:] = (motion1[1:] - self.mean) / self.std
        motion2[1:] = (motion2[1:] - self.mean) / self.std
        motion1[0, :4] = (motion1[0, :4] - self.init_mean) / self.init_std
        motion2[0, :4] = (motion2[0, :4] - self.init_mean) / self.init_std
        if self.eval_mode or self.train_eval:
            return self.cap2classid[caption1], motion1, motion2, d['length'], file_id
        if swap:
            motion1, motion2 = motion2, motion1
        return caption1, caption2, motion1, motion2, d['length'], file_id


class DeviceMotionBank(object):
    """All motions of a `Text2MotionMulDataset` resident in HBM + the fused batch builder."""

    def __init__(self, dataset, device):
        self.ds = dataset
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceMotionBank: ROCm device required (no CPU fallback)")
        offs, chunks, o = {}, [], 0
        for name in dataset.name_list:
            m = np.ascontiguousarray(dataset.data_dict[name]['motion'], dtype=np.float32)
            assert m.ndim == 3 and m.shape[0] == 2, "two-person motions are (2, frames + 1, F)"
            offs[name] = (o, m.shape[1], m.shape[2])
            chunks.append(m.reshape(-1))
            o += m.size
        self.F = int(dataset.data_dict[dataset.name_list[0]]['motion'].shape[2])
        self.offsets = offs
        self.bank = torch.from_numpy(np.concatenate(chunks)).to(self.device)
        # the statistics keep their on-disk precision: numpy normalises in float64 when mean / std are
        # float64 (and rounds once on the store into the float32 motion), in float32 otherwise
        self.stat_f64 = dataset.mean.dtype == np.float64 or dataset.std.dtype == np.float64
        sdt = torch.float64 if self.stat_f64 else torch.float32
        st = [torch.from_numpy(np.ascontiguousarray(a)).to(sdt) for a in
              (dataset.mean, dataset.std, dataset.init_mean, dataset.init_std)]
        self.stats = torch.cat(st).to(self.device)       # [mean F | std F | init_mean 4 | init_std 4]

    def make_batch(self, items):
        """items: dataset indices of one batch -> (caption1, caption2, motion1, motion2, m_lens, file_ids)
        with motion1 / motion2 (B, 91, F) device tensors, views of ONE (2B, 91, F) buffer laid out as the
        two-person denoiser wants it (`DDPMMulTrainer.forward` concatenates them again at no cost)."""
        B, T, F = len(items), NUM_FRAMES + 1, self.F
        seq_off = np.empty(2 * B, dtype=np.int64)
        frame_ix = np.empty((2 * B, T), dtype=np.int32)
        cap1, cap2, lens, ids = [], [], [], []
        for b, item in enumerate(items):
            file_id, d, fix, c1, c2, swap = self.ds._draw(item)
            base, rows, _ = self.offsets[file_id]
            p1, p2 = (1, 0) if swap else (0, 1)
            seq_off[b] = base + p1 * rows * F
            seq_off[B + b] = base + p2 * rows * F
            frame_ix[b] = frame_ix[B + b] = fix
            cap1.append(c1)
            cap2.append(c2)
            lens.append(d['length'])
            ids.append(file_id)
        off_d = torch.from_numpy(seq_off).to(self.device, non_blocking=True)
        fix_d = torch.from_numpy(frame_ix).to(self.device, non_blocking=True)
        out = torch.empty(2 * B, T, F, device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().hig_gather_frames(
            _lib.ptr(self.bank), _lib.ptr(off_d), _lib.ptr(fix_d), _lib.ptr(self.stats), int(self.stat_f64),
            2 * B, T, F, _lib.ptr(out), _lib.stream_ptr()))
        return cap1, cap2, out[:B], out[B:], torch.tensor(lens), ids
