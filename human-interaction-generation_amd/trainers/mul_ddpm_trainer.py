"""MI355X-native `DDPMMulTrainer`: the reference's two-person trainer
(codes/trainers/mul_ddpm_trainer.py:51-440) over the HIP two-person denoiser.

Batch layouts (reference :96-131).  With a label file: model batch = [m1 | m2] with captions
[c1 | c2].  Without one (PIT, permutation-invariant training): the noised motions are run twice,
[m1, m1 | m2, m2] with captions [c1, c2 | c2, c1], and per pair the cheaper caption assignment is
the loss (:234-243).  `multi=False` falls back to the single-person trainer's step.
"""
import os
from collections import Counter, OrderedDict

import numpy as np
import torch

from .. import _lib
from .ddpm_trainer import DDPMTrainer, _core


class DDPMMulTrainer(DDPMTrainer):

    def __init__(self, args, encoder):
        super().__init__(args, encoder)
        self.multi = args.multi
        self.with_label = args.label_path is not None
        self.cap_id = args.cap_id

    # ---- one training forward ---------------------------------------------------------------
    def _pair_inputs(self, batch_data):
        caption1, caption2, motion1, motion2, m_lens = batch_data[:5]
        motion1 = motion1.detach().to(self.device).float()
        motion2 = motion2.detach().to(self.device).float()
        x_start = torch.cat([motion1, motion2], dim=0)
        caption = list(caption1) + list(caption2)
        T = x_start.shape[1]
        cur_len = torch.LongTensor([min(T, int(m_len)) for m_len in m_lens]).to(self.device)
        return caption, list(caption1), list(caption2), x_start, cur_len, motion1.shape[0], T

    def forward(self, batch_data, eval_mode=False):
        """mul_ddpm_trainer.py:90-161."""
        if not self.multi:
            return super().forward(batch_data, eval_mode)
        caption, caption1, caption2, x_start, cur_len, B, T = self._pair_inputs(batch_data)
        t, _ = self.sampler.sample(B, x_start.device)
        t = torch.cat([t, t], dim=0)
        if not self.with_label:
            caption = caption + caption2 + caption1      # (c1, c2, c2, c1) against (m1, m1, m2, m2)
            cur_len = torch.cat([cur_len] * 4, dim=0)
            forward_twice = True
        else:
            cur_len = torch.cat([cur_len, cur_len], dim=0)
            forward_twice = False
        output = self.diffusion.training_losses(
            model=self.encoder, x_start=x_start, t=t,
            model_kwargs={"text": caption, "length": cur_len}, forward_twice=forward_twice)
        self.real_noise = output['target']
        self.fake_noise = output['pred']
        self.src_mask = _core(self.encoder).generate_src_mask(T, cur_len).to(x_start.device)

    def _token_loss(self):
        """(rows, T) per-token MSE; the init-pose token is scored on its 4 features (:226-228)."""
        init = self.mse_criterion(self.fake_noise[:, 0, :4], self.real_noise[:, 0, :4]).mean(dim=-1)
        move = self.mse_criterion(self.fake_noise[:, 1:], self.real_noise[:, 1:]).mean(dim=-1)
        return torch.cat([init.unsqueeze(1), move], dim=1)

    def backward_G(self):
        """mul_ddpm_trainer.py:223-249."""
        if not self.multi:
            return super().backward_G()
        if self.with_label:
            if _core(self.encoder).two_embed:
                loss = self._token_loss()
            else:
                loss = self.mse_criterion(self.fake_noise, self.real_noise).mean(dim=-1)
            loss_mot_rec = (loss * self.src_mask).sum() / self.src_mask.sum()
        else:
            loss = self._token_loss()
            rows = loss.shape[0]
            # per caption assignment: person-1 rows + person-2 rows  -> (2 assignments x pairs)
            per_assign = (loss * self.src_mask).sum(dim=1).view(2, rows // 2).sum(dim=0)
            loss_mot_rec = per_assign.view(2, rows // 4).min(dim=0).values.sum() / (self.src_mask.sum() / 2)
        self.loss_mot_rec = loss_mot_rec
        return OrderedDict({'loss_mot_rec': self.loss_mot_rec.item()})

    # ---- MI355X fused step (inherits train_step_fused / train_step_captured) ---------------------
    def _fused_fwd_bwd(self, x_start, t, length, xf_proj, xf_out, noise, clip_out=None, eot=None, exchange=None):
        """Two-person version of the fused forward/backward: x_start = cat([motion1, motion2]) (2B, T, F),
        t (B,) or (2B,), length (B,) per pair.  PIT mode (no label file): the noised motions run twice,
        rows [m1|c1, m1|c2, m2|c2, m2|c1] -- xf_proj / xf_out must hold the 4B text embeddings in that
        order (2B rows [c1 | c2] with a label file) -- and `hig_pair_mse` picks the cheaper caption
        assignment per pair (mul_ddpm_trainer.py:96-131, 223-247)."""
        if not self.multi:
            return super()._fused_fwd_bwd(x_start, t, length, xf_proj, xf_out, noise, clip_out=clip_out, eot=eot,
                                          exchange=exchange)
        core = _core(self.encoder)
        L = _lib.lib()
        st = self.fused_state()
        tstate = None
        if clip_out is not None:     # CLIP features of the model batch's captions (4B rows in PIT mode)
            xf_proj, xf_out, tstate = self._text_forward(clip_out, eot)
        B2, T, F = x_start.shape
        B = B2 // 2
        t2 = t if t.numel() == B2 else torch.cat([t, t])
        if noise is None:
            noise = torch.randn_like(x_start)
        x_t = self.diffusion.q_sample(x_start, t2, noise=noise)
        pit = not self.with_label
        if pit:   # gaussian_diffusion.py:996-1001 (forward_twice)
            x_t = torch.cat([x_t[:B], x_t[:B], x_t[B:], x_t[B:]])
            noise = torch.cat([noise[:B], noise[:B], noise[B:], noise[B:]])
            t2 = torch.cat([t2, t2])
        rows = x_t.shape[0]
        len_rows = torch.cat([length] * (rows // B)).contiguous()
        assert xf_proj.shape[0] == rows and xf_out.shape[0] == rows, "text embeddings must cover the model batch"
        pred, saved = core._launch_forward(x_t, t2, len_rows, xf_proj, xf_out, training=True)
        dpred = torch.empty_like(pred)
        scr = st.get("pair_scratch")
        if scr is None or scr.numel() < 2 * rows:
            scr = st["pair_scratch"] = torch.zeros(2 * rows, device=pred.device, dtype=torch.float32)
        _lib.check(L.hig_pair_mse(_lib.ptr(pred), _lib.ptr(noise.contiguous()), _lib.ptr(len_rows), rows, T, F, int(pit),
                                  _lib.ptr(st["loss"]), _lib.ptr(dpred), _lib.ptr(scr), _lib.stream_ptr()))
        hook = {} if exchange is None else dict(layer_hook=exchange.layer_done, comm_stream=exchange.stream)
        _, dxp, dxo = core._launch_backward(x_t, t2, len_rows, xf_out, saved, dpred, want_dx=False, **hook)
        if tstate is not None:
            self._text_backward(tstate, dxp, dxo)

    def train_fused_batch(self, batch_data, captured=False, noise=None):
        """forward(batch) + update() of the reference (mul_ddpm_trainer.py:90-161, 251-258) as one fused step."""
        if not self.multi:
            return super().train_fused_batch(batch_data, captured, noise)
        caption, caption1, caption2, x_start, cur_len, B, T = self._pair_inputs(batch_data)
        if not self.with_label:
            caption = caption + caption2 + caption1
        t, _ = self.sampler.sample(B, x_start.device)
        if self.cap_id:
            raise NotImplementedError("cap_id models take class embeddings, not the text head: use forward()/update()")
        clip_out, eot = self.clip_inputs(caption)
        step = self.train_step_captured if captured else self.train_step_fused
        return step(x_start.contiguous(), t, cur_len, noise=noise, clip_out=clip_out, eot=eot)

    # ---- sampling ---------------------------------------------------------------------------
    def generate_batch(self, caption1, caption2, m_lens, dim_pose):
        """mul_ddpm_trainer.py:163-199."""
        m_lens = torch.cat([m_lens, m_lens], dim=0)
        core = _core(self.encoder)
        T = min(int(m_lens.max()), core.num_frames)
        if self.cap_id:
            caption = [torch.as_tensor(caption1).view(-1), torch.as_tensor(caption2).view(-1)]
            B = len(caption1) + len(caption2)
            kwargs = {'text': caption, 'length': m_lens}
        else:
            caption = list(caption1) + list(caption2)
            B = len(caption)
            xf_proj, xf_out = core.encode_text(caption, self.device)
            kwargs = {'xf_proj': xf_proj, 'xf_out': xf_out, 'length': m_lens}
        return self.diffusion.p_sample_loop(self.encoder, (B, T, dim_pose), clip_denoised=False,
                                            progress=True, model_kwargs=kwargs)

    def generate(self, caption1, caption2, m_lens, dim_pose, batch_size=512):
        """mul_ddpm_trainer.py:201-221 -> list of [motion1, motion2] per pair."""
        N = len(caption1)
        cur_idx = 0
        self.encoder.eval()
        all_output = []
        while cur_idx < N:
            end = N if cur_idx + batch_size >= N else cur_idx + batch_size
            batch_caption1 = caption1[cur_idx:end]
            # the reference slices caption1 again for the second person of a non-final chunk (:212);
            # kept, so chunked generation gives the same conditioning as upstream
            batch_caption2 = caption2[cur_idx:] if end == N else caption1[cur_idx:end]
            batch_m_lens = m_lens[cur_idx:end]
            output = self.generate_batch(batch_caption1, batch_caption2, batch_m_lens, dim_pose)
            B = len(batch_caption1)
            motion1, motion2 = output[:B], output[B:]
            for i in range(B):
                all_output.append([motion1[i], motion2[i]])
            cur_idx += batch_size
        return all_output

    # ---- pseudo-labelling of the caption order (:313-440) --------------------------------------
    def label_batch(self, batch_data, learned_indices, t, label_mode=False):
        caption1, caption2 = batch_data[0], batch_data[1]
        file_id = batch_data[5]
        if learned_indices is not None:
            cap1 = [learned_indices[cap] for cap in caption1[0].numpy()]
            cap2 = [learned_indices[cap] for cap in caption2[0].numpy()]
            active_indices = np.argmin([cap1, cap2], axis=0)
        caption, c1, c2, x_start, cur_len, B, T = self._pair_inputs(batch_data)
        caption = caption + c2 + c1
        tt = torch.full((2 * B,), int(t), dtype=torch.long, device=x_start.device)
        cur_len = torch.cat([cur_len] * 4, dim=0)
        output = self.diffusion.training_losses(
            model=self.encoder, x_start=x_start, t=tt,
            model_kwargs={"text": caption, "length": cur_len}, forward_twice=True)
        self.real_noise = output['target']
        self.fake_noise = output['pred']
        self.src_mask = _core(self.encoder).generate_src_mask(T, cur_len).to(x_start.device)
        loss = self._token_loss()
        rows = loss.shape[0]
        per_assign = (loss * self.src_mask).sum(dim=1).view(2, rows // 2).sum(dim=0).view(2, rows // 4)
        result = per_assign.min(dim=0).indices.cpu().numpy()
        if learned_indices is not None:
            outs = [0 if int(res) == int(active_indices[i]) else 1 for i, res in enumerate(result)]
            return (outs, file_id) if label_mode else np.array(outs)
        active_list = {}
        ids1, ids2 = caption1[0].numpy(), caption2[0].numpy()
        for i, res in enumerate(result):
            a, b = int(ids1[i]), int(ids2[i])
            pair_key = '%d_%d' % (a, b)
            active_list.setdefault(pair_key, []).append('%d_%d' % ((a, b) if res == 0 else (b, a)))
        return active_list

    def eval_data(self, train_dataset, rank, world_size, learned_indices, max_class_num=42, save_dir=None):
        from ..parallel import ShardedSampler
        self.to(self.device)
        sampler = ShardedSampler(len(train_dataset), rank, world_size, shuffle=False)
        loader = torch.utils.data.DataLoader(train_dataset, batch_size=self.opt.batch_size, sampler=sampler,
                                             drop_last=False, num_workers=getattr(self.opt, "num_workers", 4))
        self.eval_mode()
        steps = range(830, 940, 30)
        with torch.no_grad():
            if learned_indices is None:
                merged = {}
                for t in steps:
                    for batch_data in loader:
                        for _ in range(5):
                            for key, v in self.label_batch(batch_data, None, t=t).items():
                                merged.setdefault(key, []).extend(v)
                learned_indices = []
                for i in range(max_class_num + 1):
                    if '%d_%d' % (i - 1, i) in merged:
                        continue
                    key = '%d_%d' % (i, i + 1)
                    if key in merged:
                        num1, num2 = Counter(merged[key]).most_common()[0][0].split('_')
                        learned_indices += [int(num1), int(num2)]
                    else:
                        learned_indices.append(i)
                return learned_indices
            for batch_data in loader:
                votes = {}
                for t in steps:
                    for _ in range(41):
                        outs, file_ids = self.label_batch(batch_data, learned_indices, t=t, label_mode=True)
                        for fid, o in zip(file_ids, outs):
                            votes.setdefault(fid, []).append(o)
                for key, v in votes.items():
                    with open(os.path.join(save_dir, key + '.txt'), 'w') as f:
                        f.write(str(Counter(v).most_common()[0][0]))
