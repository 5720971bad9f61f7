"""BriVIS meta-architecture — mirror of openvis/brivis.py:26-265 (eval path, WINDOW_INFERENCE False; registered "BriVIS").

SANOnline per-frame outputs -> Hungarian linker over the per-frame query embeddings -> TemporalInstanceResampler ->
mask / attention-bias heads -> CLIP back blocks -> per-query logits averaged over time.

Frame sharding (BASELINE.json configs[3]): everything up to the per-frame query embeddings is
independent per frame, so ranks own contiguous frame blocks; ONE RCCL all-gather of the [t_local,Q,256] embeddings
precedes the linker + temporal resampler (replicated: deterministic, tiny), prediction heads and the CLIP back pass run
on the local frames only, and ONE all-reduce of the per-frame logit sums gives every rank the same classification."""
import numpy as np
import torch

from . import ops
from .config import resampler_precision
from . import distributed as D
from .modeling.minvis import batch_video_match_via_embeds
from .modeling.resampler import TemporalInstanceResampler
from .modeling.video_maskformer import retry_if_oom
from .registry import META_ARCH_REGISTRY
from .san import SANOnline


@META_ARCH_REGISTRY.register()
class BriVIS(SANOnline):
    def __init__(self, *, resampler=None, **kwargs):
        super().__init__(**kwargs)
        self.resampler = resampler or TemporalInstanceResampler(hidden_dim=256, feed_dim=2048, nheads=8, nlayers=6)

    @classmethod
    def from_config(cls, cfg):
        args = SANOnline.from_config(cfg)
        args["resampler"] = TemporalInstanceResampler(
            hidden_dim=256, feed_dim=2048, nheads=8, nlayers=6,                        # hard-coded, brivis.py:47
            precision=resampler_precision(cfg))
        return args

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.resampler.load_state_dict(sd, "resampler.", self.device)
        return self

    @retry_if_oom
    def forward(self, batched_inputs, stages=None, frame_range=None, gather_masks_to=None):
        """frame_range=(begin, end): this rank's contiguous frame block of the clip (frame-sharded mode; requires an
        initialised process group). Default: all frames on this rank.  gather_masks_to=r: the selected output masks of all
        frames are gathered on rank r (the other ranks return `pred_masks: []`); None: every rank keeps the masks of its
        own frames (`pred_masks_frames` names them) for a sharded evaluator."""
        dataset_name = list(set(x["dataset_name"] for x in batched_inputs))[0]
        class_names = self.get_class_name_list(dataset_name)
        self.sem_seg_head.num_classes = len(class_names)
        all_frames = [f for video in batched_inputs for f in video["image"]]
        T_total = len(all_frames)
        b0, b1 = frame_range if frame_range is not None else (0, T_total)
        frames = self._frames_to_device([{"image": all_frames[b0:b1]}])
        # the ONE exchange (C5) starts on a side stream as soon as the decoder has produced the local query embeddings; the
        # CLIP back pass of the local frames (per_window's post_encode_image) runs under it
        gather = []
        on_embeds = ((lambda e: gather.append(D.all_gather_frames_async(e[0], T_total) if e[0].shape[0] == b1 - b0 else None))
                     if frame_range is not None else None)
        io, images, image_size, padded = self.image_outputs(frames, class_names, on_embeds=on_embeds)   # brivis.py:149-171
        emb_local = io["pred_embeds"][0]                                                   # [t_local,Q,C]
        if frame_range is None:
            emb = emb_local
        else:                                          # (several windows: the hook saw partial blocks -> gather the whole block now)
            with D.span("all_gather_wait", host=True):     # what the compute stream / the host still waits for when it needs the embeddings
                emb = gather[0].wait() if len(gather) == 1 and gather[0] is not None else D.all_gather_frames(emb_local, T_total)
        with D.span("linker"):                                                             # replicated on every rank
            idx, frame_embeds = batch_video_match_via_embeds(emb.unsqueeze(0))             # brivis.py:173
        with D.span("temporal_resampler"):
            x = self.resampler.temporal(frame_embeds[0])                                   # [T,Q,C], replicated
        n = self.clip_adapter.num_heads
        pred_masks, biases, emb_out = self.resampler.prediction_heads(x[b0:b1], io["mask_feats"], io["attn_feats"], n)
        clip_feats = self.clip_adapter.post_encode_image(io["clip_tokens"], biases)        # resampler.py:313
        logits = self.clip_adapter.cal_sim_logits(io["text_feats"], clip_feats)            # [t_local,Q,K+1]
        probs = self.classify_sharded(logits, T_total, frame_range is not None)            # brivis.py:247-252
        if stages is not None:
            stages.update(dict(pred_masks=pred_masks.unsqueeze(0), pred_logits=logits.unsqueeze(0), indices=idx, probs=probs,
                               pred_embeds=emb_out))
        inp = batched_inputs[0]
        row_ids = np.arange(self.num_queries, dtype=np.int32)
        out = self.inference_video(self.num_queries, len(class_names), probs, row_ids, pred_masks, padded, image_size,
                                   inp.get("height", image_size[0]), inp.get("width", image_size[1]),
                                   redo=lambda: self.forward(batched_inputs, stages, frame_range, gather_masks_to),
                                   sync_guard=frame_range is not None,
                                   mask_gather=self.gather_masks_fn(T_total, gather_masks_to) if frame_range is not None else None)
        if frame_range is not None and gather_masks_to is None:
            out["pred_masks_frames"] = (b0, b1)
        return out

    __call__ = forward
