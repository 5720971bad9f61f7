"""SAN / SANOnline meta-architectures — mirror of openvis/san.py:23-283 (eval path; registered as "SAN", "SANOnline").

CLIP front blocks feed the pixel decoder, the side-adapter frame decoder predicts masks + per-head attention biases,
the CLIP back blocks classify every query with [SOS] tokens, the MinVIS tracker links queries over time."""
import numpy as np
import torch

from .config import side_adapter_precision
from . import ops
from . import distributed as D
from .catalog import MetadataCatalog
from .modeling.clip_adapter.side_adapter import SideAdapter
from .modeling.minvis import MinVIS
from .modeling.video_maskformer import VideoMaskFormer, retry_if_oom
from .registry import META_ARCH_REGISTRY


def _build_side_adapter(cfg):
    return SideAdapter(cfg.MODEL.CLIP_ADAPTER.CLIP_MODEL_NAME, broken_idx=cfg.MODEL.CLIP_ADAPTER.BROKEN_ID,
                       merge_ids=cfg.MODEL.CLIP_ADAPTER.MERGE_IDS, num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
                       precision=side_adapter_precision(cfg))


def _classify(pred_logits):
    """mean over frames, softmax, drop the background column (san.py:116,257; video_maskformer.py:218-219) -> probs [Q,K]."""
    lg = pred_logits[0]                                                                   # [T,Q,K+1]
    T, Q, K1 = lg.shape
    slot = torch.arange(T * Q, dtype=torch.int32, device=lg.device).view(T, Q)           # every (t,q) is valid
    probs, _ = ops.openvis_aggregate(lg.reshape(T * Q, K1).contiguous(), slot)
    return probs[:, :-1].contiguous()


@META_ARCH_REGISTRY.register()
class SAN(VideoMaskFormer):
    """openvis/san.py:23-144 (eval path; registered as "SAN"): the offline (clip-level) side-adapter model.  CLIP front
    blocks feed the pixel decoder, SideAdapterVideoMultiScaleMaskedTransformerDecoder predicts one set of query masks for
    the clip plus per-frame per-head attention biases, the CLIP back blocks classify every (frame, query) with [SOS]
    tokens and the logits are averaged over the frames."""

    def __init__(self, *, clip_adapter, **kwargs):
        super().__init__(**kwargs)
        self.clip_adapter = clip_adapter

    @classmethod
    def from_config(cls, cfg):
        args = VideoMaskFormer.from_config(cfg)
        args["clip_adapter"] = _build_side_adapter(cfg)
        return args

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.clip_adapter.load_state_dict(sd, "clip_adapter.", self.device)
        return self

    def get_class_name_list(self, dataset_name):
        return [c.strip() for c in MetadataCatalog.get(dataset_name).thing_classes]

    @retry_if_oom
    def forward(self, batched_inputs, stages=None):
        dataset_name = list(set(x["dataset_name"] for x in batched_inputs))[0]
        class_names = self.get_class_name_list(dataset_name)
        self.sem_seg_head.num_classes = len(class_names)
        frames = self._frames_to_device(batched_inputs)
        images, image_size, padded = self.preprocess(frames)
        mg_feats, clip_tokens = self.clip_adapter.front_encode_image(frames, padded)          # san.py:103
        text_feats = self.clip_adapter.encode_text(class_names)
        features = self.backbone(images)
        outputs = self.sem_seg_head(features, extra_feats=mg_feats)
        clip_feats = self.clip_adapter.post_encode_image(clip_tokens, outputs["class_attn_biases"][0])   # san.py:115
        pred_logits = self.clip_adapter.cal_sim_logits(text_feats, clip_feats).unsqueeze(0)               # [1,T,Q,K+1]
        probs = _classify(pred_logits)
        masks_lowres = outputs["pred_masks"][0]
        if stages is not None:
            stages.update(dict(images=images, pred_masks=outputs["pred_masks"], pred_logits=pred_logits, probs=probs,
                               class_attn_biases=outputs["class_attn_biases"]))
        inp = batched_inputs[0]
        row_ids = np.arange(self.num_queries, dtype=np.int32)
        return self.inference_video(self.num_queries, len(class_names), probs, row_ids, masks_lowres, padded, image_size,
                                    inp.get("height", image_size[0]), inp.get("width", image_size[1]),
                                    redo=lambda: self.forward(batched_inputs, stages))

    __call__ = forward


@META_ARCH_REGISTRY.register()
class SANOnline(MinVIS):
    def __init__(self, *, clip_adapter, **kwargs):
        super().__init__(**kwargs)
        self.clip_adapter = clip_adapter

    @classmethod
    def from_config(cls, cfg):
        args = MinVIS.from_config(cfg)
        args["clip_adapter"] = _build_side_adapter(cfg)
        return args

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.clip_adapter.load_state_dict(sd, "clip_adapter.", self.device)
        return self

    def get_class_name_list(self, dataset_name):
        return [c.strip() for c in MetadataCatalog.get(dataset_name).thing_classes]

    def image_outputs(self, frames, class_names, on_embeds=None):
        """frames uint8 [T,3,H,W] -> per-frame head outputs incl. pred_logits [1,T,Q,K+1] (san.py:211-231); long videos
        go through windows of MODEL.MASK_FORMER.TEST.WINDOW_SIZE frames (san.py:285-307)."""
        T, _, H, W = frames.shape
        d = self.size_divisibility
        padded = ((H + d - 1) // d * d, (W + d - 1) // d * d) if d > 1 else (H, W)
        text_feats = self.clip_adapter.encode_text(class_names)                               # san.py:222

        def per_window(b0, b1):
            fr = frames[b0:b1]
            images, _, _ = self.preprocess(fr)
            mg_feats, clip_tokens = self.clip_adapter.front_encode_image(fr, padded)          # san.py:221
            outputs = self.sem_seg_head(self.backbone(images), extra_feats=mg_feats)
            if on_embeds is not None:                   # frame-sharded BriVIS: the query all-gather starts here (side stream)
                on_embeds(outputs["pred_embeds"])
            clip_feats = self.clip_adapter.post_encode_image(clip_tokens, outputs["class_attn_biases"][0])   # san.py:230
            outputs["pred_logits"] = self.clip_adapter.cal_sim_logits(text_feats, clip_feats).unsqueeze(0)   # [1,t,Q,K+1]
            outputs["clip_tokens"], outputs["images"] = clip_tokens, images
            return outputs

        outputs = self.run_window_inference(per_window, T)
        outputs["text_feats"] = text_feats
        return outputs, outputs.pop("images"), (H, W), padded

    def classify(self, pred_logits):
        """mean over frames, softmax, drop the background column (san.py:257,264-265) -> probs [Q,K]."""
        return _classify(pred_logits)

    @retry_if_oom
    def forward(self, batched_inputs, stages=None, frame_range=None, gather_masks_to=None):
        """frame_range=(begin, end): this rank's contiguous frame block of the clip (frame-sharded mode; needs an initialised
        process group whose ranks hold the blocks of `distributed.inference_shard`): every tensor up to the query embeddings is per-frame; ONE
        all-gather of the embeddings in front of the replicated tracker, ONE all-reduce of the per-frame logit sums for the temporal mean
        (san.py:257).  gather_masks_to=r: the selected masks of all frames end up on rank r; None: every rank keeps its own frames' masks."""
        dataset_name = list(set(x["dataset_name"] for x in batched_inputs))[0]
        class_names = self.get_class_name_list(dataset_name)
        self.sem_seg_head.num_classes = len(class_names)
        sharded = frame_range is not None
        if sharded:
            all_frames = [f for video in batched_inputs for f in video["image"]]
            T_total = len(all_frames)
            b0, b1 = frame_range
            if not (0 <= b0 < b1 <= T_total):
                raise ValueError(f"frame_range {frame_range} of a {T_total}-frame clip: every rank needs at least one frame")
            frames = self._frames_to_device([{"image": all_frames[b0:b1]}])
        else:
            frames = self._frames_to_device(batched_inputs)
        outputs, images, image_size, padded = self.image_outputs(frames, class_names)
        outputs = self.post_processing(outputs, shard=(T_total, b0) if sharded else None)     # tracker (minvis.py:320-338)
        probs = self.classify_sharded(outputs["pred_logits"][0], T_total, True) if sharded else self.classify(outputs["pred_logits"])
        masks_lowres = outputs["pred_masks"][0]
        if stages is not None:
            stages.update(dict(images=images, pred_masks=outputs["pred_masks"], pred_logits=outputs["pred_logits"],
                               indices=outputs["indices"], probs=probs, class_attn_biases=outputs["class_attn_biases"],
                               pred_embeds=outputs["pred_embeds"]))
        inp = batched_inputs[0]
        row_ids = np.arange(self.num_queries, dtype=np.int32)
        out = self.inference_video(self.num_queries, len(class_names), probs, row_ids, masks_lowres, padded, image_size,
                                   inp.get("height", image_size[0]), inp.get("width", image_size[1]),
                                   redo=lambda: self.forward(batched_inputs, stages, frame_range, gather_masks_to), sync_guard=sharded,
                                   mask_gather=self.gather_masks_fn(T_total, gather_masks_to) if sharded else None)
        if sharded and gather_masks_to is None:
            out["pred_masks_frames"] = (b0, b1)
        return out

    __call__ = forward

    def classify_sharded(self, logits, T_total, sharded):
        """softmax(mean over ALL frames of the logits)[:, :-1]; sharded: per-rank frame sums are all-reduced."""
        if not sharded:
            return self.classify(logits.unsqueeze(0))
        t, Q, K1 = logits.shape
        # mean over the local frames (kernel) weighted by the shard's share of the clip -> all-reduce = mean over ALL frames
        # -> softmax through the aggregate kernel.  The weighting is one elementwise scale of a [Q,K+1] tensor.
        local = ops.mean_over_dim0(logits.contiguous()) * (float(t) / float(T_total))
        flag = self._forward_flag()
        with D.span("logit_all_reduce", host=True):
            if flag is not None:
                # fp16x2: every constant-weight GEMM of this forward is queued by now; the range flags of the ranks ride on this all-reduce
                # (one more element), so that all ranks agree on whether the clip has to be repeated under bf16x3 (inference_video, sync_guard)
                packed = D.all_reduce_sum(torch.cat([local.reshape(-1), flag.to(torch.float32)]))
                total = packed[:-1].view_as(local)
                flag.copy_((packed[-1:] > 0).to(torch.int32))
            else:
                total = D.all_reduce_sum(local)
        one = torch.arange(Q, dtype=torch.int32, device=logits.device).view(1, Q)
        probs, _ = ops.openvis_aggregate(total.contiguous(), one)
        return probs[:, :-1].contiguous()
