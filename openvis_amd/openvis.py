"""OpenVIS meta-architecture — mirror of openvis/openvis.py:20-147 (eval path; registered as "OpenVIS").

forward(batched_inputs) takes the reference's input (a list with ONE video dict: "image": list of T uint8 [3,H,W]
tensors, "dataset_name", optional "height"/"width") and returns the reference's `video_output` dict
(video_maskformer.py:290-298).  Everything between runs on the gfx950 kernels; the [Q,T,Hp,Wp] upsampled mask
tensor of openvis.py:87-96 is never materialised (crops and the 10 output masks sample the low-res logits)."""
import numpy as np
import torch

from . import ops
from .catalog import MetadataCatalog
from .modeling.clip_adapter import build_clip_adapter
from .modeling.video_maskformer import VideoMaskFormer, retry_if_oom
from .registry import META_ARCH_REGISTRY


@META_ARCH_REGISTRY.register()
class OpenVIS(VideoMaskFormer):
    def __init__(self, *, clip_adapter, **kwargs):
        super().__init__(**kwargs)
        self.clip_adapter = clip_adapter
        self._all_rows = {}                       # (Q, device) -> arange(Q) int32: the row ids of the device crop-list path

    @classmethod
    def from_config(cls, cfg):
        args = VideoMaskFormer.from_config(cfg)
        assert cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES == 1          # openvis.py:35
        args["clip_adapter"] = build_clip_adapter(cfg.MODEL.CLIP_ADAPTER)
        return args

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.clip_adapter.load_state_dict(sd, "clip_adapter.", self.device)
        return self

    def get_class_name_list(self, dataset_name):
        return [c.strip() for c in MetadataCatalog.get(dataset_name).thing_classes]

    @retry_if_oom
    def forward(self, batched_inputs, stages=None, frame_range=None, gather_masks_to=None):
        """frame_range=(begin, end): this rank's contiguous frame block of the clip -- ONE clip over several GPUs (OpenVIS row:
        backbone, pixel decoder and CLIP crops per frame; the offline decoder's cross-attention as split-KV, one all-gather of flash partials
        per layer).  Needs an initialised process group whose ranks hold the blocks of `distributed.inference_shard`, every rank with at
        least one frame.  gather_masks_to=r: the selected masks of all frames end up on rank r (the others return `pred_masks: []`);
        None: every rank keeps the masks of its own frames (`pred_masks_frames`).  Default: the whole clip on this GPU."""
        dataset_name = list(set(x["dataset_name"] for x in batched_inputs))[0]
        class_names = self.get_class_name_list(dataset_name)
        self.sem_seg_head.num_classes = len(class_names)
        if frame_range is not None:
            return self._forward_split(batched_inputs, class_names, stages, frame_range, gather_masks_to)

        frames = self._frames_to_device(batched_inputs)                       # uint8 [T,3,H,W]
        images, image_size, padded = self.preprocess(frames)                  # A1
        features = self.backbone(images)                                      # A2
        outputs = self.sem_seg_head(features)                                 # A3-A8
        mask_score = outputs["pred_logits"][0]
        masks_lowres = outputs["pred_masks"][0]                               # [Q,T,h,w] logits

        probs, row_ids, extras = self.open_vocabulary_inference(mask_score, masks_lowres, frames, class_names, padded)
        extras = self._host_view_of_crops(extras) if stages is not None else extras
        if stages is not None:
            stages.update(dict(images=images, features=features, pred_masks=outputs["pred_masks"],
                               pred_logits=outputs["pred_logits"], probs=probs, row_ids=row_ids, **extras))
        inp = batched_inputs[0]
        height = inp.get("height", image_size[0])
        width = inp.get("width", image_size[1])
        dc = extras.get("device_crops")
        return self.inference_video(self.num_queries, len(class_names), probs, row_ids, masks_lowres, padded, image_size,
                                    height, width, redo=lambda: self.forward(batched_inputs, stages), n_valid=dc.counts if dc is not None else None)

    __call__ = forward

    def _forward_split(self, batched_inputs, class_names, stages, frame_range, gather_masks_to):
        from . import distributed as D
        all_frames = [f for video in batched_inputs for f in video["image"]]
        T_total = len(all_frames)
        b0, b1 = frame_range
        if not (0 <= b0 < b1 <= T_total):
            raise ValueError(f"frame_range {frame_range} of a {T_total}-frame clip: every rank of a split clip needs at least one frame")
        frames = self._frames_to_device([{"image": all_frames[b0:b1]}])
        images, image_size, padded = self.preprocess(frames)
        features = self.backbone(images)

        staged = D.backend_name() == "gloo"                 # test rigs stage through the host: wall time; RCCL: HIP events around the queued collective

        def exchange(part):
            with D.span("partial_all_gather", host=staged):
                return D.all_gather_rows(part)

        outputs = self.sem_seg_head(features, shard=(T_total, b0, exchange))
        masks_lowres = outputs["pred_masks"][0]                               # [Q, t_local, h, w]
        probs, row_ids, n_valid, extras = self._classify_split(masks_lowres, frames, class_names, padded, T_total)
        if stages is not None:
            stages.update(dict(images=images, features=features, pred_masks=outputs["pred_masks"], pred_logits=outputs["pred_logits"],
                               probs=probs, row_ids=row_ids, **extras))
        inp = batched_inputs[0]
        out = self.inference_video(self.num_queries, len(class_names), probs, row_ids, masks_lowres, padded, image_size,
                                   inp.get("height", image_size[0]), inp.get("width", image_size[1]),
                                   redo=lambda: self.forward(batched_inputs, stages, frame_range, gather_masks_to), sync_guard=True,
                                   n_valid=n_valid,
                                   mask_gather=self.gather_masks_fn(T_total, gather_masks_to))
        if gather_masks_to is None:
            out["pred_masks_frames"] = (b0, b1)
        return out

    def _classify_split(self, masks_lowres, frames, class_names, padded_hw, T_total):
        """openvis.py:110-147 with the clip's frames on several GPUs.  The crops of a frame need that frame only; the per-query mean over
        the frames with a non-empty mask (:130-138) needs all of them: every rank lays its crop logits out as [t_local, Q, K | valid] and
        ONE all-gather (distributed.all_gather_frames: ragged blocks padded) gives every rank the clip's [T, Q, K] -- the aggregate kernel
        then runs on exactly the rows, in exactly the order, of the one-GPU path, so every rank holds the same probabilities and picks the
        same top-10.  (T x Q x (K + 1) f32: 80 KB for 5 frames x 40 classes, 17 MB for 36 x 1196.)"""
        from . import distributed as D
        from .modeling.clip_adapter.adapter import DeviceCrops
        logits, valid, crops = self.clip_adapter(frames, class_names, masks_lowres, padded_hw)
        K = len(class_names)
        Q, t = masks_lowres.shape[0], masks_lowres.shape[1]
        dev = masks_lowres.device
        if logits is not None and logits.shape[1] != K:
            raise ValueError(f"{type(self.clip_adapter).__name__} returns {logits.shape[1]} logits for {K} classes; OpenVIS needs "
                             "ClipAdapter or AdaptedClipAdapter")
        packed = torch.zeros((t, Q, K + 1), dtype=torch.float32, device=dev)
        if isinstance(valid, DeviceCrops):
            ok = valid.slot >= 0                                              # [t, Q]; logit row of (t, q) = slot
            rows = logits[valid.slot.clamp(min=0).reshape(-1).long()].view(t, Q, K)
            packed[..., :K] = torch.where(ok.unsqueeze(-1), rows, torch.zeros((), device=dev))
            packed[..., K] = ok.float()
            extras = {"device_crops": valid, "crop_logits_all": logits}
        else:
            ok = ops.to_device_async(np.ascontiguousarray(valid.astype(np.uint8)), dev).bool()
            if logits is not None:
                full = packed[..., :K]
                full[ok] = logits                                             # (t, q) lexicographic = the adapter's row order
            packed[..., K] = ok.float()
            extras = {"valid": valid, "crops": crops, "crop_logits": logits}
        flag = self._forward_flag()
        with D.span("logit_all_gather", host=True):
            every = D.all_gather_frames(packed, T_total)                       # [T, Q, K + 1]
            if flag is not None:          # fp16x2: all ranks must agree on whether the clip is repeated under bf16x3 (inference_video, sync_guard)
                flag.copy_((D.all_reduce_sum(flag.to(torch.float32)) > 0).to(torch.int32))
        ok_all = every[..., K] > 0
        slot = torch.where(ok_all, torch.arange(T_total * Q, device=dev, dtype=torch.int32).view(T_total, Q),
                           torch.full((), -1, dtype=torch.int32, device=dev)).contiguous()
        probs, _ = ops.openvis_aggregate(every[..., :K].reshape(T_total * Q, K).contiguous(), slot, fill=-1.0)
        n_valid = ok_all.sum().to(torch.int32).reshape(1)
        rid = self._all_rows.get((Q, str(dev)))
        if rid is None:
            rid = self._all_rows[(Q, str(dev))] = torch.arange(Q, dtype=torch.int32, device=dev)
        return probs, rid, n_valid, extras

    @staticmethod
    def _host_view_of_crops(extras):
        """stage dumps (tests, bench): the device crop list in the host path's form -- valid [T,Q] numpy, crops / crop_logits of the valid
        (frame, query) pairs in (t, q) order.  Synchronises; the forward itself never calls it."""
        dc = extras.get("device_crops")
        if dc is None:
            return extras
        valid = dc.valid_host()
        keep = torch.from_numpy(valid.reshape(-1)).to(dc.crops.device)
        out = dict(extras)
        out.update(valid=valid, crops=dc.crops[keep].cpu().numpy(), crop_logits=extras["crop_logits_all"][keep])
        return out

    def open_vocabulary_inference(self, scores, masks_lowres, frames, class_names, padded_hw):
        """openvis.py:110-147. The reference walks the clip in chunks of 5 frames (part_len) only to bound memory; the
        per-(frame,query) crops are independent, so one batch over all T frames gives identical rows in the same
        (frame, query) order.  Returns (probs [Q,K] with the valid queries' rows filled, valid query ids, extras)."""
        if len(scores) == 0:
            return None, None, {}
        logits, valid, crops = self.clip_adapter(frames, class_names, masks_lowres, padded_hw)
        from .modeling.clip_adapter.adapter import DeviceCrops
        if isinstance(valid, DeviceCrops):
            # crop list built on the device (MODEL.CLIP_ADAPTER.CROP_LIST): one logit row per (frame, query), empty masks ignored through
            # slot = -1; EVERY query row takes part in the top-k, rows without a crop hold -1 and lose against any probability; nothing
            # here depends on the data, so the host never waits for the GPU (inference_video reads the count back with the outputs)
            dc = valid
            if logits.shape[1] != len(class_names):
                raise ValueError(f"{type(self.clip_adapter).__name__} returns {logits.shape[1]} logits for {len(class_names)} classes; OpenVIS needs "
                                 "ClipAdapter or AdaptedClipAdapter")
            probs, _ = ops.openvis_aggregate(logits, dc.slot, fill=-1.0)
            Q = dc.slot.shape[1]
            rid = self._all_rows.get((Q, str(probs.device)))
            if rid is None:
                rid = self._all_rows[(Q, str(probs.device))] = torch.arange(Q, dtype=torch.int32, device=probs.device)
            return probs, rid, {"device_crops": dc, "crop_logits_all": logits}
        if logits is None:                                                    # openvis.py:127-128
            return None, None, {"valid": valid}
        if logits.shape[1] != len(class_names):
            # Bg* adapters add a "non-object" column; the reference's inference_video builds its label table from
            # len(class_names) (openvis.py:108, video_maskformer.py:267-270) and mis-indexes it — only SimpleBaseline
            # (out of this path's scope) consumes those K + 1 logits
            raise ValueError(f"{type(self.clip_adapter).__name__} returns {logits.shape[1]} logits for {len(class_names)} "
                             "classes; OpenVIS needs ClipAdapter or AdaptedClipAdapter")
        T, Q = valid.shape
        slot = -np.ones((T, Q), np.int32)
        slot[valid] = np.arange(crops.shape[0], dtype=np.int32)
        probs, _ = ops.openvis_aggregate(logits, ops.to_device_async(slot, self.device))
        row_ids = np.nonzero(valid.any(axis=0))[0].astype(np.int32)
        return probs, row_ids, {"crop_logits": logits, "valid": valid, "crops": crops}


@META_ARCH_REGISTRY.register()
class OpenVISOnline(OpenVIS):
    """openvis/openvis.py:150-281: per-frame decoder + MinVIS tracker, then the same masked-crop CLIP classification.
    (The reference chunks CLIP crops by 10 frames instead of 5 only to bound memory, openvis.py:247.)"""

    def __init__(self, *, window=(False, 10), **kwargs):
        super().__init__(**kwargs)
        from .modeling.minvis import MinVIS
        self._post = MinVIS.post_processing
        self._windowed = MinVIS.run_window_inference
        self._TIME_DIM = MinVIS._TIME_DIM
        self.window_inference, self.window_size = window

    @classmethod
    def from_config(cls, cfg):
        args = OpenVIS.from_config(cfg)
        args["window"] = (cfg.MODEL.MASK_FORMER.TEST.WINDOW_INFERENCE, cfg.MODEL.MASK_FORMER.TEST.WINDOW_SIZE)
        return args

    @retry_if_oom
    def forward(self, batched_inputs, stages=None, frame_range=None, gather_masks_to=None):
        """frame_range / gather_masks_to: ONE clip's frames sharded over ranks: the per-frame decoder runs on the rank's own
        frames, the query embeddings are all-gathered in front of the replicated tracker (MinVIS.post_processing), the crop logits of all
        frames are all-gathered for the per-query mean (OpenVIS._classify_split)."""
        dataset_name = batched_inputs[0]["dataset_name"]
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
        images, image_size, padded = self.preprocess(frames)
        features = None

        def per_window(b0_, b1_):                                          # openvis.py:283-305 (run_window_inference)
            nonlocal features
            features = self.backbone(images[b0_:b1_])
            return self.sem_seg_head(features)

        raw = self._windowed(self, per_window, images.shape[0])
        outputs = self._post(self, raw, shard=(T_total, b0)) if sharded else self._post(self, raw)   # tracker (minvis.py:320-338)
        masks_lowres = outputs["pred_masks"][0]
        if sharded:
            probs, row_ids, n_valid, extras = self._classify_split(masks_lowres, frames, class_names, padded, T_total)
        else:
            probs, row_ids, extras = self.open_vocabulary_inference(outputs["pred_logits"][0], masks_lowres, frames,
                                                                    class_names, padded)
            dc = extras.get("device_crops")
            n_valid = dc.counts if dc is not None else None
        if stages is not None:
            extras = self._host_view_of_crops(extras)
            stages.update(dict(images=images, features=features, pred_masks=outputs["pred_masks"],
                               pred_embeds=outputs["pred_embeds"], indices=outputs["indices"], probs=probs,
                               row_ids=row_ids, **extras))
        inp = batched_inputs[0]
        out = self.inference_video(self.num_queries, len(class_names), probs, row_ids, masks_lowres, padded, image_size,
                                   inp.get("height", image_size[0]), inp.get("width", image_size[1]),
                                   redo=lambda: self.forward(batched_inputs, stages, frame_range, gather_masks_to), n_valid=n_valid,
                                   sync_guard=sharded,
                                   mask_gather=self.gather_masks_fn(T_total, gather_masks_to) if sharded else None)
        if sharded and gather_masks_to is None:
            out["pred_masks_frames"] = (b0, b1)
        return out

    __call__ = forward
