"""VideoMaskFormer — mirror of openvis/modeling/video_maskformer.py:23-298 (eval path)."""
import numpy as np
import torch

from .. import ops
from ..registry import META_ARCH_REGISTRY, BACKBONE_REGISTRY, SEM_SEG_HEADS_REGISTRY


def retry_if_oom(forward):
    """The reference wraps its big eval stages in detectron2's `retry_if_cuda_oom` (openvis.py:108, video_maskformer.py:205 / 213,
    brivis.py:201 / 211): run; on a CUDA out-of-memory error empty the caching allocator and run again; then run with the inputs moved to
    the CPU.  Here the same ladder sits around the whole eval forward (the stages the reference protects -- the x4 mask upsample and the
    output masks -- are never materialised by this implementation, so what can run out is a whole-video activation set):
      1. run;  2. OutOfMemoryError -> synchronize, torch.cuda.empty_cache(), run again;
      3. still out of memory -> the per-frame (online) models run once more as WINDOWS of MODEL.MASK_FORMER.TEST.WINDOW_SIZE frames
         (minvis.py:340-362: same outputs, activation memory bounded by the window) -- there is no CPU path to fall back to, by design;
         the offline models (decoder joint over all frames) re-raise.
    Frame-sharded runs (every rank must take the same path through the collectives) re-raise at the FIRST out-of-memory error."""
    import functools
    import warnings

    @functools.wraps(forward)
    def wrapped(self, batched_inputs, *args, **kwargs):
        sharded = kwargs.get("frame_range") is not None or (len(args) >= 2 and args[1] is not None)     # BriVIS.forward(.., stages, frame_range)
        try:
            return forward(self, batched_inputs, *args, **kwargs)
        except torch.OutOfMemoryError:
            # frame-sharded: NO local retry at all.  The forward's `on_embeds` hook has launched the all-gather as soon as the decoder output
            # existed; a rank that runs out of memory behind it and repeats the forward would issue a second all-gather while its peers have
            # moved on to the logit all-reduce / mask gather -- mismatched collective sequences hang or exchange garbage instead of failing.
            if sharded:
                raise
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        try:
            return forward(self, batched_inputs, *args, **kwargs)
        except torch.OutOfMemoryError:
            if not hasattr(self, "window_inference") or self.window_inference:
                raise
        warnings.warn(f"{type(self).__name__}: out of device memory on a whole clip; repeating it as windows of {self.window_size} frames")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        self._fwd.force_windows = True          # per host thread (like the range flag): other threads' forwards on this model are not touched
        try:
            return forward(self, batched_inputs, *args, **kwargs)
        finally:
            self._fwd.force_windows = False
    return wrapped


class StickyFlag:
    """OR of the fp16x2 range flags of every forward since it was made, for callers that drop outputs unread.  `bitwise_or_` on ONE device
    word is a read-modify-write queued from each host thread's own stream: with several clips in flight (ClipPipeline, --streams > 1) two
    forwards can race and lose a bit.  So: one device word per host thread (each only ever touched from that thread's stream), OR-ed at
    readout after a device synchronize."""

    def __init__(self, device):
        import threading
        self.device, self._lock, self._words = torch.device(device), threading.Lock(), {}

    def or_(self, flag):
        import threading
        tid = threading.get_ident()
        w = self._words.get(tid)
        if w is None:
            w = torch.zeros((1,), dtype=torch.int32, device=self.device)
            with self._lock:
                self._words[tid] = w
        w.bitwise_or_(flag)

    def item(self):
        torch.cuda.synchronize(self.device)
        with self._lock:
            words = list(self._words.values())
        return int(any(int(w.item()) != 0 for w in words))


def build_backbone(cfg):
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg)


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME).from_config(cfg, input_shape)


@META_ARCH_REGISTRY.register()
class VideoMaskFormer:
    def __init__(self, *, backbone, sem_seg_head, num_queries, object_mask_threshold, overlap_threshold,
                 size_divisibility, pixel_mean, pixel_std, num_frames, device="cuda", **unused):
        self.backbone, self.sem_seg_head = backbone, sem_seg_head
        self.num_queries = num_queries
        self.overlap_threshold, self.object_mask_threshold = overlap_threshold, object_mask_threshold
        self.size_divisibility = size_divisibility if size_divisibility >= 0 else backbone.size_divisibility
        self.pixel_mean, self.pixel_std = tuple(pixel_mean), tuple(pixel_std)
        self.num_frames = num_frames
        self.device = torch.device(device)
        self.training = False

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        return dict(backbone=backbone, sem_seg_head=build_sem_seg_head(cfg, backbone.output_shape()),
                    num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
                    object_mask_threshold=cfg.MODEL.MASK_FORMER.TEST.OBJECT_MASK_THRESHOLD,
                    overlap_threshold=cfg.MODEL.MASK_FORMER.TEST.OVERLAP_THRESHOLD,
                    size_divisibility=cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY, pixel_mean=cfg.MODEL.PIXEL_MEAN,
                    pixel_std=cfg.MODEL.PIXEL_STD, num_frames=cfg.INPUT.SAMPLING_FRAME_NUM)

    def load_state_dict(self, sd):
        self.backbone.load_state_dict(sd, "backbone.", self.device)
        self.sem_seg_head.load_state_dict(sd, "sem_seg_head.", self.device)
        return self

    def eval(self):
        return self

    f32_gemm_mode = None      # MODEL.F32_GEMM_SPLIT resolved by config.build_model (None: leave the library's setting alone)
    _fwd = __import__("threading").local()     # .flag: the fp16x2 range flag of the forward this host thread is running (or None)

    def _forward_flag(self):
        return getattr(self._fwd, "flag", None)

    def _frames_to_device(self, batched_inputs):
        """list of T uint8 [3,H,W] -> one uint8 [T,3,H,W] device tensor (openvis.py:57-60).  Host frames are copied
        frame by frame straight into the device tensor (async from pinned memory, e.g. DataLoader(pin_memory=True)): no
        host-side torch.stack pass (a 13.8 MB memcpy per 720p clip) and no pageable staging copy.

        First call of every forward: also (re)applies this model's f32-GEMM split, a process-wide library setting, so that
        a model's results never depend on which model ran before it."""
        mode = self.f32_gemm_mode                 # read ONCE: _range_guard may change it from another thread's finish() meanwhile
        if mode is not None and ops.f32_gemm_mode() != mode:
            ops.set_f32_gemm_mode(mode)
        # fp16x2: this forward's range flag (read back with the outputs, _range_guard), kept per host thread next to the library's
        # per-thread split so that the END of this forward checks the flag its own kernels raise whatever the shared attribute says by then
        self._fwd.flag = ops.f16x2_begin(self.device) if mode == 3 else None
        # a clip that came through data.resize_and_preprocess() brings its A1 output along (batched_inputs[0]["images_nhwc4"],
        # f32 [T,Hp,Wp,4]); the NEXT preprocess() call of this host thread on frames of the same geometry returns it instead of launching A1
        self._fwd.pre = batched_inputs[0].get("images_nhwc4") if len(batched_inputs) == 1 else None
        frames = [f for video in batched_inputs for f in video["image"]]
        f0 = frames[0]
        if any(f.dtype != torch.uint8 for f in frames):
            raise TypeError("frames must be uint8 [3,H,W] tensors (ytvis_dataset_mapper.py:293-313)")
        if any(f.shape != f0.shape for f in frames):
            raise ValueError("all frames of a clip must have the same size (ImageList.from_tensors pads, openvis.py:62; "
                             "the dataset mapper resizes every frame of a video identically)")
        nb = f0.numel()
        one_buffer = all(f.is_contiguous() and f.untyped_storage().data_ptr() == f0.untyped_storage().data_ptr()
                         and f.storage_offset() == f0.storage_offset() + i * nb for i, f in enumerate(frames))
        if all(f.is_cuda for f in frames):
            if one_buffer and f0.device == self.device:       # a clip already resident in HBM as one [T,3,H,W] tensor: a view, no copy
                return torch.empty(0, dtype=torch.uint8, device=self.device).set_(f0.untyped_storage(), f0.storage_offset(), (len(frames),) + tuple(f0.shape))
            return torch.stack(frames).to(self.device).contiguous()
        x = torch.empty((len(frames),) + tuple(f0.shape), dtype=torch.uint8, device=self.device)
        if not f0.is_cuda and one_buffer:
            # the frames are consecutive slices of ONE host buffer (a collated / stacked clip): one copy instead of T -- every
            # async copy costs the host ~0.1 ms of launch latency during which the GPU has nothing queued (tools/trace_gaps.py)
            whole = torch.empty(0, dtype=torch.uint8).set_(f0.untyped_storage(), f0.storage_offset(), x.shape)
            x.copy_(whole, non_blocking=True)
            return x
        for i, f in enumerate(frames):
            x[i].copy_(f, non_blocking=True)
        return x

    def preprocess(self, frames_u8):
        T, _, H, W = frames_u8.shape
        d = self.size_divisibility
        Hp, Wp = ((H + d - 1) // d * d, (W + d - 1) // d * d) if d > 1 else (H, W)
        pre, self._fwd.pre = getattr(self._fwd, "pre", None), None
        if pre is not None and tuple(pre.shape) == (T, Hp, Wp, 4) and pre.dtype == torch.float32 and pre.device == frames_u8.device and pre.is_contiguous():
            return pre, (H, W), (Hp, Wp)                       # written by the resize's vertical pass (csrc/resize.hip), bit-identical to A1's
        return ops.preprocess_u8(frames_u8, Hp, Wp, self.pixel_mean, self.pixel_std), (H, W), (Hp, Wp)

    def _range_guard(self, flag_host, redo):
        """fp16x2: `flag_host` is this forward's range flag on the host (after its copy has completed).  Set = an activation of a
        constant-weight layer left the fp16 range (|a| >= 65 504 / a_scale) and the layer's output is NaN: this model switches to the
        f32-grade bf16x3 split for good and repeats the clip (redo() -> the repeated forward's output).  Returns None when all is well."""
        if flag_host is None or int(flag_host) == 0:
            return None
        import warnings
        warnings.warn("fp16x2: an activation left the fp16 range; this model now runs MODEL.F32_GEMM_SPLIT = bf16x3 and the clip is repeated")
        self.f32_gemm_mode = 1
        if redo is None:
            raise RuntimeError("fp16x2: an activation left the fp16 range and the forward cannot be repeated here: set MODEL.F32_GEMM_SPLIT to bf16x3")
        out = redo()
        return dict(out.items()) if hasattr(out, "items") else out

    @staticmethod
    def _check_selected_rows(scores, topk, K):
        """Device crop list: EVERY query row takes part in the top-k and rows without a crop hold -1.  A selected -1 means fewer than `topk`
        (valid query, class) pairs exist -- where the reference's `scores.flatten(0, 1).topk(10)` raises (video_maskformer.py:269) and the
        host crop-list path raises inside ovis_topk_entropy; never hand out the filler rows as detections."""
        if scores and min(scores) < 0:
            n = sum(1 for v in scores if v >= 0)
            raise RuntimeError(f"inference_video: top-{topk} over {n // max(K, 1)} valid queries x {K} classes: selected index k out of range "
                               "(fewer (query, class) pairs than topk; video_maskformer.py:269)")

    sticky_range_flag = None  # optional StickyFlag: OR of the fp16x2 range flags of every forward since it was set (never read by the model)
    output_rle = False        # MODEL.MASK_FORMER.TEST.OUTPUT_RLE (not a reference key): RLE hand-off instead of dense masks

    @staticmethod
    def gather_masks_fn(total_frames, dst):
        """`mask_gather` argument of inference_video for gather_masks_to=dst (None -> None: every rank keeps the masks of its own frames)."""
        if dst is None:
            return None
        from .. import distributed as D

        def gather(m):
            with D.span("mask_gather", host=True):
                return D.gather_frame_masks(m, total_frames, dst)
        return gather

    def inference_video(self, num_queries, num_classes, probs, row_ids, pred_masks_lowres, padded_hw, img_size,
                        output_height, output_width, topk=10, redo=None, sync_guard=False, n_valid=None, mask_gather=None):
        """video_maskformer.py:262-298.  probs [Q,K] (rows of valid queries filled), row_ids = valid query ids.
        redo: callable repeating this forward (fp16x2 only: used when the range flag came back set, _range_guard); sync_guard: read the
        flag back NOW instead of with the outputs (frame-sharded runs: every rank holds the all-reduced flag, distributed.reduce_flag, and
        all of them must repeat the clip at the same point of their collective sequence).  n_valid: device int32 [1], the number of
        non-empty masks when the crop list was built on the device (row_ids then names EVERY query): read back with the outputs; 0 means
        what `row_ids is None` means on the host path -- an empty result.  mask_gather: frame-sharded runs, callable(device masks
        [n,t_local,H,W]) -> the masks of ALL frames on the output rank, None on the others (`gather_masks_to`); an argument, not model state:
        forwards of one model may run on several host threads."""
        flag = self._forward_flag()
        if flag is not None and self.sticky_range_flag is not None:
            self.sticky_range_flag.or_(flag)                  # callers that drop outputs unread (bench.py's timed loop) still learn of an overflow
        if flag is not None and sync_guard:
            again = self._range_guard(flag.cpu()[0], redo)
            if again is not None:
                return again
            flag = None
        if row_ids is None or len(row_ids) == 0:
            again = self._range_guard(flag.cpu()[0] if flag is not None else None, redo)
            if again is not None:
                return again
            return {"image_size": (output_height, output_width), "pred_entropys": [], "pred_scores": [],
                    "pred_labels": [], "pred_masks": []}
        dev = probs.device
        K = probs.shape[1]
        rid = (row_ids.to(device=dev, dtype=torch.int32) if torch.is_tensor(row_ids)
               else ops.to_device_async(np.ascontiguousarray(np.asarray(row_ids, dtype=np.int32)), dev))      # no host block (ops.to_device_async)
        # the kernel also emits the query id of every selected row, so the mask kernels are launched without waiting for
        # the host to read the indices back (the reference syncs on .tolist() here, video_maskformer.py:267-272)
        idx, score, ent, sel_q = ops.topk_entropy(probs, rid, topk)          # raises if rows*K < topk (as torch.topk)
        Q, T, h, w = pred_masks_lowres.shape
        if self.output_rle and mask_gather is not None:
            raise ValueError("MODEL.MASK_FORMER.TEST.OUTPUT_RLE with gather_masks_to: the run-length hand-off covers this rank's frames only; "
                             "keep the masks sharded (gather_masks_to=None) and merge the per-rank RLEs in the evaluator")
        if self.output_rle:
            # hand the evaluator COCO RLE instead of dense masks -- the masks are produced column-major
            # and run-length encoded on the GPU, only the run lengths cross PCIe (ytvis_eval.py:258-301 does this per
            # mask on the host after a dense D2H copy)
            from .. import rle
            cm = ops.final_masks(pred_masks_lowres, sel_q, padded_hw[0], padded_hw[1], img_size[0], img_size[1],
                                 output_height, output_width, column_major=True)
            counts, n_runs = ops.rle_encode(cm.view(-1, output_height * output_width))
            # the guard FIRST (as finish() below): an fp16x2 overflow turns the pixel decoder's output into NaN, every mask reads as empty,
            # and "no valid mask" is exactly the state an overflow produces
            again = self._range_guard(flag.cpu()[0] if flag is not None else None, redo)
            if again is not None:
                return again
            if n_valid is not None and int(n_valid.cpu()[0]) == 0:
                return {"image_size": (output_height, output_width), "pred_entropys": [], "pred_scores": [], "pred_labels": [], "pred_masks_rle": [],
                        "pred_queries": []}
            labels = [i % K for i in idx.cpu().tolist()]
            scores = score.cpu().tolist()
            self._check_selected_rows(scores, topk, K)
            return {"image_size": (output_height, output_width), "pred_entropys": ent.cpu().tolist(),
                    "pred_scores": scores, "pred_labels": labels,
                    "pred_masks_rle": rle.encode_video_masks(counts, n_runs, topk, T, output_height, output_width),
                    "pred_queries": sel_q.cpu().tolist()}
        masks = ops.final_masks(pred_masks_lowres, sel_q, padded_hw[0], padded_hw[1], img_size[0], img_size[1],
                                output_height, output_width)
        if mask_gather is not None:
            masks = mask_gather(masks)
            if masks is None:                                                 # not the output rank
                again = self._range_guard(flag.cpu()[0] if flag is not None else None, redo)
                if again is not None:
                    return again
                if n_valid is not None and int(n_valid.cpu()[0]) == 0:        # the output rank returns the empty result (finish() below)
                    return {"image_size": (output_height, output_width), "pred_entropys": [], "pred_scores": [], "pred_labels": [],
                            "pred_masks": [], "pred_queries": []}
                scores = score.cpu().tolist()
                self._check_selected_rows(scores, topk, K)                    # every rank raises where the output rank raises
                return {"image_size": (output_height, output_width), "pred_entropys": ent.cpu().tolist(),
                        "pred_scores": scores, "pred_labels": [i % K for i in idx.cpu().tolist()],
                        "pred_masks": [], "pred_queries": sel_q.cpu().tolist()}
        # D2H of the 10 output masks and of the top-10 scalars (video_maskformer.py:267-283): pinned staging buffers from torch's
        # caching host allocator, async copies on the hand-off side stream behind the last kernel; uint8 {0,1} is re-viewed as
        # bool (no host-side conversion pass).  Nobody waits here: the VideoOutput does when a field is read (output.py).
        from ..output import VideoOutput, copy_stream
        cur, side = torch.cuda.current_stream(), copy_stream(masks.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            host = torch.empty(masks.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(masks, non_blocking=True)
            small = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (idx, score, ent, sel_q)]
            for h_, t in zip(small, (idx, score, ent, sel_q)):
                h_.copy_(t, non_blocking=True)
                t.record_stream(side)
            nv_host = None
            if n_valid is not None:
                nv_host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                nv_host.copy_(n_valid, non_blocking=True)
                n_valid.record_stream(side)
            flag_host = None
            if flag is not None:
                flag_host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
                flag_host.copy_(flag, non_blocking=True)
                flag.record_stream(side)
            masks.record_stream(side)
            done = torch.cuda.Event()
            done.record(side)

        def finish():
            again = self._range_guard(flag_host[0] if flag_host is not None else None, redo)
            if again is not None:
                return {k: v for k, v in again.items() if k != "image_size"}
            if nv_host is not None and int(nv_host[0]) == 0:                  # no mask had a positive pixel (openvis.py:127-128)
                return {"pred_entropys": [], "pred_scores": [], "pred_labels": [], "pred_masks": [], "pred_queries": []}
            i_, s_, e_, q_ = (h_.tolist() for h_ in small)
            self._check_selected_rows(s_, topk, K)
            return {"pred_entropys": e_, "pred_scores": s_, "pred_labels": [i % K for i in i_],      # video_maskformer.py:269-270
                    "pred_masks": [m for m in host.view(torch.bool)], "pred_queries": q_}
        return VideoOutput({"image_size": (output_height, output_width)}, done, finish)
