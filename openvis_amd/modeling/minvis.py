"""MinVIS — mirror of openvis/modeling/minvis.py:75-368 (eval path): frame decoder outputs + Hungarian tracker.

`post_processing` (minvis.py:320-338) runs the whole T-frame Hungarian chain in one kernel launch
(csrc/linker.hip) and applies the per-frame permutation with one gather kernel per tensor; the reference syncs to the
CPU and calls scipy once per frame (minvis.py:33-41)."""
import torch

from .. import ops
from ..registry import META_ARCH_REGISTRY
from .video_maskformer import VideoMaskFormer


def batch_video_match_via_embeds(pred_embeds):
    """pred_embeds [1,T,Q,C] -> (indices int32 [1,T,Q], permuted embeds [1,T,Q,C]) (minvis.py:44-72)."""
    assert pred_embeds.shape[0] == 1, "eval path: one video per call"
    _, T, Q, C = pred_embeds.shape
    emb = pred_embeds.view(T, Q, C).contiguous()
    idx = ops.hungarian_link(emb)
    out = torch.empty_like(emb)
    ops.batch_index_rows(emb, idx, out, Q * C, C, Q * C, C, C)
    return idx.view(1, T, Q), out.view(1, T, Q, C)


@META_ARCH_REGISTRY.register()
class MinVIS(VideoMaskFormer):
    def __init__(self, *, window_inference=False, window_size=10, **kwargs):
        super().__init__(**kwargs)
        self.window_inference, self.window_size = window_inference, window_size

    # time dimension of every per-frame output tensor (others, e.g. text features, are window independent)
    _TIME_DIM = {"pred_masks": 2, "pred_embeds": 1, "pred_logits": 1, "class_attn_biases": 1, "attn_feats": 0, "mask_feats": 0,
                 "clip_tokens": 0, "images": 0}

    def run_window_inference(self, fn, T):
        """minvis.py:340-362: run the per-frame part `fn(begin, end) -> dict` on windows of WINDOW_SIZE frames and
        concatenate the per-frame outputs along time.  Every tensor up to the tracker is per-frame, so the result equals
        the un-windowed one; the windows only bound the activation memory of long videos (the reference additionally parks
        the masks on the CPU, minvis.py:358 -- with 288 GB of HBM they stay on the device)."""
        windows = self.window_inference or getattr(self._fwd, "force_windows", False)     # (retry_if_oom: this host thread's forward only)
        if not windows or T <= self.window_size:
            return fn(0, T)
        parts = [fn(b, min(b + self.window_size, T)) for b in range(0, T, self.window_size)]
        out = dict(parts[0])
        for k, d in self._TIME_DIM.items():
            if k in out and torch.is_tensor(out[k]):
                out[k] = torch.cat([p[k] for p in parts], dim=d)
        return out

    @classmethod
    def from_config(cls, cfg):
        args = VideoMaskFormer.from_config(cfg)
        args["window_inference"] = cfg.MODEL.MASK_FORMER.TEST.WINDOW_INFERENCE
        args["window_size"] = cfg.MODEL.MASK_FORMER.TEST.WINDOW_SIZE
        return args

    def post_processing(self, outputs, shard=None):
        """Reorder per-frame logits and masks by the tracker's assignment (minvis.py:320-338).

        shard = (T_total, b0): `outputs` hold the frames [b0, b0 + t) of a clip whose frames are sharded over ranks.
        The tracker is a sequential chain over ALL frames: the per-frame query embeddings are all-gathered (distributed.all_gather_frames,
        [t,Q,256] f32 per rank), every rank runs the identical chain (one deterministic kernel: no broadcast) and applies the rows of its own
        frames.  out["indices"] is then the clip's [1, T_total, Q]."""
        if shard is None:
            idx, _ = batch_video_match_via_embeds(outputs["pred_embeds"])
            idx_all = idx
        else:
            from .. import distributed as D
            T_total, b0 = shard
            local = outputs["pred_embeds"][0].contiguous()                # [t,Q,C]
            with D.span("all_gather_wait", host=True):
                full = D.all_gather_frames(local, T_total)
            with D.span("linker"):
                idx_all, _ = batch_video_match_via_embeds(full.unsqueeze(0))
            idx = idx_all[:, b0:b0 + local.shape[0]].contiguous()
        _, T, Q = idx.shape
        idx2 = idx.view(T, Q)
        masks = outputs["pred_masks"][0]                                 # [Q,T,h,w]
        n = masks.shape[2] * masks.shape[3]
        out_masks = torch.empty_like(masks)
        # batch = frame t, rows = queries: element (t, q) of a [Q,T,...] tensor sits at q*T*n + t*n
        ops.batch_index_rows(masks, idx2, out_masks, n, T * n, n, T * n, n)
        out = dict(outputs)
        out["pred_masks"] = out_masks.unsqueeze(0)
        if "pred_logits" in outputs:
            lg = outputs["pred_logits"][0].contiguous()                  # [T,Q,K]
            K = lg.shape[-1]
            if K % 4 == 0:
                lo = torch.empty_like(lg)
                ops.batch_index_rows(lg, idx2, lo, Q * K, K, Q * K, K, K)
            else:                                                        # tiny / odd K (class-agnostic 2 logits): plain gather
                lo = torch.gather(lg, 1, idx2.long().unsqueeze(-1).expand(-1, -1, K))
            out["pred_logits"] = lo.unsqueeze(0)
        out["indices"] = idx_all
        return out
