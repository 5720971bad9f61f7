"""The hot stages as PyTorch dispatcher ops (`torch.ops.ovis_mi.*`, csrc/torch_ext/hot_ops.cpp; SURVEY.md 8(b) B1: "a torch extension ...
and TORCH_LIBRARY(ovis_mi, ...) ops").  CPU part: schemas and Meta kernels (shapes / dtypes on the meta device: what torch.compile and
fake tensors see).  GPU part: every op gives the result of the C-ABI entry point it forwards to, and the Meta kernel agrees with the
real output."""
import numpy as np
import pytest
import torch

OPS = ["ms_deform_attn_forward", "gemm_nt_f16", "msda_encoder_fused", "attention_f16", "mask_bbox", "clip_crop_patches", "hungarian_link",
       "topk_entropy", "gemm_nt_f16_ln", "gemm_nt_f16_res16_stats", "row_stats_f16", "row_stats_finalize"]


def _mi():
    import MultiScaleDeformableAttention  # noqa: F401  (registers the ops)
    return torch.ops.ovis_mi


def _args(dev):
    g = torch.Generator().manual_seed(0)
    sizes = [(4, 7), (8, 14), (15, 27)]
    shapes = torch.tensor(sizes)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    M, K, N = 300, 64, 40
    qkv = torch.randn(2 * 50, 3 * 128, generator=g).half()
    masks = torch.randn(6, 2, 12, 16, generator=g)
    crops = torch.tensor([[0, 0, 3, 2, 40, 30], [1, 4, 0, 0, 63, 47], [0, 5, 10, 12, 20, 44]], dtype=torch.int32)
    a = {
        "gemm_nt_f16": (torch.randn(M, K, generator=g).half(), torch.randn(N, K, generator=g).half(), torch.randn(N, generator=g), None, 2, True),
        "msda_encoder_fused": (torch.randn(2, S, 256, generator=g), torch.randn(2, S, 288, generator=g), shapes, lsi, 8, 3, 4),
        "attention_f16": (qkv, qkv[:, 128:], qkv[:, 256:], 2, 2, 50, 50, 64, 50 * 384, 384, 50 * 384, 384, 50 * 384, 384),
        "mask_bbox": (masks, 48, 64),
        "clip_crop_patches": ((torch.rand(2, 3, 45, 60, generator=g) * 255).to(torch.uint8), masks, crops, 48, 64, 32, 16,
                              [0.48, 0.46, 0.41], [0.27, 0.26, 0.28], False),
        "hungarian_link": (torch.randn(4, 20, 32, generator=g),),
        "topk_entropy": (torch.softmax(torch.randn(9, 30, generator=g), -1), torch.arange(9, dtype=torch.int32), 10),
    }
    mv = lambda x: x.to(dev) if torch.is_tensor(x) else x
    return {k: tuple(mv(x) for x in v) for k, v in a.items()}


def test_ops_are_registered_with_schemas_and_meta_kernels():
    mi = _mi()
    for name in OPS:
        assert hasattr(mi, name), name
    for name, args in _args("meta").items():
        if name == "attention_f16":                      # views of one buffer: build them on the meta device
            q = torch.empty(100, 384, dtype=torch.float16, device="meta")
            args = (q, q[:, 128:], q[:, 256:]) + args[3:]
        out = getattr(mi, name)(*args)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        assert all(o.device.type == "meta" for o in outs), name
    assert mi.gemm_nt_f16(*_args("meta")["gemm_nt_f16"]).dtype == torch.float16
    # the LayerNorm-folded CLIP GEMMs (shapes only: the real kernels take >= 256 tiles of 256 x 256, tests/test_gemm_gpu.py)
    h = lambda *sh: torch.empty(*sh, dtype=torch.float16, device="meta")
    f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device="meta")
    st = mi.row_stats_f16(h(1970, 768))
    assert tuple(st.shape) == (1970, 2) and st.dtype == torch.float32
    o = mi.gemm_nt_f16_ln(h(1970, 768), h(2304, 768), f(2304), f(2304), st, 0)
    assert tuple(o.shape) == (1970, 2304) and o.dtype == torch.float16
    o, part = mi.gemm_nt_f16_res16_stats(h(1970, 768), h(768, 768), f(768), h(1970, 768))
    assert tuple(o.shape) == (1970, 768) and o.dtype == torch.float16 and tuple(part.shape) == (1970, 12, 2) and part.dtype == torch.float32
    assert tuple(mi.row_stats_finalize(part, 768).shape) == (1970, 2)
    assert tuple(mi.mask_bbox(*_args("meta")["mask_bbox"]).shape) == (2, 6, 4)
    assert tuple(mi.clip_crop_patches(*_args("meta")["clip_crop_patches"]).shape) == (3 * 4, 768)
    with pytest.raises((RuntimeError, NotImplementedError)):   # no CPU kernels: the product path has no CPU fallback
        mi.hungarian_link(torch.randn(2, 4, 8))


@pytest.mark.gpu
def test_dispatcher_ops_forward_to_the_c_abi_and_match_their_meta_kernels():
    import ctypes
    from openvis_amd import _lib, ops
    mi = _mi()
    real, meta = _args("cuda"), _args("meta")
    q = torch.empty(100, 384, dtype=torch.float16, device="meta")
    meta["attention_f16"] = (q, q[:, 128:], q[:, 256:]) + meta["attention_f16"][3:]
    for name in real:
        out = getattr(mi, name)(*real[name])
        mo = getattr(mi, name)(*meta[name])
        outs, mos = (out, mo) if isinstance(out, (tuple, list)) else ((out,), (mo,))
        for o, m in zip(outs, mos):
            assert o.is_cuda and tuple(o.shape) == tuple(m.shape) and o.dtype == m.dtype, name
    # the same numbers as the C ABI called directly (ctypes): gemm, tracker, boxes
    a, w, b = real["gemm_nt_f16"][:3]
    ref = torch.empty((a.shape[0], w.shape[0]), dtype=torch.float16, device="cuda")
    _lib.call("ovis_gemm_nt_f16", a, ctypes.c_longlong(a.shape[1]), w, ctypes.c_longlong(w.shape[1]), ref, ctypes.c_longlong(w.shape[0]),
              a.shape[0], w.shape[0], a.shape[1], b, None, ctypes.c_longlong(w.shape[0]), 2, 1, _lib.stream_ptr())
    assert torch.equal(mi.gemm_nt_f16(*real["gemm_nt_f16"]), ref)
    emb = real["hungarian_link"][0]
    idx = mi.hungarian_link(emb).cpu().numpy()
    from oracle import torch_ref as TR
    assert np.array_equal(idx, TR.video_match_via_embeds(emb.cpu())[0].numpy())
    # ops.py goes through the dispatcher: one path
    assert torch.equal(ops.mask_bbox(real["mask_bbox"][0], 48, 64), mi.mask_bbox(*real["mask_bbox"]))
    assert torch.equal(ops.hungarian_link(emb), mi.hungarian_link(emb))
    x = real["gemm_nt_f16"]
    assert torch.equal(ops.gemm_nt_f16(x[0], x[1], x[2], None, 2, out_f16=True), ref)
    # a fake-tensor trace of a wrapper reaches the Meta kernel instead of launching
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode(allow_non_fake_inputs=False) as mode:
        fa, fw = mode.from_tensor(x[0]), mode.from_tensor(x[1])
        fo = mi.gemm_nt_f16(fa, fw, None, None, 0, False)
        assert tuple(fo.shape) == (x[0].shape[0], x[1].shape[0]) and fo.dtype == torch.float32
