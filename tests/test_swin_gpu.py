"""GPU: Swin backbone (openvis_amd/modeling/backbone/swin.py) and its data-movement kernels vs the reference golden
(tests/golden/swin.npz, produced by the reference's SwinTransformer) and vs the torch oracle on other geometries."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as TR
from tests.test_oracle_path import load_swin_case
from tests._synth import synth_weights, synth_inputs

pytestmark = pytest.mark.gpu


def _nhwc4(x):
    return torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 1)).contiguous().cuda()


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("fp16", 3e-2)])
def test_swin_matches_reference_golden(precision, tol):
    from openvis_amd.modeling.backbone.swin import SwinTransformer
    g, cfg, Wd, x = load_swin_case()
    m = SwinTransformer(4, cfg["embed_dim"], cfg["depths"], cfg["num_heads"], cfg["ws"], precision=precision).load_state_dict(Wd)
    out = m(_nhwc4(x))
    for k in ("res2", "res3", "res4", "res5"):
        got = out[k].permute(0, 3, 1, 2).cpu().numpy()
        assert got.shape == g[k].shape
        assert np.abs(got - g[k]).max() < tol, (k, np.abs(got - g[k]).max())


@pytest.mark.parametrize("B,H,W,C,ws,shift", [(2, 16, 26, 64, 5, 2), (1, 7, 9, 32, 4, 0), (3, 12, 24, 96, 12, 6), (1, 5, 5, 32, 7, 3)])
def test_window_partition_merge_mask_vs_torch(B, H, W, C, ws, shift):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, H, W, C, generator=g)
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    h = torch.nn.functional.pad(x, (0, 0, 0, Wp - W, 0, Hp - H))
    if shift:
        h = torch.roll(h, (-shift, -shift), (1, 2))
    ref = h.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    win = ops.swin_window_partition(x.cuda(), ws, shift)
    assert torch.equal(win.cpu(), ref)
    a = torch.randn(ref.shape, generator=g)
    r = a.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift:
        r = torch.roll(r, (shift, shift), (1, 2))
    out = ops.swin_window_merge_add(a.cuda(), x.cuda(), ws, shift)
    assert torch.equal(out.cpu(), x + r[:, :H, :W])
    if shift:
        ld = (ws * ws + 3) // 4 * 4
        m = ops.swin_shift_mask(H, W, ws, shift, ld, "cuda").cpu()
        refm = TR._swin_shift_mask(Hp, Wp, ws, shift) != 0
        assert torch.equal(m[:, :, :ws * ws].bool(), refm) and not m[:, :, ws * ws:].any()


def test_patch_merge_gather_and_relpos_bias():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 7, 13, 32, generator=g)
    h = torch.nn.functional.pad(x, (0, 0, 0, 1, 0, 1))
    ref = torch.cat([h[:, 0::2, 0::2], h[:, 1::2, 0::2], h[:, 0::2, 1::2], h[:, 1::2, 1::2]], -1)
    assert torch.equal(ops.swin_patch_merge_gather(x.cuda()).cpu(), ref)
    ws, heads = 5, 4
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    ref = table[TR._swin_rel_index(ws).view(-1)].view(ws * ws, ws * ws, heads).permute(2, 0, 1)
    b = ops.swin_relpos_bias(table.cuda(), heads, ws, 28).cpu()
    assert torch.equal(b[:, :, :25], ref) and not b[:, :, 25:].any()


def test_gelu_epilogue():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(4)
    a = torch.randn(300, 64, generator=g)
    w = torch.randn(96, 64, generator=g) / 8
    b = torch.randn(96, generator=g)
    ref = torch.nn.functional.gelu(a.double() @ w.double().T + b.double())
    out = ops.gemm_nt(a.cuda(), w.cuda(), b.cuda(), None, ops.ACT_GELU).cpu()
    assert (out.double() - ref).abs().max() < 2e-6


def test_swin_window12_720p_stage_shapes_vs_oracle():
    """Swin-L-like geometry (window 12, paddings 184->192, 320->324) at reduced depth/width vs the torch oracle."""
    from openvis_amd.modeling.backbone.swin import SwinTransformer
    depths, heads, E, ws = [1, 2, 1, 1], [1, 2, 4, 8], 32, 12
    spec = []
    spec += [("patch_embed.proj.weight", (E, 3, 4, 4)), ("patch_embed.proj.bias", (E,)), ("patch_embed.norm.weight", (E,)),
             ("patch_embed.norm.bias", (E,))]
    for i, d in enumerate(depths):
        C = E * 2 ** i
        for j in range(d):
            p = f"layers.{i}.blocks.{j}."
            spec += [(p + "norm1.weight", (C,)), (p + "norm1.bias", (C,)), (p + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, heads[i])),
                     (p + "attn.qkv.weight", (3 * C, C)), (p + "attn.qkv.bias", (3 * C,)), (p + "attn.proj.weight", (C, C)),
                     (p + "attn.proj.bias", (C,)), (p + "norm2.weight", (C,)), (p + "norm2.bias", (C,)), (p + "mlp.fc1.weight", (4 * C, C)),
                     (p + "mlp.fc1.bias", (4 * C,)), (p + "mlp.fc2.weight", (C, 4 * C)), (p + "mlp.fc2.bias", (C,))]
        if i < 3:
            spec += [(f"layers.{i}.downsample.reduction.weight", (2 * C, 4 * C)), (f"layers.{i}.downsample.norm.weight", (4 * C,)),
                     (f"layers.{i}.downsample.norm.bias", (4 * C,))]
        spec += [(f"norm{i}.weight", (C,)), (f"norm{i}.bias", (C,))]
    Wd = synth_weights(spec, 7, "backbone.")
    x = synth_inputs([(1, 3, 736, 1280)], 8)[0]
    with torch.no_grad():
        ref = TR.swin(x, Wd, E, depths, heads, ws)
    m = SwinTransformer(4, E, depths, heads, ws, precision="fp32").load_state_dict(Wd)
    out = m(_nhwc4(x))
    for k in ref:
        got = out[k].permute(0, 3, 1, 2).cpu()
        assert got.shape == ref[k].shape
        assert (got - ref[k]).abs().max() < 5e-5, (k, (got - ref[k]).abs().max())


def test_swin_absolute_position_embedding_matches_oracle():
    """MODEL.SWIN.APE (swin.py:567-578, 706-713): absolute_pos_embed [1,E,h0,w0] is interpolated bicubically to the patch grid
    and added after the patch embedding."""
    from openvis_amd import weights
    from openvis_amd.modeling.backbone.swin import SwinTransformer
    from oracle import torch_ref as TR
    E, depths, heads, ws = 32, (1, 1, 1, 1), (1, 2, 4, 8), 7
    spec = weights.swin_spec("backbone.", E, depths, heads, ws) + [("backbone.absolute_pos_embed", (1, E, 14, 14))]
    Wd = weights.random_init(spec, seed=9)
    x = synth_inputs([(2, 3, 96, 160)], 4)[0]
    with torch.no_grad():
        ref = TR.swin(x, Wd, E, depths, heads, ws)
        plain = TR.swin(x, {k: v for k, v in Wd.items() if k != "backbone.absolute_pos_embed"}, E, depths, heads, ws)
    m = SwinTransformer(4, E, depths, heads, ws, ape=True, precision="fp32").load_state_dict(Wd)
    out = m(_nhwc4(x))
    for k in ref:
        got = out[k].permute(0, 3, 1, 2).cpu()
        assert (got - ref[k]).abs().max() < 5e-5, (k, (got - ref[k]).abs().max())
    assert (ref["res2"] - plain["res2"]).abs().max() > 1e-2          # the embedding is not a no-op on this fixture


def test_san_online_with_swin_backbone_end_to_end():
    """SANOnline on a small Swin backbone (window 7, head_dim 32) + small side-adapter CLIP vs the oracle, exact-f32 policy."""
    from openvis_amd import config, weights
    from openvis_amd.catalog import MetadataCatalog
    from openvis_amd.modeling.clip_adapter.side_adapter import SideAdapter
    from tests.test_openvis_gpu import _frames, K
    from tests.test_san_gpu import SAN_E2E_ARCH

    Q = 100
    swin = dict(embed_dim=32, depths=(2, 1, 2, 1), num_heads=(1, 2, 4, 8), window=7)
    sd = weights.random_init(weights.san_spec(swin, SAN_E2E_ARCH, Q), seed=23)
    cfg = config.get_cfg()
    cfg.MODEL.META_ARCHITECTURE = "SANOnline"
    cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME = "SideAdapterFrameMultiScaleMaskedTransformerDecoder"
    cfg.MODEL.CLIP_ADAPTER.CLIP_NUM_HEADS = 4
    cfg.MODEL.BACKBONE.NAME = "D2SwinTransformer"
    cfg.MODEL.SWIN.EMBED_DIM, cfg.MODEL.SWIN.DEPTHS = swin["embed_dim"], list(swin["depths"])
    cfg.MODEL.SWIN.NUM_HEADS, cfg.MODEL.SWIN.WINDOW_SIZE = list(swin["num_heads"]), swin["window"]
    cfg.MODEL.PRECISION = "fp32"
    model = config.build_model(cfg)
    model.clip_adapter = SideAdapter("tiny", broken_idx=3, merge_ids=[1, 2, 3], num_queries=Q, arch=SAN_E2E_ARCH, precision="fp32")
    model.load_state_dict(sd)
    names = [f"class_{i}" for i in range(K)]
    MetadataCatalog.get("synthetic_val").set(thing_classes=names)
    gen = torch.Generator().manual_seed(1)
    base = torch.randn(1, 64, generator=gen)
    text = torch.nn.functional.normalize(base + 0.05 * torch.randn(K, 64, generator=gen), dim=-1)
    model.clip_adapter.set_text_features(names, text)
    frames = _frames(3)
    st, ref_st = {}, {}
    model([{"image": [f for f in frames], "dataset_name": "synthetic_val"}], stages=st)
    bb = lambda images, W: TR.swin(images, W, swin["embed_dim"], swin["depths"], swin["num_heads"], swin["window"])
    with torch.no_grad():
        TR.san_online_forward(frames, sd, text, stages=ref_st, broken_idx=3, merge_ids=(1, 2, 3), resolution=64, clip_heads=4,
                              num_queries=Q, backbone_fn=bb)
    ig, ir = st["indices"].cpu().numpy().reshape(ref_st["indices"].shape), ref_st["indices"].numpy()
    same = (ig == ir).all(axis=0)
    assert same.mean() > 0.9, same.mean()
    sel = torch.from_numpy(np.nonzero(same)[0])
    g, r = st["pred_masks"].cpu()[:, sel], ref_st["pred_masks"][:, sel]
    agree = ((g > 0) == (r > 0)).float().mean().item()
    assert agree > 0.999, agree
    lg, lr = st["pred_logits"].cpu()[:, :, sel], ref_st["pred_logits"][:, :, sel]
    assert (lg - lr).abs().max().item() < 2e-2, (lg - lr).abs().max().item()


@pytest.mark.parametrize("ws,heads,nW,B,masked", [(12, 6, 6, 2, True), (12, 3, 4, 1, False), (7, 2, 9, 2, True), (5, 4, 3, 1, True), (4, 1, 2, 3, False)])
def test_swin_window_attention_f16_kernel(ws, heads, nW, B, masked):
    from openvis_amd import ops
    N, C = ws * ws, heads * 32
    ld = (N + 3) // 4 * 4
    g = torch.Generator().manual_seed(ws * 100 + heads)
    qkv = torch.randn(B * nW, N, 3 * C, generator=g).half()
    bias = torch.zeros(heads, N, ld)
    bias[:, :, :N] = torch.randn(heads, N, N, generator=g)
    mask = torch.zeros(nW, N, ld, dtype=torch.uint8)
    if masked:
        ids = torch.randint(0, 3, (nW, N), generator=g)
        mask[:, :, :N] = (ids[:, :, None] != ids[:, None, :]).to(torch.uint8)
    q, k, v = [t.view(B * nW, N, heads, 32).permute(0, 2, 1, 3).double() for t in qkv.split(C, dim=-1)]
    att = q @ k.transpose(-1, -2) * 32 ** -0.5 + bias[None, :, :, :N].double()
    if masked:
        att = att + (mask[:, :, :N].double() * -100.0).repeat(B, 1, 1)[:, None]
    ref = (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * nW, N, C)
    out = ops.swin_window_attention_f16(qkv.cuda(), bias.cuda(), mask.cuda() if masked else None, heads).cpu().double()
    assert (out - ref).abs().max().item() < 6e-3, (out - ref).abs().max().item()
