"""MODEL.F32_GEMM_SPLIT "fp16x2" (round 4): three fp16 MFMA products of the fp16 hi / lo split (11 + 11 significand bits per operand) for
the constant-weight f32 layers -- the f32 grade at the MFMA cost of bf16x2 (include/openvis_hip.h, csrc/gemm_f16_pp.hip FH, gemm_f32x3.h FH).
Checked against f64 next to the native f32 MFMA kernel (an fmaf chain) and the bf16x3 split; the range contract (|a| < 65 504 / 16, device
flag otherwise) and the model-level fall-back to bf16x3 are exercised explicitly.  Reference layers: msdeformattn.py:329 keeps them f32."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture
def modes():
    from openvis_amd import ops
    yield ops
    ops.set_f32_gemm_mode(1)


def _errs(ops, fn, ref, scale, modes_=(0, 1, 3)):
    out = {}
    for m in modes_:
        ops.set_f32_gemm_mode(m)
        if m == 3:
            flag = ops.f16x2_begin("cuda")
        o = fn()
        out[m] = ((o.double() - ref).abs() / scale).max().item()
        if m == 3:
            assert int(flag.item()) == 0
    return out


@pytest.mark.parametrize("M,N,K,act", [(96600, 256, 256, 1), (40001, 288, 260, 1), (33000, 256, 1024, 0), (96600, 1024, 256, 1), (70000, 768, 264, 0)])
def test_large_gemm_fp16x2_split_is_f32_grade(modes, M, N, K, act):
    """Rows of very different scale inside the range contract (row scale e^-4 .. e^4, |a| < 4 094): the condition-aware error of fp16x2 is
    that of the native f32 MFMA kernel and of bf16x3.  Shapes: ping-pong kernel (f32-A), its 192-row tiles, gemm_f32x3_kernel (N = 288,
    K = 260: not a multiple of 32), residual + ReLU."""
    ops = modes
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g) * torch.exp((1.5 * torch.randn(M, 1, generator=g)).clamp(-4, 4))
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ac, wc, bc, rc = a.cuda(), w.cuda(), b.cuda(), r.cuda()
    ref = ac.double() @ wc.double().T + bc.double() + rc.double()
    ref = ref.relu() if act else ref
    row_scale = ac.double().abs() @ wc.double().abs().T + bc.double().abs() + rc.double().abs()
    e = _errs(ops, lambda: ops.gemm_nt(ac, wc, bc, rc, act, cw=True), ref, row_scale)
    print(f"\n[fp16x2 grade] M={M} N={N} K={K}: native f32 {e[0]:.3e}  bf16x3 {e[1]:.3e}  fp16x2 {e[3]:.3e}")
    assert e[0] < 2e-6 and e[1] < 2e-6 and e[3] < 2e-6, e
    assert e[3] <= 2.0 * e[0] + 1e-8, e                        # the bar the bf16x3 split is held to (tests/test_gemm_gpu.py)


def test_fp16x2_carries_22_operand_bits_and_is_integer_exact(modes):
    ops = modes
    ops.set_f32_gemm_mode(3)
    flag = ops.f16x2_begin("cuda")
    # integers: |a| <= 2000 (x 16 = 32 000: hi + lo hold 22 bits), |w| <= 48 -- every product and sum exact
    ai = (torch.arange(70000 * 72).reshape(70000, 72) % 4001 - 2000).float()
    wi = (torch.arange(130 * 72).reshape(130, 72) % 97 - 48).float()
    assert torch.equal(ops.gemm_nt(ai.cuda(), wi.cuda(), cw=True).cpu(), (ai.double() @ wi.double().T).float())
    assert int(flag.item()) == 0
    # the weight planes: hi + lo reproduce w * scale to 2^-22 relative, scale puts max |w| in [2^14, 2^15)
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(256, 256, generator=g) * 0.03).cuda()
    h2, s = ops.h2_of(w)
    assert h2.dtype == torch.float16 and tuple(h2.shape) == (2, 256, 256) and ops.h2_of(w)[0] is h2
    top = w.abs().max().item() * s
    assert 2.0 ** 14 <= top < 2.0 ** 15 and math.frexp(s)[0] == 0.5
    rec = (h2[0].double() + h2[1].double()) / s
    big = w.abs() > 1e-4
    assert ((rec - w.double()).abs()[big] / w.double().abs()[big]).max().item() < 2.0 ** -21


def test_fp16x2_range_flag_and_bf16x3_rows_unaffected(modes):
    """An activation beyond 65 504 / a_scale turns its row into NaN and raises the device flag -- on the ping-pong kernel and on
    gemm_f32x3_kernel; rows inside the range are unaffected; a fresh flag starts at 0."""
    ops = modes
    ops.set_f32_gemm_mode(3)
    g = torch.Generator().manual_seed(3)
    for (M, N, K) in [(96600, 256, 256), (40001, 288, 264)]:
        a = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / 16).cuda()
        flag = ops.f16x2_begin("cuda")
        good = ops.gemm_nt(a, w, cw=True).clone()
        assert int(flag.item()) == 0 and torch.isfinite(good).all()
        a2 = a.clone()
        a2[777, 5] = 5000.0                                    # x 16 = 80 000 > 65 504
        flag = ops.f16x2_begin("cuda")
        bad = ops.gemm_nt(a2, w, cw=True)
        assert int(flag.item()) == 1
        assert not torch.isfinite(bad[777]).any()
        keep = torch.ones(M, dtype=torch.bool, device="cuda"); keep[777] = False
        assert torch.equal(bad[keep], good[keep])
        flag = ops.f16x2_begin("cuda")
        a2[777, 5] = 4000.0                                    # x 16 = 64 000: inside
        ok = ops.gemm_nt(a2, w, cw=True)
        assert int(flag.item()) == 0 and torch.isfinite(ok).all()


@pytest.mark.parametrize("M,K", [(96600, 256), (96600, 1024), (100003, 256)])
def test_layernorm_epilogue_under_fp16x2(modes, M, K):
    """msdeformattn.py:139-146 post-norms in the epilogue of the fp16x2 GEMM: against GEMM + LayerNorm kernel and f64; run-to-run identical."""
    ops = modes
    ops.set_f32_gemm_mode(3)
    flag = ops.f16x2_begin("cuda")
    N = 256
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = (0.1 * torch.randn(N, generator=g)).cuda()
    r = (2 * torch.randn(M, N, generator=g) + torch.randn(M, 1, generator=g)).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(N, generator=g)).cuda(), (0.2 * torch.randn(N, generator=g)).cuda()
    outs = [ops.gemm_nt_layernorm(a, w, b, r, gamma, beta) for _ in range(3)]
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    two = ops.layernorm(ops.gemm_nt(a, w, b, r, cw=True), gamma, beta)
    ref = F.layer_norm(a.double() @ w.double().T + b.double() + r.double(), (N,), gamma.double(), beta.double(), 1e-5)
    e_fused, e_two = (outs[0].double() - ref).abs().max().item(), (two.double() - ref).abs().max().item()
    ops.set_f32_gemm_mode(0)
    e_native = (ops.layernorm(ops.gemm_nt(a, w, b, r), gamma, beta).double() - ref).abs().max().item()
    print(f"\n[fp16x2 LNO] M={M} K={K}: fused {e_fused:.3e} two-kernel {e_two:.3e} native f32 {e_native:.3e}")
    assert int(flag.item()) == 0
    assert (outs[0] - two).abs().max().item() < 2e-5
    assert e_fused < 1.5 * e_native + 2e-6, (e_fused, e_native)


@pytest.mark.parametrize("T,H,W,Cin,Cout,act", [(5, 184, 320, 256, 256, 0), (2, 92, 160, 256, 256, 1), (7, 131, 167, 128, 512, 1)])
def test_conv3x3_padded_and_gathering_conv_under_fp16x2(modes, T, H, W, Cin, Cout, act):
    """FPN output convolutions (msdeformattn.py:287-296, 372): the padded-map walk and the im2col loader, both on fp16 products, against
    f64 next to the native f32 convolution."""
    ops = modes
    g = torch.Generator().manual_seed(T * H + W)
    x = torch.randn(T, H, W, Cin, generator=g).cuda()
    gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).cuda(), (0.1 * torch.randn(Cin, generator=g)).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).cuda()
    dense = ops.groupnorm_nhwc(x, gamma, beta, relu=False)
    padded = ops.groupnorm_nhwc(x, gamma, beta, relu=False, pad=True)
    ref = F.conv2d(dense.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    ref = ref.relu() if act else ref
    ops.set_f32_gemm_mode(0)
    e_native = (ops.conv2d_nhwc(dense, w, 1, 1, act=act).double() - ref).abs().max().item()
    ops.set_f32_gemm_mode(3)
    flag = ops.f16x2_begin("cuda")
    gather = ops.conv2d_nhwc(dense, w, 1, 1, act=act, cw=True)
    outs = [ops.conv3x3_padded(padded, w, act=act) for _ in range(3)]
    assert all(torch.equal(o, outs[0]) for o in outs[1:]) and int(flag.item()) == 0
    e_pad, e_gather = (outs[0].double() - ref).abs().max().item(), (gather.double() - ref).abs().max().item()
    print(f"\n[fp16x2 conv3x3] {T}x{H}x{W} {Cin}->{Cout}: padded {e_pad:.3e} gathering {e_gather:.3e} native f32 {e_native:.3e}")
    assert e_pad < 2.0 * e_native + 1e-6 and e_gather < 2.0 * e_native + 1e-6, (e_pad, e_gather, e_native)


def test_1x1_and_strided_convs_under_fp16x2(modes):
    """input_proj 1x1 convolutions (msdeformattn.py:227-235) and a stride-2 3x3 (f32-class backbone of SAN / BriVIS) on fp16 products."""
    ops = modes
    g = torch.Generator().manual_seed(21)
    # ... and the 64-channel layers of that backbone (64-column tiles): res2's 3x3, a 256 -> 64 1x1 with odd sizes, the stem's padded 7x8 kernel
    for (N, H, W, Cin, Cout, k, s, p) in [(5, 46, 80, 1024, 256, 1, 1, 0), (5, 92, 160, 128, 256, 3, 2, 1), (5, 184, 320, 64, 64, 3, 1, 1),
                                           (3, 91, 157, 256, 64, 1, 1, 0), (2, 93, 161, 64, 48, 3, 1, 1)]:
        x = torch.randn(N, H, W, Cin, generator=g).relu().cuda()
        w = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).cuda()
        b = torch.randn(Cout, generator=g).cuda()
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=s, padding=p).permute(0, 2, 3, 1)
        ops.set_f32_gemm_mode(0)
        e0 = (ops.conv2d_nhwc(x, w, s, p, b).double() - ref).abs().max().item()
        ops.set_f32_gemm_mode(3)
        flag = ops.f16x2_begin("cuda")
        e3 = (ops.conv2d_nhwc(x, w, s, p, b, cw=True).double() - ref).abs().max().item()
        assert int(flag.item()) == 0 and e3 < 2.0 * e0 + 1e-6, (e0, e3)


def test_small_and_activation_activation_gemms_stay_exact_under_fp16x2(modes):
    """Below 256 tiles of 128x128 the *_h2 entry points run the exact f32 kernels on the f32 weights (bit-identical to mode 0); GEMMs between
    two activations (cw=False) run bf16x3 (bit-identical to mode 1)."""
    ops = modes
    g = torch.Generator().manual_seed(8)
    a, w = torch.randn(100, 256, generator=g).cuda(), torch.randn(256, 256, generator=g).cuda()
    big_a, big_w = torch.randn(70000, 256, generator=g).cuda(), torch.randn(200, 256, generator=g).cuda()
    ops.set_f32_gemm_mode(0); small0 = ops.gemm_nt(a, w, cw=True)
    ops.set_f32_gemm_mode(1); big1 = ops.gemm_nt(big_a, big_w)
    ops.set_f32_gemm_mode(3); ops.f16x2_begin("cuda")
    assert torch.equal(ops.gemm_nt(a, w, cw=True), small0) and torch.equal(ops.gemm_nt(big_a, big_w), big1)


def test_model_falls_back_to_bf16x3_when_an_activation_leaves_the_fp16_range():
    """VideoMaskFormer._range_guard: the flag of a forward comes back with its outputs; set -> the model switches to bf16x3 and repeats the
    clip.  Forced here with a_scale = 2^14 (range |a| < 4): the pixel decoder's activations exceed it."""
    import warnings
    from openvis_amd import ops
    from bench import build_model, synth_frames
    try:
        model, _, _ = build_model("cuda", f32_split="fp16x2")
        frames = synth_frames(2, 360, 640, 5, "cuda")
        inp = [{"image": [f for f in frames], "dataset_name": "synthetic_burst_val"}]
        assert model.f32_gemm_mode == 3
        out = model(inp)
        out.wait()
        assert model.f32_gemm_mode == 3 and len(out["pred_masks"]) == 10           # in range: stays on fp16x2
        ref_model, _, _ = build_model("cuda", f32_split="bf16x3")
        ref = ref_model(inp)
        ops._MODE.a_scale = 2.0 ** 14
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            out2 = model(inp)
            out2.wait()
        assert model.f32_gemm_mode == 1 and any("fp16 range" in str(w.message) for w in wlist)
        assert out2["pred_labels"] == ref["pred_labels"] and all(torch.equal(a, b) for a, b in zip(out2["pred_masks"], ref["pred_masks"]))
    finally:
        ops._MODE.a_scale = 16.0
        ops.set_f32_gemm_mode(1)


def test_rle_output_with_device_crop_list_repeats_the_clip_on_overflow():
    """The OUTPUT_RLE branch of inference_video with the crop list built on the device: an fp16x2 overflow makes the pixel decoder's output
    NaN, every mask reads as empty and `n_valid == 0` -- the guard has to come BEFORE the "no valid mask -> empty result" return, or the clip
    would come back empty without a warning (round-4 review)."""
    import warnings
    from openvis_amd import ops
    from bench import build_model, synth_frames
    try:
        model, _, _ = build_model("cuda", f32_split="fp16x2", crop_list="device")
        ref_model, _, _ = build_model("cuda", f32_split="bf16x3", crop_list="device")
        for m in (model, ref_model):
            m.output_rle = True
        frames = synth_frames(2, 360, 640, 5, "cuda")
        inp = [{"image": [f for f in frames], "dataset_name": "synthetic_burst_val"}]
        ref = ref_model(inp)
        assert len(ref["pred_masks_rle"]) == 10
        ops._MODE.a_scale = 2.0 ** 14
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            out = model(inp)
        assert model.f32_gemm_mode == 1 and any("fp16 range" in str(w.message) for w in wlist)
        assert len(out["pred_masks_rle"]) == 10 and out["pred_labels"] == ref["pred_labels"] and out["pred_masks_rle"] == ref["pred_masks_rle"]
    finally:
        ops._MODE.a_scale = 16.0
        ops.set_f32_gemm_mode(1)


@pytest.mark.parametrize("T,S,N2", [(5, 19320, 288), (3, 1001, 288), (2, 4600, 256)])
def test_dual_gemm_indexing_is_exact_on_small_integers(modes, T, S, N2):
    """ovis_gemm_nt_f32_h2_dual (value_proj + offset / weight projection of an encoder layer as one launch): with small-integer operands
    every product and sum is exact in any arithmetic, so the two outputs must EQUAL a w[:256]^T + b and a w[256:]^T + b + r[m % S] bit for
    bit -- row-periodic term, column split, shifted edge tiles (M = T S is no multiple of 192 / 256, the last column tile starts at N - 256)."""
    ops = modes
    ops.set_f32_gemm_mode(3)
    flag = ops.f16x2_begin("cuda")
    g = torch.Generator().manual_seed(S)
    C = 256
    a = torch.randint(-8, 9, (T, S, C), generator=g).float().cuda()
    w = torch.randint(-4, 5, (C + N2, C), generator=g).float().cuda()
    b = torch.randint(-50, 51, (C + N2,), generator=g).float().cuda()
    r = torch.randint(-100, 101, (S, N2), generator=g).float().cuda()
    both = ops.gemm_nt_dual(a, w, b, r, C)
    assert both is not None, "the two-output kernel must take the encoder's shapes"
    c1, c2 = both
    ref = a.double().view(-1, C) @ w.double().t() + b.double()
    ref2 = ref[:, C:].view(T, S, N2) + r.double()
    assert torch.equal(c1.double().view(-1, C), ref[:, :C])
    assert torch.equal(c2.double(), ref2)
    assert int(flag.item()) == 0


def test_dual_gemm_matches_the_two_projections(modes):
    """... and on real-valued data it is the f32-grade approximation of the same two projections: (src + pos) Woa^T computed as
    src Woa^T + pos Woa^T (ms_deform_attn.py:98-104, msdeformattn.py:138)."""
    ops = modes
    g = torch.Generator().manual_seed(5)
    T, S, C, N2 = 5, 19320, 256, 288
    src = (torch.randn(T, S, C, generator=g) * torch.exp(torch.randn(T, S, 1, generator=g))).cuda()
    pos = torch.randn(S, C, generator=g).cuda()
    w = (torch.randn(C + N2, C, generator=g) / 16).cuda()
    b = torch.randn(C + N2, generator=g).cuda()
    ref_v = src.double() @ w[:C].double().t() + b[:C].double()
    ref_o = (src.double() + pos.double()) @ w[C:].double().t() + b[C:].double()
    sv = src.double().abs() @ w[:C].double().abs().t() + b[:C].double().abs()
    so = (src.double().abs() + pos.double().abs()) @ w[C:].double().abs().t() + b[C:].double().abs()
    ops.set_f32_gemm_mode(3)
    flag = ops.f16x2_begin("cuda")
    posw = ops.gemm_nt(pos, w[C:].contiguous(), None, cw=True)
    c1, c2 = ops.gemm_nt_dual(src, w, b, posw, C)
    e1 = ((c1.double() - ref_v).abs() / sv).max().item()
    e2 = ((c2.double() - ref_o).abs() / so).max().item()
    ops.set_f32_gemm_mode(0)                                       # the exact-f32 MFMA kernel on the same problem
    n1 = ((ops.gemm_nt(src, w[:C].contiguous(), b[:C].contiguous()).double() - ref_v).abs() / sv).max().item()
    n2 = ((ops.gemm_nt(ops.add_bcast(src, pos), w[C:].contiguous(), b[C:].contiguous()).double() - ref_o).abs() / so).max().item()
    print(f"dual GEMM, condition-aware error: value {e1:.2e} (native f32 {n1:.2e}), offsets / weights {e2:.2e} (native f32 {n2:.2e})")
    assert int(flag.item()) == 0
    assert e1 < max(2 * n1, 4e-7) and e2 < max(2 * n2, 4e-7)
