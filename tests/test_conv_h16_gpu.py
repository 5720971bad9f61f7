"""GPU: fp16-storage kernels of the ResNet backbone (round 4; csrc/conv_h16.hip, gemm_f16cvt.hip): 3x3 / 1x1 convolutions on fp16 NHWC maps
(LDS-DMA ring), the fp16-in / fp16-out forms of the 1x1 GEMM, the fp16 max pool and the stem writing fp16 -- against f64 convolutions of the
SAME fp16-rounded operands (the arithmetic contract: fp16 operands, f32 accumulation; detectron2 BottleneckBlock under autocast,
configs/openvoc_ytvis/Base.yaml:2-16, train_net.py:241) -- and the whole backbone with fp16 storage against the f32-storage path."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv(x16, w16, bias, stride, pad, residual=None, relu=True):
    y = F.conv2d(x16.double().permute(0, 3, 1, 2), w16.double().permute(0, 3, 1, 2), bias.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual.double()
    return y.relu() if relu else y


@pytest.mark.parametrize("T,H,W,Cin,Cout,k,stride", [(2, 46, 80, 64, 64, 3, 1), (1, 37, 53, 128, 128, 3, 1), (2, 46, 80, 128, 128, 3, 2), (1, 23, 40, 256, 256, 3, 1),
                                                      (1, 31, 45, 512, 512, 3, 2), (3, 184, 320, 64, 64, 3, 1), (2, 33, 47, 64, 256, 1, 1), (1, 23, 40, 512, 2048, 1, 1),
                                                      (1, 46, 80, 256, 512, 1, 2)])
@pytest.mark.parametrize("slots", [0, 3, 2])
def test_conv_h16_against_f64_on_the_same_fp16_operands(T, H, W, Cin, Cout, k, stride, slots):
    from openvis_amd import ops, _lib
    _lib.call("ovis_conv_h16_slots", slots)
    try:
        g = torch.Generator().manual_seed(H * W + Cin + k)
        x16 = torch.randn(T, H, W, Cin, generator=g).relu().half().cuda()
        w16 = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).half().cuda()
        b = torch.randn(Cout, generator=g).cuda()
        ref = _ref_conv(x16, w16, b, stride, k // 2)
        y16 = ops.conv_h16(x16, w16, k, stride, b, None, ops.ACT_RELU, out_f16=True)
        y32 = ops.conv_h16(x16, w16, k, stride, b, None, ops.ACT_RELU, out_f16=False)
        assert y16.dtype == torch.float16 and y32.dtype == torch.float32 and tuple(y32.shape) == tuple(ref.shape)
        scale = ref.abs().max().item()
        assert (y32.double() - ref).abs().max().item() < 2e-5 * scale + 1e-5          # f32 accumulation of exact fp16 products
        assert torch.equal(y16, y32.half())                                          # fp16 output = ONE rounding of the f32 result
        if k == 1 or stride == 1:
            r = torch.randn(tuple(ref.shape), generator=g).cuda()
            yr = ops.conv_h16(x16, w16, k, stride, b, r, ops.ACT_RELU, out_f16=False)
            assert (yr.double() - _ref_conv(x16, w16, b, stride, k // 2, r)).abs().max().item() < 2e-5 * scale + 1e-5
        again = ops.conv_h16(x16, w16, k, stride, b, None, ops.ACT_RELU, out_f16=True)
        assert torch.equal(again, y16)                                               # run-to-run identical
    finally:
        _lib.call("ovis_conv_h16_slots", 0)


def test_conv_h16_race_screen_under_memory_traffic():
    """counted vmcnt waits + LDS-DMA ring: repeat while a side stream streams 512 MB copies through HBM"""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(9)
    big_a = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    big_b = torch.empty_like(big_a)
    side = torch.cuda.Stream()
    for (T, H, W, Cin, Cout, k, s) in [(5, 184, 320, 64, 64, 3, 1), (5, 46, 80, 256, 256, 3, 1), (5, 23, 40, 512, 512, 3, 1), (5, 92, 160, 128, 512, 1, 1)]:
        x16 = torch.randn(T, H, W, Cin, generator=g).relu().half().cuda()
        w16 = (torch.randn(Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5).half().cuda()
        b = torch.randn(Cout, generator=g).cuda()
        ref = ops.conv_h16(x16, w16, k, s, b, None, ops.ACT_RELU, out_f16=True).clone()
        torch.cuda.synchronize()
        for it in range(30):
            if it % 2 == 0:
                with torch.cuda.stream(side):
                    big_b.copy_(big_a, non_blocking=True)
            assert torch.equal(ops.conv_h16(x16, w16, k, s, b, None, ops.ACT_RELU, out_f16=True), ref), (H, W, Cin, it)
        torch.cuda.synchronize()


def test_x16_gemm_forms_pool_and_stem():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(4)
    a = torch.randn(73600, 512, generator=g).relu().cuda()
    w16 = (torch.randn(128, 512, generator=g) / 512 ** 0.5).half().cuda()
    b = torch.randn(128, generator=g).cuda()
    # f32 A rounded while staged == fp16 A stored: bit-identical results; fp16 C == one rounding of the f32 C
    c32 = ops.gemm_nt_x16(a, w16, b, None, ops.ACT_RELU)
    c32h = ops.gemm_nt_x16(a.half(), w16, b, None, ops.ACT_RELU)
    assert torch.equal(c32, c32h)
    c16 = ops.gemm_nt_x16(a, w16, b, None, ops.ACT_RELU, out_f16=True)          # (bias added in the epilogue instead of as the accumulators'
    assert torch.equal(c16, ops.gemm_nt_x16(a.half(), w16, b, None, ops.ACT_RELU, out_f16=True))   # start value: f32 sums in another order)
    assert (c16.float() - c32).abs().max().item() <= 2.0 ** -10 * c32.abs().max().item()
    ref = (a.half().double() @ w16.double().T + b.double()).relu()
    assert (c32.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    r = torch.randn(73600, 128, generator=g).cuda()
    cr = ops.gemm_nt_x16(a.half(), w16, b, r, ops.ACT_RELU)
    assert (cr.double() - (a.half().double() @ w16.double().T + b.double() + r.double()).relu()).abs().max().item() < 2e-5 * ref.abs().max().item()
    # pool: max commutes with the fp16 rounding
    x = torch.randn(2, 61, 90, 64, generator=g).relu().cuda()
    assert torch.equal(ops.maxpool3x3s2(x.half()), ops.maxpool3x3s2(x).half())
    # stem writing fp16 == fp16 of the stem writing f32
    img = torch.randn(2, 96, 128, 4, generator=g).cuda()
    ws = (torch.randn(64, 7, 8, 4, generator=g) / 14).cuda()
    ws16 = ops.cast_f16(ws)
    bs = torch.randn(64, generator=g).cuda()
    y32 = ops.conv2d_nhwc(img, ws, 2, 3, bs, None, ops.ACT_RELU, w16=ws16)
    y16 = ops.conv2d_nhwc_o16(img, ws16, 2, 3, bs, ops.ACT_RELU)                # (bias in the epilogue instead of as the start value: another f32 order)
    assert y16.dtype == torch.float16 and (y16.float() - y32).abs().max().item() <= 2.0 ** -10 * y32.abs().max().item()


def test_backbone_fp16_storage_equals_f32_storage_up_to_summation_order():
    """A2 with fp16 storage of the intra-bottleneck tensors against the f32-storage path.  Given identical f32 inputs, a convolution's fp16
    operands are identical on both paths (the f32 path rounds the same values while staging); the new kernels sum in another f32 order, so a
    value that sits within 1e-7 of an fp16 rounding boundary can round the other way and move a few downstream values by up to an fp16 ulp:
    besides, the f32-storage kernels start their accumulators at bias (+ residual) where the new ones add them in the epilogue, and the 16-bit
    MFMA truncates small products against a large accumulator (csrc/gemm_f16_pp.hip, FH notes): ~1e-6 per convolution.  Each such difference
    flips an fp16 rounding now and then and the flips random-walk through 16 bottlenecks: measured mean 9e-6 (res2) .. 9e-5 (res4) of the
    feature scale, maximum 1e-3 -- both paths are the same fp16-operand arithmetic, and both hold the parity bars against the f32 oracle
    (tests/test_openvis_gpu.py, test_c1 / test_c2)."""
    import bench
    model, sd, _ = bench.build_model("cuda")
    bb = model.backbone
    assert bb.precision == "fp16" and bb.h16_storage
    frames = bench.synth_frames(2, 360, 640, 3, "cuda")
    images, _, _ = model.preprocess(frames)
    new = bb(images)
    bb.h16_storage = False
    try:
        old = bb(images)
    finally:
        bb.h16_storage = True
    for k in ("res2", "res3", "res4", "res5"):
        scale = old[k].abs().max().item()
        d, dm = (new[k] - old[k]).abs().max().item() / scale, (new[k] - old[k]).abs().mean().item() / scale
        print(f"A2 {k}: fp16 storage vs f32 storage, max rel diff {d:.2e}, mean {dm:.2e}")
        assert new[k].dtype == torch.float32 and new[k].shape == old[k].shape and d < 5e-3 and dm < 3e-4


def test_backbone_chunks_the_frame_axis_beyond_the_32_bit_offset_limit():
    """conv_h16 addresses its fp16 input with 32-bit byte offsets, so a whole video handed to the backbone in one piece (openvis.py:64: the
    reference runs the backbone on every frame of the video at once) runs as chunks of the frame axis: the chunked result equals the
    un-chunked one bit for bit (the convolutions are per frame), and the default limit leaves room below the kernel's 2^31 guard."""
    import bench
    from openvis_amd.modeling.backbone.resnet import ResNet
    model, _, _ = bench.build_model("cuda")
    bb = model.backbone
    frames = bench.synth_frames(5, 96, 160, 3, "cuda")
    images, _, _ = model.preprocess(frames)
    whole = bb(images)
    # the derived per-frame table against the shapes ops.conv_h16 really receives (res3.0.conv2 reads [T, H/4, W/4, 128],
    # twice res2's bytes -- sizing the chunk from res2 let 143-281 frame 720p videos through to the kernel's guard)
    from openvis_amd import ops
    seen = []
    real = ops.conv_h16
    try:
        ops.conv_h16 = lambda x, *a, **k: (seen.append(tuple(x.shape)), real(x, *a, **k))[1]
        bb(images)
    finally:
        ops.conv_h16 = real
    Hp, Wp = images.shape[1:3]
    table = bb.h16_conv_inputs(Hp, Wp)
    assert [(h, w, c) for _, h, w, c in table] == [s[1:] for s in seen], (table, seen)
    per_frame = bb.h16_bytes_per_frame(Hp, Wp)
    assert per_frame == max(s[1] * s[2] * s[3] * 2 for s in seen) == (Hp // 4) * (Wp // 4) * 128 * 2
    calls = []
    inner = bb._forward_h16
    try:
        bb.H16_BYTE_LIMIT = 2 * per_frame + bb.h16_guard_slack(Hp, Wp)   # two frames per chunk: 2 + 2 + 1
        bb._forward_h16 = lambda x: (calls.append(x.shape[0]), inner(x))[1]
        chunked = bb(images)
    finally:
        del bb.H16_BYTE_LIMIT, bb._forward_h16
    assert calls == [2, 2, 1]
    for k in whole:
        assert torch.equal(whole[k], chunked[k]), k
    # 720p: 15.1 MB per frame (res3.0.conv2) -> the default limit admits 142 frames per chunk, and EVERY conv_h16 call of such a chunk
    # satisfies the kernel's guard T H W Cin 2 + (2 W + 2) Cin 2 < 2^31 (csrc/conv_h16.hip: ovis_conv_h16)
    n = (ResNet.H16_BYTE_LIMIT - bb.h16_guard_slack(736, 1280)) // bb.h16_bytes_per_frame(736, 1280)
    assert n == 142
    for _, h, w, c in bb.h16_conv_inputs(736, 1280):
        assert n * h * w * c * 2 + (2 * w + 2) * c * 2 < (1 << 31)
    assert (n + 1) * 184 * 320 * 128 * 2 + (2 * 320 + 2) * 128 * 2 >= (1 << 31)      # and one more frame would not


@pytest.mark.gpu
def test_conv_h16_guard_accepts_the_largest_chunk_the_backbone_builds():
    """The 720p case the advisor could not run: res3.0.conv2 on a 142-frame chunk [142, 184, 320, 128] fp16 (2.14 GB in, 0.54 GB out)
    passes the kernel's guard and equals the same convolution run on two halves."""
    from openvis_amd import ops
    from openvis_amd.modeling.backbone.resnet import ResNet
    g = torch.Generator().manual_seed(5)
    n = 142
    x = (torch.randn(2, 184, 320, 128, generator=g).half().cuda()).repeat(n // 2, 1, 1, 1).contiguous()
    w = (torch.randn(128, 3, 3, 128, generator=g) / 34).half().cuda()
    b = torch.randn(128, generator=g).cuda()
    y = ops.conv_h16(x, w, 3, 2, b, None, ops.ACT_RELU, out_f16=True)
    y0 = ops.conv_h16(x[:2].contiguous(), w, 3, 2, b, None, ops.ACT_RELU, out_f16=True)
    assert torch.equal(y[:2], y0) and torch.equal(y[-2:], y0)
    with pytest.raises(Exception, match="too large"):
        ops.conv_h16(torch.cat([x, x[:1]]), w, 3, 2, b, None, ops.ACT_RELU, out_f16=True)


def test_two_source_gemms_against_f64_on_the_same_fp16_operands():
    """conv3 + projection shortcut of a bottleneck as one GEMM (ovis_gemm_nt_x16_2a: both sources fp16; ovis_conv1x1_pair_x16: the f32 block
    input read at stride 2 and rounded to fp16 while staged) against f64 on the same fp16-rounded operands: relu([a1 | a2] [w3 | ws]^T + b)."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(11)
    # res2.0: M = 3 x 23 x 41 pixels, K1 = K2 = 64, N = 256
    T, H, W, K1, K2, N = 3, 23, 41, 64, 64, 256
    a1 = torch.randn(T, H, W, K1, generator=g).relu().half().cuda(); a2 = torch.randn(T, H, W, K2, generator=g).relu().half().cuda()
    w = (torch.randn(N, K1 + K2, generator=g) / (K1 + K2) ** 0.5).half().cuda(); b = torch.randn(N, generator=g).cuda()
    ref = (torch.cat([a1, a2], -1).double() @ w.double().t() + b.double()).relu()
    y = ops.gemm_nt_x16_2a(a1, a2, w, b, ops.ACT_RELU)
    assert y.dtype == torch.float32 and (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    # res3.0 / res5.0 shapes: second source f32 [T, H, W, C2] at stride 2 (odd sizes: OH = (H - 1) // 2 + 1)
    for (T, H, W, K1, C2, N) in ((2, 37, 51, 128, 256, 512), (1, 23, 40, 512, 1024, 2048), (2, 16, 24, 64, 64, 256)):
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        a1 = torch.randn(T, OH, OW, K1, generator=g).relu().half().cuda(); x2 = torch.randn(T, H, W, C2, generator=g).relu().cuda()
        w = (torch.randn(N, K1 + C2, generator=g) / (K1 + C2) ** 0.5).half().cuda(); b = torch.randn(N, generator=g).cuda()
        x2s = x2[:, ::2, ::2].half()                                                    # the pixels the stride-2 1x1 shortcut reads, as the MFMA sees them
        ref = (torch.cat([a1, x2s], -1).double() @ w.double().t() + b.double()).relu()
        y = ops.conv1x1_pair_x16(a1, x2, 2, w, b, ops.ACT_RELU)
        assert y.shape == (T, OH, OW, N) and (y.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item(), (T, H, W)


def test_backbone_fused_shortcut_equals_separate_shortcut_up_to_summation_order():
    """The first block of every stage with conv3 and the projection shortcut as one GEMM against the two-GEMM form: the same fp16 operands, one
    f32 accumulation chain instead of two added in f32 -- differences of the f32 summation order only."""
    import bench
    model, sd, _ = bench.build_model("cuda")
    bb = model.backbone
    assert bb.fuse_shortcut and len(bb.pair) == 4
    frames = bench.synth_frames(2, 360, 640, 3, "cuda")
    images, _, _ = model.preprocess(frames)
    new = bb(images)
    bb.fuse_shortcut = False
    try:
        old = bb(images)
    finally:
        bb.fuse_shortcut = True
    for k in ("res2", "res3", "res4", "res5"):
        scale = old[k].abs().max().item()
        d, dm = (new[k] - old[k]).abs().max().item() / scale, (new[k] - old[k]).abs().mean().item() / scale
        print(f"A2 {k}: fused vs separate shortcut, max rel diff {d:.2e}, mean {dm:.2e}")
        assert d < 5e-3 and dm < 3e-4


@pytest.mark.parametrize("T,H,W", [(2, 96, 160), (1, 736, 1280), (3, 70, 90), (1, 33, 34)])
def test_stem_pool_kernel_equals_stem_conv_then_pool(T, H, W):
    """ovis_resnet_stem_pool_f16 (stem conv + ReLU + max pool in one launch) against the two-launch form (gemm_f16cvt ConvA -> fp16, fp16 pool) and
    against f64 on the same fp16-rounded operands; sizes that leave partial tiles and odd conv / pool extents."""
    from openvis_amd import ops
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(T, H, W, 4, generator=g)
    x[..., 3] = 0
    w = torch.randn(64, 7, 8, 4, generator=g) / 12
    w[:, :, 7] = 0
    w[..., 3] = 0
    b = torch.randn(64, generator=g)
    xd, w16, bd = x.cuda(), w.half().cuda(), b.cuda()
    y = ops.resnet_stem_pool(xd, w16, bd)
    two = ops.maxpool3x3s2(ops.conv2d_nhwc_o16(xd, w16, 2, 3, bd, ops.ACT_RELU))
    assert y.shape == two.shape and y.dtype == torch.float16
    ref = F.conv2d(x.half().double().permute(0, 3, 1, 2), w.half().double()[:, :, :7].permute(0, 3, 1, 2), b.double(), stride=2, padding=3).relu()
    ref = F.max_pool2d(ref, 3, 2, 1).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    assert (y.double().cpu() - ref).abs().max().item() < 2e-3 * scale          # fp16 rounding of the conv map (2^-11 relative) + f32 order
    assert (y.float() - two.float()).abs().max().item() < 2e-3 * scale
    same = (y == two).float().mean().item()
    print(f"stem + pool {T}x{H}x{W}: identical fp16 values {same:.5f}")
    assert same > 0.99
