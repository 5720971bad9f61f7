"""GPU: f32 MFMA GEMM / implicit-GEMM conv vs a plain torch fp64->fp32 reference of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(out, ref, tol=2e-5):
    err = (out.double().cpu() - ref.double()).abs().max().item()
    scale = ref.double().abs().max().item() + 1e-12
    assert err / scale < tol, (err, scale)


@pytest.mark.parametrize("M,N,K", [(100, 256, 256), (1, 8, 3), (257, 130, 77), (96600 // 5, 288, 256), (1000, 2048, 256),
                                   (300, 482, 512), (4097, 256, 1024)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_nt(M, N, K, act):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(M + N + K + act)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().T + b.double() + r.double()
    if act == 1:
        ref = ref.relu()
    elif act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    out = ops.gemm_nt(a.cuda(), w.cuda(), b.cuda(), r.cuda(), act)
    _close(out, ref)
    out2 = ops.gemm_nt(a.cuda(), w.cuda())
    _close(out2, a.double() @ w.double().T)


def test_gemm_integer_exact_layout():
    # asymmetric integer data: catches transposed / permuted fragment maps exactly
    from openvis_amd import ops
    M, N, K = 133, 97, 70
    a = (torch.arange(M * K).reshape(M, K) % 13 - 6).float()
    w = (torch.arange(N * K).reshape(N, K) % 7 - 3).float()
    out = ops.gemm_nt(a.cuda(), w.cuda()).cpu()
    assert torch.equal(out, a @ w.T)


@pytest.mark.parametrize("cfg", [
    dict(N=2, H=17, W=23, Cin=4, Cout=64, k=7, s=2, p=3),      # stem-like (3 channels padded to 4)
    dict(N=1, H=12, W=20, Cin=64, Cout=64, k=3, s=1, p=1),
    dict(N=2, H=12, W=20, Cin=128, Cout=128, k=3, s=2, p=1),
    dict(N=1, H=9, W=7, Cin=256, Cout=512, k=1, s=2, p=0),
    dict(N=1, H=46, W=80, Cin=256, Cout=256, k=3, s=1, p=1),
])
def test_conv2d_nhwc(cfg):
    from openvis_amd import ops
    g = torch.Generator().manual_seed(cfg["H"] * cfg["W"])
    x = torch.randn(cfg["N"], cfg["Cin"], cfg["H"], cfg["W"], generator=g)
    w = torch.randn(cfg["Cout"], cfg["Cin"], cfg["k"], cfg["k"], generator=g) / (cfg["Cin"] * cfg["k"] ** 2) ** 0.5
    b = torch.randn(cfg["Cout"], generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=cfg["s"], padding=cfg["p"])
    res = torch.randn(ref.shape, generator=g)
    ref = (ref + res.double()).relu()
    y = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(),
                        cfg["s"], cfg["p"], b.cuda(), res.permute(0, 2, 3, 1).contiguous().cuda(), 1)
    _close(y.permute(0, 3, 1, 2), ref)


def test_gemm_rejects_cpu_tensors():
    from openvis_amd import ops, _lib
    with pytest.raises(_lib.OvisError):
        ops.gemm_nt(torch.zeros(4, 4), torch.zeros(4, 4))


@pytest.fixture
def gemm_modes():
    from openvis_amd import ops
    yield ops
    ops.set_f32_gemm_mode(1)


@pytest.mark.parametrize("M,N,K", [(96600, 256, 256), (40001, 288, 260), (33000, 256, 1024)])
def test_large_gemm_bf16x3_split_is_f32_grade(gemm_modes, M, N, K):
    # large problems run as six bf16 MFMA products of the exact 3-way bf16 split: must be as accurate as the f32 MFMA kernel
    ops = gemm_modes
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))     # rows of very different scale
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ac, wc, bc = a.cuda(), w.cuda(), b.cuda()
    ref = (ac.double() @ wc.double().T + bc.double()).relu()
    row_scale = ac.double().abs() @ wc.double().abs().T + bc.double().abs()                  # condition-aware error measure
    errs = []
    for mode in (0, 1):
        ops.set_f32_gemm_mode(mode)
        out = ops.gemm_nt(ac, wc, bc, None, 1)
        errs.append(((out.double() - ref).abs() / row_scale).max().item())
    assert errs[0] < 2e-6 and errs[1] < 2e-6, errs
    assert errs[1] <= 2.0 * errs[0] + 1e-8, errs


def test_large_gemm_bf16x2_opt_in_carries_16_operand_bits(gemm_modes):
    # MODEL.F32_GEMM_SPLIT "bf16x2" (mode 2): three products of the two leading bf16 planes -- operands rounded to 16
    # significand bits (2^-17 relative each), products and sums in f32: error between the f32 grade and 1e-4 of fp16 operands
    ops = gemm_modes
    M, N, K = 96600, 256, 256
    g = torch.Generator().manual_seed(2)
    a = torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))
    w = torch.randn(N, K, generator=g) / K ** 0.5
    ac, wc = a.cuda(), w.cuda()
    ref = ac.double() @ wc.double().T
    row_scale = ac.double().abs() @ wc.double().abs().T
    errs = {}
    for mode in (1, 2):
        ops.set_f32_gemm_mode(mode)
        errs[mode] = ((ops.gemm_nt(ac, wc).double() - ref).abs() / row_scale).max().item()
    assert errs[1] < 2e-6 and 2e-6 < errs[2] < 2 * 2.0 ** -17, errs
    # integers below 2^16 are exact in two planes
    ai = (torch.arange(70000 * 72).reshape(70000, 72) % 251 - 125).float()
    wi = (torch.arange(130 * 72).reshape(130, 72) % 97 - 48).float()
    assert torch.equal(ops.gemm_nt(ai.cuda(), wi.cuda()).cpu(), (ai.double() @ wi.double().T).float())
    with pytest.raises(Exception):
        ops.set_f32_gemm_mode(4)


def test_large_gemm_bf16x3_integer_exact(gemm_modes):
    ops = gemm_modes
    M, N, K = 70000, 130, 72
    a = (torch.arange(M * K).reshape(M, K) % 251 - 125).float()
    w = (torch.arange(N * K).reshape(N, K) % 97 - 48).float()
    ops.set_f32_gemm_mode(1)
    out = ops.gemm_nt(a.cuda(), w.cuda())
    assert torch.equal(out.cpu(), (a.double() @ w.double().T).float())


def test_large_conv_bf16x3_matches_native(gemm_modes):
    ops = gemm_modes
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 120, 200, 64, generator=g).cuda()
    w = (torch.randn(256, 3, 3, 64, generator=g) / 24).cuda()
    b = torch.randn(256, generator=g).cuda()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).permute(0, 2, 3, 1)
    outs = []
    for mode in (0, 1):
        ops.set_f32_gemm_mode(mode)
        outs.append(ops.conv2d_nhwc(x, w, 1, 1, b, None, 0))
    e0 = (outs[0].double() - ref).abs().max().item()
    e1 = (outs[1].double() - ref).abs().max().item()
    assert e1 <= 2 * e0 + 1e-7 and e1 < 2e-5, (e0, e1)


def test_presplit_weight_planes_give_identical_results(gemm_modes):
    """cw=True (constant weights split once into bf16 planes) must equal the on-the-fly split bit for bit."""
    ops = gemm_modes
    g = torch.Generator().manual_seed(5)
    a = torch.randn(40000, 256, generator=g).cuda()
    w = (torch.randn(288, 256, generator=g) / 16).cuda()
    b = torch.randn(288, generator=g).cuda()
    r = torch.randn(40000, 288, generator=g).cuda()
    assert torch.equal(ops.gemm_nt(a, w, b, r, 1), ops.gemm_nt(a, w, b, r, 1, cw=True))
    p = ops.w3_of(w).float()
    assert torch.equal(p[0] + p[1] + p[2], w) and ops.w3_of(w) is ops.w3_of(w)
    x = torch.randn(3, 120, 200, 64, generator=g).cuda()
    cwt = (torch.randn(256, 3, 3, 64, generator=g) / 24).cuda()
    assert torch.equal(ops.conv2d_nhwc(x, cwt, 1, 1, b[:256].contiguous(), None, 1), ops.conv2d_nhwc(x, cwt, 1, 1, b[:256].contiguous(), None, 1, cw=True))
    small = torch.randn(100, 256, generator=g).cuda()                     # small problems keep using the f32 weights
    assert torch.equal(ops.gemm_nt(small, w, b), ops.gemm_nt(small, w, b, cw=True))


@pytest.mark.parametrize("N,K,act,res,planes", [(256, 256, 0, True, False), (520, 192, 1, False, False), (512, 128, 1, False, True)])
def test_gemm_on_presplit_bf16_planes_is_f32_grade(N, K, act, res, planes):
    """ovis_gemm_nt_bf16x3_planes (gemm_f16_pp.hip, X3 mode): operands given as their exact 3-way bf16 split, the six plane-pair
    products run as one K axis of 6 K on the ping-pong kernel; f32 output or the split of the result (chained linear layers)."""
    from openvis_amd import ops
    M = 66000 + 37
    assert ops.x3pp_eligible(M, N, K)
    g = torch.Generator().manual_seed(N + K)
    a = (torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, 1, generator=g))).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    a3, w3 = ops.split_planes(a), ops.split_planes(w)
    assert torch.equal(a3.float().sum(0), a)                                     # the split is exact
    out = ops.gemm_nt_planes(a3, w3, b, r, act, out_planes=planes)
    ref = a.double() @ w.double().T + b.double() + (r.double() if res else 0)
    scale = a.double().abs() @ w.double().abs().T + b.double().abs() + (r.double().abs() if res else 0)
    if act:
        ref = ref.relu()
    got = out.double().sum(0) if planes else out.double()
    assert ((got - ref).abs() / scale).max().item() < 1e-6


@pytest.mark.parametrize("M,N,K,act,bias,res", [(100, 256, 256, 0, True, False), (100, 2048, 256, 1, True, False), (100, 256, 2048, 0, True, True),
                                                (100, 768, 256, 0, True, False), (7, 41, 256, 0, True, True), (128, 483, 512, 2, False, False),
                                                (100, 1024, 4096, 0, True, True), (1, 256, 256, 0, False, False)])
def test_skinny_f32_gemm_of_the_decoders(M, N, K, act, bias, res):
    # gemm_f32_skinny.hip: M <= 128 rows, the K axis spread over the 8 wavefronts of a workgroup (and over workgroups + a fixed-order
    # fix-up for K > 512); exact f32 MFMA -> f32 accuracy, deterministic run to run
    from openvis_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    r = torch.randn(M, N, generator=g).cuda() if res else None
    ref = a.double() @ w.double().T
    if bias:
        ref = ref + b.double()
    if res:
        ref = ref + r.double()
    ref = {0: lambda x: x, 1: torch.relu, 2: lambda x: x * torch.sigmoid(1.702 * x)}[act](ref)
    lib = ops._lib.lib()
    # mode 1 (default): K <= 512 on 32-row workgroups (blockIdx.z = row tile), long K on the split-K form; mode 2: the 128-row form for every K
    for mode in (1, 2):
        lib.ovis_set_skinny_gemm(mode)
        try:
            outs = [ops.gemm_nt(a, w, b, r, act) for _ in range(3)]
        finally:
            lib.ovis_set_skinny_gemm(1)
        assert outs[0].shape == (M, N)
        assert (outs[0].double() - ref).abs().max().item() < 2e-5, mode
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    lib.ovis_set_skinny_gemm(0)                             # the general kernel on the same problem
    try:
        old = ops.gemm_nt(a, w, b, r, act)
    finally:
        lib.ovis_set_skinny_gemm(1)
    assert (outs[0] - old).abs().max().item() < 2e-5


@pytest.fixture
def pp_tile_rows():
    from openvis_amd import ops
    lib = ops._lib.lib()
    yield lib.ovis_pp_tile_rows
    lib.ovis_pp_tile_rows(0)


@pytest.mark.parametrize("M,tm", [(22000 + 37, 256), (22000 + 37, 192), (66000 + 5, 0), (700, 0)])
def test_fp16_gemm_on_the_fp16_residual_stream(pp_tile_rows, M, tm):
    pp_tile_rows(tm)                                        # 0 = the launch picks 256- or 192-row tiles
    # out-proj / c_proj of a CLIP block with the residual stream in fp16: C = fp16(f32(R) + bias + A B^T), one rounding at the end.
    # M = 22 037 rows is taken by the ping-pong kernel's fp16-residual instantiation, 700 rows by the fallback (f32-residual kernels + a cast)
    from openvis_amd import ops
    N, K = 768, 768
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).cuda()
    r = (3 * torch.randn(M, N, generator=g)).half().cuda()
    ref = (a.double() @ w.double().T + b.double() + r.double())
    out = ops.gemm_nt_f16(a, w, b, r)
    assert out.dtype == torch.float16 and out.shape == (M, N)
    err = (out.double() - ref).abs()
    assert (err <= 1e-3 + 2.0 ** -10 * ref.abs()).all(), err.max().item()       # fp16 rounding of the result (+ f32 accumulation noise)
    assert torch.equal(out, ops.gemm_nt_f16(a, w, b, r))


def test_layernorm_and_token_embedding_on_fp16_streams():
    from openvis_amd import ops
    g = torch.Generator().manual_seed(5)
    x = (2 * torch.randn(3001, 768, generator=g) + 0.5).half().cuda()
    gamma, beta = torch.randn(768, generator=g).cuda(), torch.randn(768, generator=g).cuda()
    ref = torch.nn.functional.layer_norm(x.double(), (768,), gamma.double(), beta.double(), 1e-5)
    y32 = ops.layernorm(x, gamma, beta)
    y16 = ops.layernorm(x, gamma, beta, out_f16=True)
    assert y32.dtype == torch.float32 and (y32.double() - ref).abs().max().item() < 2e-5
    assert y16.dtype == torch.float16 and (y16.double() - ref).abs().max().item() < 5e-3
    # class-token rows of an fp16 token tensor -> dense f32
    tok = x[:3000].view(100, 30, 768)
    assert torch.equal(ops.cast_f16_to_f32_rows(tok[:, 0, :]), tok[:, 0, :].float())
    # ln_pre(cat(cls, patches) + pos): fp16 patch embeddings in, fp16 tokens out == the f32 kernel on the same values, rounded
    M, L1, C = 7, 197, 768
    patch = torch.randn(M * (L1 - 1), C, generator=g).cuda()
    cls, pos = torch.randn(C, generator=g).cuda(), torch.randn(L1, C, generator=g).cuda()
    t32 = ops.vit_embed_ln(patch.half().float(), cls, pos, gamma, beta, M, L1)
    t16 = ops.vit_embed_ln(patch.half(), cls, pos, gamma, beta, M, L1)
    assert t16.dtype == torch.float16 and torch.equal(t16, t32.half())


@pytest.mark.parametrize("tm", [256, 192])
@pytest.mark.parametrize("N,K,act,res", [(1024, 256, 1, False), (256, 256, 0, True), (768, 192, 3, False), (512, 1024, 2, False), (256, 2048, 1, True)])
def test_bf16x2_on_the_ping_pong_schedule_matches_the_tiled_kernel(gemm_modes, pp_tile_rows, N, K, act, res, tm):
    pp_tile_rows(tm)
    # MODEL.F32_GEMM_SPLIT bf16x2 with constant weights: eligible shapes run on the ping-pong kernel's f32-A mode (f32 A tiles by LDS-DMA,
    # split into hi / lo bf16 in registers); same three products as gemm_f32x3_kernel<.., 2> -> same error class, every activation
    ops = gemm_modes
    M = 70000 + 13
    g = torch.Generator().manual_seed(N + K + act)
    a = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda() if res else None
    pre = a.double() @ w.double().T + b.double() + (r.double() if res else 0)
    ref = {0: lambda x: x, 1: torch.relu, 2: lambda x: x * torch.sigmoid(1.702 * x), 3: lambda x: torch.nn.functional.gelu(x)}[act](pre)
    scale = a.double().abs() @ w.double().abs().T + b.double().abs() + (r.double().abs() if res else 0)
    lib = ops._lib.lib()
    ops.set_f32_gemm_mode(2)
    lib.ovis_gemm_nt_f32_w3_kernel.restype = __import__("ctypes").c_char_p
    outs = {}
    for pp in (1, 0):
        lib.ovis_set_f32a_pp(pp)
        try:
            outs[pp] = [ops.gemm_nt(a, w, b, r, act, cw=True) for _ in range(2)]
        finally:
            lib.ovis_set_f32a_pp(1)
    for pp in (1, 0):
        err = ((outs[pp][0].double() - ref).abs() / scale).max().item()
        assert err < 2.0 ** -15, (pp, err)
    assert torch.equal(outs[1][0], outs[1][1])                       # deterministic
    assert ((outs[1][0] - outs[0][0]).abs().double() / scale).max().item() < 2.0 ** -15


@pytest.mark.parametrize("M,N,K", [(96600 + 5, 256, 64), (98500 - 3, 392, 160), (70001, 776, 96), (131072 + 9, 512, 32 * 9)])
def test_192_row_tiles_on_ragged_shapes(gemm_modes, pp_tile_rows, M, N, K):
    # edge tiles of the 192-row variant are shifted inside the matrix and masked; ragged M (not a multiple of 192 or 256), ragged N
    # (a shifted column tile), short K (two or three 32-float K steps)
    ops = gemm_modes
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).cuda()
    ref = a.double() @ w.double().T + b.double() + r.double()
    scale = a.double().abs() @ w.double().abs().T + b.double().abs() + r.double().abs()
    ops.set_f32_gemm_mode(2)
    import ctypes
    fn = ops._lib.lib().ovis_gemm_nt_f32_w3_kernel
    fn.restype = ctypes.c_char_p
    name = fn(ctypes.c_void_p(a.data_ptr()), ctypes.c_longlong(K), ctypes.c_void_p(ops.w3_of(w).data_ptr()), ctypes.c_longlong(K),
              ctypes.c_longlong(w.numel()), ctypes.c_void_p(a.data_ptr()), ctypes.c_longlong(N), M, N, K, ctypes.c_void_p(b.data_ptr()),
              ctypes.c_void_p(r.data_ptr()), ctypes.c_longlong(N), 0).decode()
    assert name.startswith("gemm_f16_pp_kernel<0,0,true,false,true"), name          # the shape IS taken by the f32-A kernel
    outs = {}
    for tm in (192, 256):
        pp_tile_rows(tm)
        outs[tm] = ops.gemm_nt(a, w, b, r, 0, cw=True)
        assert ((outs[tm].double() - ref).abs() / scale).max().item() < 2.0 ** -15, tm
    # the same products in the same order per output element: the two tile heights agree to the last bit
    assert torch.equal(outs[192], outs[256])


def _ln_problem(M, N, K, seed, big_mean):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * 1.5
    if big_mean:                                            # rows with a large mean and one outlier channel: the cancellation case of the fold
        x = x + torch.randn(M, 1, generator=g) * 4.0
        x[:, 7] += 40.0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    gamma = 1.0 + 0.3 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    return x.half().cuda(), w.cuda(), b.cuda(), gamma.cuda(), beta.cuda()


@pytest.mark.parametrize("big_mean", [False, True])
@pytest.mark.parametrize("M,N,K,act", [(8229, 2304, 768, 0), (7000, 3072, 768, 2), (70001, 1536, 768, 0), (20000, 3072, 1024, 2)])
def test_layernorm_folded_into_the_fp16_gemm(M, N, K, act, big_mean):
    """ln_1 -> in_proj / ln_2 -> c_fc (mask_adapted_clip/model.py:262-267) with the LayerNorm folded into the GEMM: from the raw fp16 rows,
    rstd (x Wg^T - mean s) + c.  Checked against f64 on the same fp16 rows, against the LayerNorm kernel + plain GEMM (which rounds the
    normalised rows to fp16 first: the fold must not be worse), and for run-to-run identity (the statistics travel by LDS-DMA)."""
    from openvis_amd import ops
    x, w, b, gamma, beta = _ln_problem(M, N, K, M + act, big_mean)
    assert ops.gemm_nt_f16_ln_eligible(M, N, K, act)
    wg, s, c = ops.fold_layernorm(w, b, gamma, beta)
    st = ops.row_stats_f16(x)
    outs = [ops.gemm_nt_f16_ln(x, wg, s, c, st, act) for _ in range(3)]
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    xd = x.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    assert (st[:, 0].double() - mu[:, 0]).abs().max().item() < 5e-6 and (st[:, 1].double() * torch.sqrt(var[:, 0] + 1e-5) - 1).abs().max().item() < 2e-6
    ref = ((xd - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()) @ w.double().T + b.double()
    if act == 2:
        ref = ref * torch.sigmoid(1.702 * ref)
    unfused = ops.gemm_nt_f16(ops.layernorm(x, gamma, beta, out_f16=True), ops.cast_f16(w), b, None, act, out_f16=True)
    e_fold, e_unf = (outs[0].double() - ref).abs(), (unfused.double() - ref).abs()
    assert (e_fold <= 2e-3 + 2.0 ** -9 * ref.abs()).all(), e_fold.max().item()
    assert e_fold.mean().item() <= 1.05 * e_unf.mean().item(), (e_fold.mean().item(), e_unf.mean().item())


def test_layernorm_fold_is_offered_only_where_the_ping_pong_kernel_runs():
    from openvis_amd import ops
    assert ops.gemm_nt_f16_ln_eligible(98500, 2304, 768) and ops.gemm_nt_f16_ln_eligible(98500, 3072, 768, ops.ACT_QUICKGELU)
    assert not ops.gemm_nt_f16_ln_eligible(98500, 4096, 1024, ops.ACT_QUICKGELU)      # c and s of 4096 columns do not fit the LDS bias region
    assert not ops.gemm_nt_f16_ln_eligible(700, 2304, 768)                            # fewer than 256 tiles: the small kernels
    assert not ops.gemm_nt_f16_ln_eligible(98500, 2304, 768, ops.ACT_RELU)            # instantiated: none, QuickGELU
    x, w, b, gamma, beta = _ln_problem(700, 2304, 768, 1, False)
    wg, s, c = ops.fold_layernorm(w, b, gamma, beta)
    with pytest.raises(ops._lib.OvisError if hasattr(ops._lib, "OvisError") else RuntimeError):
        ops._lib.call("ovis_gemm_nt_f16_ln", x, ops._ll(768), wg, ops._ll(768), torch.empty(700, 2304, dtype=torch.float16, device="cuda"), ops._ll(2304),
                      700, 2304, 768, c, s, ops.row_stats_f16(x), 0, ops._lib.stream_ptr())


@pytest.mark.parametrize("M,N,K,tm", [(22000 + 37, 768, 768, 256), (22000 + 37, 768, 3072, 192), (66000 + 5, 768, 768, 0), (30000, 1024, 4096, 0)])
def test_fp16_residual_gemm_also_emits_the_row_statistics(pp_tile_rows, M, N, K, tm):
    """out_proj / c_proj on the fp16 stream: the epilogue's partial (sum, sum of squares) per 64-column piece give the next LayerNorm's
    statistics without another pass over the stream; the output itself is bit-identical to the plain fp16-residual GEMM."""
    from openvis_amd import ops
    pp_tile_rows(tm)
    g = torch.Generator().manual_seed(M)
    a = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    r = (2 * torch.randn(M, N, generator=g) + 3 * torch.randn(M, 1, generator=g)).half().cuda()
    ref = ops.gemm_nt_f16(a, w, b, r)
    runs = [ops.gemm_nt_f16_res16_stats(a, w, b, r) for _ in range(3)]
    assert all(st is not None and torch.equal(o, ref) and torch.equal(st, runs[0][1]) for o, st in runs)
    two_pass = ops.row_stats_f16(ref)
    st = runs[0][1]
    assert (st[:, 0] - two_pass[:, 0]).abs().max().item() < 1e-5 and (st[:, 1] / two_pass[:, 1] - 1).abs().max().item() < 2e-5
    # a problem the ping-pong kernel does not take: plain result, no statistics (callers fall back to row_stats_f16)
    o2, s2 = ops.gemm_nt_f16_res16_stats(a[:700], w, b, r[:700].contiguous())
    assert s2 is None and torch.equal(o2, ops.gemm_nt_f16(a[:700], w, b, r[:700].contiguous()))


def test_folded_gemm_race_screen_under_memory_traffic():
    """The statistics of the LayerNorm-folded GEMM reach LDS by LDS-DMA behind counted vmcnt waits: a misplaced wait would show as rare
    run-to-run differences that depend on memory latency.  Repeat the launch while a side stream streams 512 MB copies through HBM."""
    from openvis_amd import ops
    big_a = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device="cuda")
    big_b = torch.empty_like(big_a)
    side = torch.cuda.Stream()
    for (M, N, K, act) in [(98500, 2304, 768, 0), (33333, 3072, 768, 2)]:
        x, w, b, gamma, beta = _ln_problem(M, N, K, 7, True)
        wg, s, c = ops.fold_layernorm(w, b, gamma, beta)
        st = ops.row_stats_f16(x)
        ref = ops.gemm_nt_f16_ln(x, wg, s, c, st, act).clone()
        torch.cuda.synchronize()
        for it in range(40):
            if it % 2 == 0:
                with torch.cuda.stream(side):
                    big_b.copy_(big_a, non_blocking=True)
            assert torch.equal(ops.gemm_nt_f16_ln(x, wg, s, c, st, act), ref), (M, N, K, it)
        torch.cuda.synchronize()
    # the LayerNorm epilogue of the bf16x2 GEMM (one more workgroup barrier inside the tile transition, row sums through LDS) and the
    # padded-map convolution (per-tile DMA row bases), same treatment
    ops.set_f32_gemm_mode(2)
    try:
        g = torch.Generator().manual_seed(3)
        a = torch.randn(96600, 256, generator=g).cuda()
        w = (torch.randn(256, 256, generator=g) / 16).cuda()
        b, gamma, beta = torch.randn(256, generator=g).cuda(), torch.rand(256, generator=g).cuda() + 0.5, torch.randn(256, generator=g).cuda()
        r = torch.randn(96600, 256, generator=g).cuda()
        xp = ops.groupnorm_nhwc(torch.randn(5, 92, 160, 256, generator=g).cuda(), gamma, beta, pad=True)
        wc = (torch.randn(256, 3, 3, 256, generator=g) / 48).cuda()
        ref_ln, ref_cv = ops.gemm_nt_layernorm(a, w, b, r, gamma, beta).clone(), ops.conv3x3_padded(xp, wc).clone()
        torch.cuda.synchronize()
        for it in range(30):
            if it % 2 == 0:
                with torch.cuda.stream(side):
                    big_b.copy_(big_a, non_blocking=True)
            assert torch.equal(ops.gemm_nt_layernorm(a, w, b, r, gamma, beta), ref_ln), it
            assert torch.equal(ops.conv3x3_padded(xp, wc), ref_cv), it
        torch.cuda.synchronize()
    finally:
        ops.set_f32_gemm_mode(1)


@pytest.mark.parametrize("M,K", [(96600, 256), (96600, 1024), (100000 + 3, 256), (695520, 512)])
def test_layernorm_in_the_epilogue_of_the_bf16x2_gemm(gemm_modes, M, K):
    """Post-norm of the pixel decoder's encoder layers (msdeformattn.py:139-146): LayerNorm(a W^T + b + residual) with the LayerNorm in the
    epilogue of the ping-pong f32-A kernel (N = 256: a tile holds whole rows) against the same GEMM followed by the LayerNorm kernel, and
    against f64; run-to-run identical (the row sums of the four wavefront columns meet in LDS across a barrier)."""
    from openvis_amd import ops
    gemm_modes.set_f32_gemm_mode(2)
    N = 256
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = (0.1 * torch.randn(N, generator=g)).cuda()
    r = (2 * torch.randn(M, N, generator=g) + torch.randn(M, 1, generator=g)).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(N, generator=g)).cuda(), (0.2 * torch.randn(N, generator=g)).cuda()
    w3 = ops.w3_of(w)
    ll = ops._ll
    assert ops._lib.lib().ovis_gemm_nt_f32_w3_ln_eligible(ops._lib._conv(a), ll(K), ops._lib._conv(w3), ll(K), ll(w.numel()), ops._lib._conv(r), ll(N),
                                                          M, N, K, ops._lib._conv(b), ops._lib._conv(r), ll(N)) == 1
    outs = [ops.gemm_nt_layernorm(a, w, b, r, gamma, beta) for _ in range(3)]
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    two = ops.layernorm(ops.gemm_nt(a, w, b, r, cw=True), gamma, beta)
    ref = torch.nn.functional.layer_norm(a.double() @ w.double().T + b.double() + r.double(), (N,), gamma.double(), beta.double(), 1e-5)
    e_fused, e_two = (outs[0].double() - ref).abs().max().item(), (two.double() - ref).abs().max().item()
    assert (outs[0] - two).abs().max().item() < 2e-5                    # same GEMM values, statistics in a different f32 order
    assert e_fused < 1.2 * e_two + 1e-5, (e_fused, e_two)
    # exact-f32 policy: no ping-pong kernel -> the two-kernel form, transparently
    gemm_modes.set_f32_gemm_mode(1)
    assert torch.equal(ops.gemm_nt_layernorm(a[:3000], w, b, r[:3000].contiguous(), gamma, beta),
                       ops.layernorm(ops.gemm_nt(a[:3000], w, b, r[:3000].contiguous(), cw=True), gamma, beta))


@pytest.mark.parametrize("T,H,W,Cin,Cout,act", [(5, 184, 320, 256, 256, 0), (2, 92, 160, 256, 256, 1), (3, 67, 131, 128, 256, 0), (36, 46, 80, 256, 256, 0), (7, 131, 167, 128, 512, 1)])
def test_conv3x3_on_a_zero_padded_map_matches_the_gathering_convolution(gemm_modes, T, H, W, Cin, Cout, act):
    """The FPN output convolutions (msdeformattn.py:287-296, 372) on the ping-pong kernel: GroupNorm writes its result zero-padded, the
    convolution walks the padded map as a dense GEMM (per-lane row bases, one tap offset per K step).  Against the im2col-loader
    convolution of the same bf16x2 policy (same products, different summation blocking) and against f64."""
    from openvis_amd import ops
    gemm_modes.set_f32_gemm_mode(2)
    g = torch.Generator().manual_seed(T * H + W)
    x = torch.randn(T, H, W, Cin, generator=g).cuda()
    gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).cuda(), (0.1 * torch.randn(Cin, generator=g)).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).cuda()
    dense = ops.groupnorm_nhwc(x, gamma, beta, relu=False)
    padded = ops.groupnorm_nhwc(x, gamma, beta, relu=False, pad=True)
    assert padded.shape == (T, H + 2, W + 2, Cin) and torch.equal(padded[:, 1:-1, 1:-1], dense)
    ring = padded.clone(); ring[:, 1:-1, 1:-1] = 0
    assert ring.abs().max().item() == 0.0
    ref_lib = ops.conv2d_nhwc(dense, w, 1, 1, act=act, cw=True)
    if not ops.conv3x3_padded_eligible(T, H, W, Cin, Cout, act):
        assert torch.equal(ops.conv3x3_padded(padded, w, act=act), ref_lib)         # falls back to the gathering convolution
        return
    outs = [ops.conv3x3_padded(padded, w, act=act) for _ in range(3)]
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    ref = torch.nn.functional.conv2d(dense.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
    if act == 1:
        ref = ref.relu()
    e_new, e_old = (outs[0].double() - ref).abs().max().item(), (ref_lib.double() - ref).abs().max().item()
    assert e_new < 2.0 * e_old + 1e-5, (e_new, e_old)
    assert (outs[0] - ref_lib).abs().max().item() < 5e-4 * ref.abs().max().item()
