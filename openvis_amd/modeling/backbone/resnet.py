"""ResNet backbone (detectron2 `build_resnet_backbone`, configs/openvoc_ytvis_coco/Base.yaml:2-16: DEPTH 50,
STRIDE_IN_1X1 False, FrozenBN, OUT_FEATURES res2..res5) on the gfx950 kernels.

NHWC activations; every conv is the f32 MFMA implicit GEMM (csrc/gemm_f32.hip) with FrozenBN folded into the
weights/bias at load time and ReLU / residual-add fused in the epilogue.  State-dict keys are detectron2's
(`stem.conv1.{weight,norm.*}`, `res{2..5}.{i}.conv{1,2,3}.*`, `shortcut.*`)."""
import torch

from ...config import backbone_precision as _backbone_precision

from ... import ops
from ...registry import BACKBONE_REGISTRY

STAGES = {50: (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2), ("res5", 3, 2))}


def _fold(sd, p, eps=1e-5, pad_cin_to=None):
    """conv weight [Cout,Cin,KH,KW] + FrozenBN -> (w' [Cout,KH,KW,Cin] with the BN scale folded in, bias)."""
    w = sd[p + ".weight"].float()
    scale = sd[p + ".norm.weight"].float() * (sd[p + ".norm.running_var"].float() + eps).rsqrt()
    bias = sd[p + ".norm.bias"].float() - sd[p + ".norm.running_mean"].float() * scale
    w = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)
    if pad_cin_to is not None and w.shape[-1] < pad_cin_to:
        w = torch.nn.functional.pad(w, (0, pad_cin_to - w.shape[-1]))
    return w.contiguous(), bias.contiguous()


class ResNet:
    size_divisibility = 32

    def __init__(self, depth=50, out_features=("res2", "res3", "res4", "res5"), precision="fp16"):
        self.depth = depth
        self.out_features = tuple(out_features)
        # "fp16": conv operands rounded to fp16, f32 accumulation (= the reference under autocast, train_net.py:241);
        # "fp32": exact-f32 MFMA.  Activations are f32 in HBM either way.
        self.precision = precision
        # fp16 policy only: the tensors BETWEEN the convolutions of a bottleneck (conv1 -> conv2 -> conv3) and stem -> pool -> res2.0 are
        # stored in fp16.  Their only readers are convolutions that round them to fp16 while staging, so every MFMA operand is bit-identical
        # to the f32-storage path; the block outputs (the residual stream, and what the pixel decoder reads) stay f32.  The 3x3 convolutions
        # then run on the LDS-DMA kernel of csrc/conv_h16.hip.  False = f32 storage everywhere (rounds 1-3).
        self.h16_storage = True
        self.fuse_stem = True          # (h16_storage) stem + max pool as one kernel (csrc/conv_h16.hip: stem_pool_kernel)
        self.fuse_shortcut = True      # (h16_storage) conv3 + projection shortcut of a stage's first block as one GEMM (load_state_dict)
        self.w = {}
        self.w16 = {}

    def output_shape(self):
        ch = {"res2": 256, "res3": 512, "res4": 1024, "res5": 2048}
        st = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        return {k: dict(channels=ch[k], stride=st[k]) for k in self.out_features}

    def load_state_dict(self, sd, prefix="backbone.", device="cuda"):
        d = lambda t: t.to(device)
        # stem 7x7: input channels padded 3 -> 4 (the NHWC4 frames), and -- fp16 policy -- an eighth, all-zero kernel COLUMN: K = 7*8*4 = 224
        # is a multiple of 8, which puts the stem on the fp16-operand convolution like every other backbone conv (K = 196 fell through to
        # the f32-class kernel: 0.53 ms per 720p clip).  Same output geometry (stride 2, pad 3: (W + 6 - 8) / 2 + 1 == (W + 6 - 7) / 2 + 1
        # for even W), the extra tap multiplies by zero.  (f32-class policy: measured, no gain from the pre-split weight planes -- unpadded.)
        sw, sb = _fold(sd, prefix + "stem.conv1", pad_cin_to=4)
        if self.precision == "fp16":
            sw = torch.nn.functional.pad(sw, (0, 0, 0, 1)).contiguous()              # [64, 7, 7, 4] -> [64, 7, 8, 4]
        self.w["stem"] = (d(sw), d(sb))
        # (f32-class policy, SAN / BriVIS: the stem stays on the native-f32 convolution, 4.2 ms per 36-frame 720p clip.  The padded 7x8 kernel
        # under the fp16x2 split runs 1.6 ms and is f32-grade, but its different rounding in the FIRST layer moved 3 mask bits of C3 outside
        # the |logit| < 1e-3 set the parity statement allows (52 instead of 30 differing bits of 29.4 M; round 5): not taken.)
        for name, nblocks, _ in STAGES[self.depth]:
            for i in range(nblocks):
                p = f"{prefix}{name}.{i}"
                for c in ("conv1", "conv2", "conv3", "shortcut"):
                    if f"{p}.{c}.weight" in sd:
                        self.w[f"{name}.{i}.{c}"] = tuple(map(d, _fold(sd, f"{p}.{c}")))
        self.w16 = {k: ops.cast_f16(v[0]) for k, v in self.w.items()} if self.precision == "fp16" else {}
        # fp16-storage path: conv3 and the projection shortcut of a stage's first block as ONE GEMM over the concatenated K axis
        # ([conv2 output | block input] x [w3 | w_shortcut]^T + (b3 + b_shortcut)): the shortcut tensor is never materialised
        self.pair = {}
        if self.precision == "fp16":
            for name, _, _ in STAGES[self.depth]:
                k = f"{name}.0"
                if k + ".shortcut" in self.w:
                    (w3, b3), (ws, bs) = self.w[k + ".conv3"], self.w[k + ".shortcut"]
                    cat = torch.cat([w3.reshape(w3.shape[0], -1), ws.reshape(ws.shape[0], -1)], 1).contiguous()
                    self.pair[k] = (ops.cast_f16(cat), (b3 + bs).contiguous())
        return self

    def _conv(self, x, key, stride=1, pad=0, residual=None, relu=True):
        w, b = self.w[key]
        w16 = self.w16.get(key)
        act = ops.ACT_RELU if relu else ops.ACT_NONE
        if w.shape[1] == 1 and w.shape[2] == 1 and stride == 1:
            N, H, W, C = x.shape
            y = ops.gemm_nt(x.view(-1, C), w.view(w.shape[0], C), b,
                            residual.view(-1, w.shape[0]) if residual is not None else None, act,
                            w16=w16.view(w.shape[0], C) if w16 is not None else None, cw=True)
            return y.view(N, H, W, -1)
        return ops.conv2d_nhwc(x, w, stride, pad, b, residual, act, w16=w16, cw=True)

    def _forward_h16(self, x):
        """forward() with fp16 storage of the intra-bottleneck tensors (see __init__).  Same arithmetic per convolution: fp16 operands,
        f32 accumulation, f32 bias / residual / ReLU; one rounding to fp16 where the f32-storage path rounds while staging."""
        w, w16 = self.w, self.w16
        relu = ops.ACT_RELU
        if self.fuse_stem and tuple(w16["stem"].shape) == (64, 7, 8, 4) and x.shape[-1] == 4:
            x = ops.resnet_stem_pool(x, w16["stem"], w["stem"][1])                             # fp16 [T,H/4,W/4,64], one launch
        else:
            x = ops.conv2d_nhwc_o16(x, w16["stem"], 2, 3, w["stem"][1], relu)                  # fp16 [T,H/2,W/2,64]
            x = ops.maxpool3x3s2(x)                                                            # fp16 [T,H/4,W/4,64]
        feats = {}
        for name, nblocks, first_stride in STAGES[self.depth]:
            for i in range(nblocks):
                stride = first_stride if i == 0 else 1
                k = f"{name}.{i}"
                c1, c2, c3 = k + ".conv1", k + ".conv2", k + ".conv3"
                fused = self.fuse_shortcut and k in self.pair and w[c2][0].shape[0] % 64 == 0
                if fused:
                    sc = None                                                                  # folded into conv3's GEMM below
                elif (k + ".shortcut") in w:
                    if x.dtype == torch.float16:                                               # res2.0: the pooled stem output
                        N_, H_, W_, C_ = x.shape
                        ws = w16[k + ".shortcut"]
                        sc = ops.gemm_nt_x16(x.view(-1, C_), ws.view(ws.shape[0], C_), w[k + ".shortcut"][1]).view(N_, H_, W_, -1)
                    else:
                        sc = self._conv(x, k + ".shortcut", stride=stride, relu=False)         # f32 in (strided pixels), f32 out
                else:
                    sc = x
                N_, H_, W_, C_ = x.shape
                wc1 = w16[c1]
                out = ops.gemm_nt_x16(x.view(-1, C_), wc1.view(wc1.shape[0], C_), w[c1][1], None, relu, out_f16=True).view(N_, H_, W_, -1)
                out = ops.conv_h16(out, w16[c2], 3, stride, w[c2][1], None, relu, out_f16=True)
                N2, H2, W2, C2 = out.shape                                                     # relu(conv3 + shortcut), f32 (HBM-bound: the
                wc3 = w16[c3]                                                                  # register-staged kernel with fp16 A rows)
                if fused:
                    wcat, bcat = self.pair[k]
                    if x.dtype == torch.float16:                                               # res2.0: both sources dense fp16
                        x = ops.gemm_nt_x16_2a(out.view(-1, C2), x.view(-1, x.shape[-1]), wcat, bcat, relu).view(N2, H2, W2, -1)
                    else:                                                                      # res3-5.0: the f32 block input at stride 2
                        x = ops.conv1x1_pair_x16(out, x, stride, wcat, bcat, relu)
                    continue
                x = ops.gemm_nt_x16(out.view(-1, C2), wc3.view(wc3.shape[0], C2), w[c3][1], sc.view(-1, wc3.shape[0]), relu).view(N2, H2, W2, -1)
            if name in self.out_features:
                feats[name] = x
        return feats

    def forward(self, x):
        """x: f32 [T,Hp,Wp,4] (normalised, channel 3 zero) -> {res2..res5} NHWC."""
        if self.w["stem"][0].shape[2] == 8 and x.shape[2] % 2 != 0:
            raise ValueError("ResNet stem with the padded 7x8 kernel needs an even input width (frames are padded to a multiple of 32)")
        if self.precision == "fp16" and self.h16_storage:
            # conv_h16 addresses its fp16 input with 32-bit byte offsets (include/openvis_hip.h: T H W Cin 2 B < 2^31).  The largest such map
            # is NOT res2's: res3.0 runs conv1 at the full H/4 resolution and puts the stride on conv2 (STRIDE_IN_1X1 False), so its conv2
            # reads [T, H/4, W/4, 128] -- twice res2's bytes (15 MB per 720p frame) -- h16_bytes_per_frame() takes the maximum over every
            # conv_h16 call of _forward_h16.  The convolutions are per frame, so a whole video the reference would hand to the backbone in
            # one piece (openvis.py:64) runs as chunks of the frame axis
            T, H, W = x.shape[:3]
            chunk = max(1, (self.H16_BYTE_LIMIT - self.h16_guard_slack(H, W)) // self.h16_bytes_per_frame(H, W))
            if T <= chunk:
                return self._forward_h16(x)
            parts = [self._forward_h16(x[t0:t0 + chunk]) for t0 in range(0, T, chunk)]
            return {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
        x = self._conv(x, "stem", stride=2, pad=3)
        x = ops.maxpool3x3s2(x)
        feats = {}
        for name, nblocks, first_stride in STAGES[self.depth]:
            for i in range(nblocks):
                stride = first_stride if i == 0 else 1
                k = f"{name}.{i}"
                sc = self._conv(x, k + ".shortcut", stride=stride, relu=False) if (k + ".shortcut") in self.w else x
                out = self._conv(x, k + ".conv1")
                out = self._conv(out, k + ".conv2", stride=stride, pad=1)
                x = self._conv(out, k + ".conv3", residual=sc)            # relu(conv3 + shortcut)
            if name in self.out_features:
                feats[name] = x
        return feats

    def h16_conv_inputs(self, H, W):
        """[(block, H_in, W_in, Cin)] of every ops.conv_h16 call of _forward_h16 on padded [T, H, W, 4] frames: conv2 of a bottleneck reads
        conv1's output, which has the block INPUT's resolution (the stride sits on conv2) and the block's bottleneck width."""
        h, w = ((H + 1) // 2 + 1) // 2, ((W + 1) // 2 + 1) // 2         # stem 7x7 / s2 (pad 3), max pool 3x3 / s2 (pad 1)
        out = []
        for name, nblocks, first_stride in STAGES[self.depth]:
            mid = self.w[f"{name}.0.conv2"][0].shape[3] if f"{name}.0.conv2" in self.w else {"res2": 64, "res3": 128, "res4": 256, "res5": 512}[name]
            for i in range(nblocks):
                out.append((f"{name}.{i}", h, w, mid))
                if i == 0 and first_stride == 2:
                    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return out

    def h16_bytes_per_frame(self, H, W):
        """Bytes of the largest fp16 map one frame hands to conv_h16 (720p: res3.0.conv2's [184, 320, 128] = 15.1 MB)."""
        return max(h * w * c * 2 for _, h, w, c in self.h16_conv_inputs(H, W))

    def h16_guard_slack(self, H, W):
        """The frame-independent term of conv_h16's guard, (2 W + 2) Cin 2 bytes, at its largest."""
        return max((2 * w + 2) * c * 2 for _, h, w, c in self.h16_conv_inputs(H, W))

    H16_BYTE_LIMIT = (1 << 31) - (1 << 20)     # per chunk, below conv_h16's 2^31 guard (tests lower it to exercise the chunking)
    __call__ = forward


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    return ResNet(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.RESNETS.OUT_FEATURES,
                  precision=_backbone_precision(cfg))
