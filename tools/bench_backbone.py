"""A2 alone: ResNet-50 on 5 x 720p frames under the tile-selection modes of gemm_f16cvt (lab switch ovis_f16cvt_small_n)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openvis_amd import _lib

model, sd, text = bench.build_model("cuda:0")
frames = bench.synth_frames(5, 720, 1280, 3, "cuda")
images, _, _ = model.preprocess(frames)
for rep in range(2):
    for mode in (0, 1, 2):
        _lib.call("ovis_f16cvt_small_n", mode)
        for _ in range(3):
            model.backbone(images)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            model.backbone(images)
        torch.cuda.synchronize()
        print(f"f16cvt tile mode {mode}: backbone {(time.perf_counter() - t) / 20 * 1e3:.3f} ms")
_lib.call("ovis_f16cvt_small_n", 2)
