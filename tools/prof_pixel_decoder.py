"""The pixel decoder (A3-A6) of the headline workload alone: N forwards on fixed backbone features.
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 $GRAFT_REPO_ROOT/tools/prof_pixel_decoder.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

model, sd, text = bench.build_model("cuda")
from openvis_amd import ops
ops.set_f32_gemm_mode(model.f32_gemm_mode)      # a forward sets it in _frames_to_device; the stages are called directly here
frames = bench.synth_frames(5, 720, 1280, 1000, "cuda")
images, _, _ = model.preprocess(frames)
feats = model.backbone(images)
pd = model.sem_seg_head.pixel_decoder
for _ in range(3):
    out = pd.forward_features(feats)
torch.cuda.synchronize()
n = int(os.environ.get("N", "30"))
t0 = time.perf_counter()
for _ in range(n):
    out = pd.forward_features(feats)
torch.cuda.synchronize()
print(f"pixel decoder forward: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per clip (wall, {n} runs)")
