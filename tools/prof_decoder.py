"""The masked-attention decoder (A7-A8) of the headline workload alone: 30 forwards on fixed pixel-decoder outputs.
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 $GRAFT_REPO_ROOT/tools/prof_decoder.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

model, sd, text = bench.build_model("cuda")
frames = bench.synth_frames(5, 720, 1280, 1000, "cuda")
images, _, _ = model.preprocess(frames)
feats = model.backbone(images)
mf, _, ms = model.sem_seg_head.pixel_decoder.forward_features(feats)
dec = model.sem_seg_head.predictor
for _ in range(3):
    out = dec(ms, mf)
torch.cuda.synchronize()
n = int(os.environ.get("N", "30"))
t0 = time.perf_counter()
for _ in range(n):
    out = dec(ms, mf)
torch.cuda.synchronize()
print(f"decoder forward: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per clip (wall, {n} runs)")
