"""Lab: does the D2H hand-off of the ten output masks (46 MB per 720p clip, a blit kernel on the side stream) slow the NEXT clip's first kernels?
rocprofv3 shows preprocess_kernel at 16 us alone and ~800 us (= the copy's duration) in the steady state.  Same process, alternating:
  (a) the forward as shipped, (b) final masks replaced by a [n,T,1,1] stub (no 46 MB copy, no final_masks kernel: 85 us).
python tools/exp_d2h_contention.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openvis_amd import ops


def run(model, inputs, n):
    for i in range(3):
        model(inputs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o = None
    for i in range(n):
        o = model(inputs[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    o.wait()
    return dt * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda", 0)
    model, _, _ = bench.build_model(dev)
    clips = [bench.synth_frames(5, 720, 1280, 1000 + i, "cpu").to(dev) for i in range(2)]
    inputs = [[{"image": [f for f in c], "dataset_name": "synthetic_burst_val"}] for c in clips]
    real = ops.final_masks

    def stub(masks, sel_q, Hp, Wp, H, W, OH, OW, column_major=False):
        return torch.zeros((sel_q.numel(), masks.shape[1], 1, 1), dtype=torch.uint8, device=masks.device)
    for rep in range(3):
        a = run(model, inputs, n)
        ops.final_masks = stub
        b = run(model, inputs, n)
        ops.final_masks = real
        print(f"rep {rep}: shipped {a:.3f} ms/step, without the 46 MB hand-off {b:.3f} ms/step, difference {a - b:.3f} ms (final_masks kernel itself: 0.085)")


if __name__ == "__main__":
    main()
