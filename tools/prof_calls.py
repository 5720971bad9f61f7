"""Per-C-ABI-call times of one eval forward: every `_lib.call` is bracketed by HIP events (device sync after the forward), grouped by
(entry point, integer arguments).   python tools/prof_calls.py [model] [top]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from openvis_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "brivis"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
T = 36 if name.startswith("brivis") else 5
RES = (1080, 1920) if name.endswith("swinl") else (720, 1280)
model, sd, text = bench.build_model("cuda", model_name=name)
frames = bench.synth_frames(T, RES[0], RES[1], 1000, "cuda")
inp = [{"image": [f for f in frames.cpu()], "dataset_name": "synthetic_burst_val", "height": RES[0], "width": RES[1]}]
for _ in range(2):
    model(inp)
torch.cuda.synchronize()
log = []
real = _lib.call


def call(fn, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(fn, *args)
    e1.record()
    ints = tuple(a for a in args if isinstance(a, int) and not isinstance(a, bool) and 0 < a < (1 << 31))
    log.append((fn, ints, e0, e1))
    return r


_lib.call = call
import openvis_amd.ops as ops_mod
model(inp)
torch.cuda.synchronize()
_lib.call = real
agg = collections.defaultdict(lambda: [0, 0.0])
for fn, ints, e0, e1 in log:
    a = agg[(fn, ints)]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"{name}: {len(log)} library calls, {tot:.1f} ms of call time in one forward")
for (fn, ints), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{ms:8.3f} ms  {n:4d} x {ms / n * 1e3:9.1f} us  {fn}  {ints}")
