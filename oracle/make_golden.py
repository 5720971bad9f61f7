"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz by importing the REFERENCE's own
Python modules from /root/reference (build container only; see oracle/ref_import.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [name ...]
The committed fixtures are data (inputs + the reference's outputs); no reference source travels.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

import numpy as np
import torch

from oracle import ref_import as R


def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def gen_msda():
    f = R.ref("openvis.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    core = f.ms_deform_attn_core_pytorch
    out = {}
    # --- the reference's own fixture: ops/test.py:24-39 (seed 3; double check first, then float) ---
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = _lsi(shapes)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for tag in ("double", "float"):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            o = core(value.double(), shapes, loc.double(), w.double())
        else:
            o = core(value, shapes, loc, w)
        out[f"testpy_{tag}_value"] = value.numpy()
        out[f"testpy_{tag}_loc"] = loc.numpy()
        out[f"testpy_{tag}_w"] = w.numpy()
        out[f"testpy_{tag}_out"] = o.numpy()
    out["testpy_shapes"] = shapes.numpy()
    out["testpy_lsi"] = lsi.numpy()

    # --- model-shaped and edge cases (locations reach outside [0,1] to hit the border rules) ---
    cases = {
        "enc": dict(N=2, M=8, D=32, L=3, P=4, shapes=[(2, 3), (4, 6), (8, 12)], Lq=None),
        "oddD": dict(N=1, M=3, D=30, L=2, P=3, shapes=[(5, 3), (2, 9)], Lq=7),
        "wide": dict(N=1, M=2, D=71, L=1, P=5, shapes=[(3, 4)], Lq=5),
        "L4": dict(N=3, M=4, D=16, L=4, P=4, shapes=[(1, 1), (2, 3), (6, 5), (9, 2)], Lq=11),
    }
    g = torch.Generator().manual_seed(1234)
    for name, c in cases.items():
        shapes = torch.as_tensor(c["shapes"], dtype=torch.long)
        lsi = _lsi(shapes)
        S = int(shapes.prod(1).sum())
        Lq = c["Lq"] or S
        value = torch.randn(c["N"], S, c["M"], c["D"], generator=g)
        loc = torch.rand(c["N"], Lq, c["M"], c["L"], c["P"], 2, generator=g) * 1.5 - 0.25
        w = torch.softmax(torch.randn(c["N"], Lq, c["M"], c["L"] * c["P"], generator=g), -1).view(
            c["N"], Lq, c["M"], c["L"], c["P"])
        o32 = core(value, shapes, loc, w)
        o64 = core(value.double(), shapes, loc.double(), w.double())
        out[f"{name}_value"] = value.numpy()
        out[f"{name}_loc"] = loc.numpy()
        out[f"{name}_w"] = w.numpy()
        out[f"{name}_shapes"] = shapes.numpy()
        out[f"{name}_lsi"] = lsi.numpy()
        out[f"{name}_out32"] = o32.numpy()
        out[f"{name}_out64"] = o64.numpy()
    np.savez_compressed(os.path.join(GOLD, "msda.npz"), **out)
    print("wrote msda.npz", {k: v.shape for k, v in out.items() if k.endswith("out32") or k.endswith("_out")})


GENERATORS = {"msda": gen_msda}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    names = sys.argv[1:] or list(GENERATORS)
    for n in names:
        GENERATORS[n]()
