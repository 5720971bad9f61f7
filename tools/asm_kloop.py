"""Summarise the innermost loop (the K loop) of one kernel's assembly: python tools/asm_kloop.py kernel.s"""
import collections
import re
import sys

L = open(sys.argv[1]).read().split("\n")
labels = {l.split(":")[0]: i for i, l in enumerate(L) if re.match(r"^\.LBB\d+_\d+:", l)}
best = None
for i, l in enumerate(L):
    m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:          # backward branch = loop
        span = (labels[m.group(1)], i)
        body = L[span[0]:span[1] + 1]
        n_mfma = sum("v_mfma" in x for x in body)
        if n_mfma and (best is None or len(body) < best[2]):
            best = (span[0], span[1], len(body))
s, e, n = best
body = L[s:e + 1]
c = collections.Counter(m.group(1) for m in (re.match(r"\s+([a-z_0-9]+)", x) for x in body) if m)
keep = ("scratch", "s_waitcnt", "v_mfma", "ds_read", "ds_write", "global_load", "global_store", "s_barrier", "v_cvt", "v_fma_mix", "v_pk", "v_mul_f32",
        "v_sub_f32", "v_fma_f32", "s_nop", "buffer")
print(f"innermost MFMA loop: lines {s}-{e} ({n}); total instr {sum(c.values())}")
print({k: v for k, v in sorted(c.items()) if k.startswith(keep)})
print("waits:", [x.strip() for x in body if "s_waitcnt" in x])
