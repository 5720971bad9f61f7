"""The product package must never import, call or link the oracle (or the reference)."""
import os
import re

from tests.conftest import ROOT


def _sources():
    for d, _, files in os.walk(os.path.join(ROOT, "openvis_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                yield os.path.join(d, f)
    yield os.path.join(ROOT, "openvis_amd", "csrc", "torch_ext", "msda_module.cpp")
    yield os.path.join(ROOT, "train_net.py")


def test_product_does_not_touch_oracle_or_reference():
    bad = []
    for p in _sources():
        s = open(p).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", s, re.M) or "liboracle" in s or "/root/reference" in s:
            bad.append(p)
    assert not bad, bad


def test_bench_uses_the_oracle_only_in_the_cpu_baseline_leg():
    s = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle\b", s, re.M)]
    assert len(hits) == 1
    start = s.index("def cpu_baseline(")
    end = s.index("\ndef ", start + 1)
    assert start < hits[0] < end                      # the timed path never touches the oracle
