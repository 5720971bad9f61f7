"""GPU: frame-sharded BriVIS (BASELINE configs[3]) with TWO ranks equals the single-rank run: same linker
indices, class probabilities and top-10; each rank returns the masks of its own frames.  (gloo rendezvous, both ranks on
cuda:0: the box has one GPU; RCCL itself is exercised by the driver's multi-GPU bench.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(world, out, port, extra_env=None, worker="_brivis_sharded_worker.py"):
    procs = []
    for r in range(world):
        # a few host threads per rank: several groups of ranks run side by side, and a default-sized OpenMP pool per process (one thread per
        # core of the box, spinning) makes eleven of them slower than one after the other
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="8", MKL_NUM_THREADS="8")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs, out


def _collect(spawned):
    procs, out = spawned
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
    return [json.load(open(f"{out}.{r}")) for r in range(len(procs))]


def _run(world, out, port, extra_env=None, worker="_brivis_sharded_worker.py"):
    return _collect(_spawn(world, out, port, extra_env, worker))


def test_two_rank_frame_sharded_brivis_equals_single_rank(tmp_path):
    single = _run(1, str(tmp_path / "single"), 29641)[0]
    two = _run(2, str(tmp_path / "two"), 29642)
    assert two[0]["range"] == [0, 4] and two[1]["range"] == [4, 7]                  # InferenceSampler layout of 7 frames
    for r in two:
        assert r["indices"] == single["indices"]                                      # replicated linker: identical tracks
        assert np.abs(np.array(r["probs"]) - np.array(single["probs"])).max() < 1e-5  # all-reduced logit sums
        assert r["labels"] == single["labels"] and r["queries"] == single["queries"]
        assert np.abs(np.array(r["scores"]) - np.array(single["scores"])).max() < 1e-5
    assert two[0]["mask_shape"][0] == 4 and two[1]["mask_shape"][0] == 3 and single["mask_shape"][0] == 7
    # the union of the ranks' masks is the single-rank result (pixel counts per instance add up)
    assert [a + b for a, b in zip(two[0]["mask_sums"], two[1]["mask_sums"])] == single["mask_sums"]
    assert two[0]["mask_frames"] == [0, 4] and two[1]["mask_frames"] == [4, 7]
    # output hand-off to ONE rank: rank 0 receives the selected masks of all 7 frames in frame order
    g = _run(2, str(tmp_path / "gather"), 29643, {"OVIS_GATHER_TO": "0"})
    assert g[0]["mask_shape"] == single["mask_shape"] and g[0]["mask_sums"] == single["mask_sums"]
    assert g[0]["frame_sums"] == single["frame_sums"]
    assert g[1]["mask_shape"] == [] and g[1]["labels"] == single["labels"]


@pytest.mark.parametrize("arch,port", [("san_online", 29644), ("openvis_online", 29647)])
def test_two_rank_frame_sharded_online_models_equal_single_rank(arch, port, tmp_path):
    """SURVEY.md 8e row 1 names three frame-shardable architectures: SANOnline and OpenVISOnline share BriVIS' exchange -- all-gather of the
    query embeddings in front of the replicated MinVIS tracker (minvis.py:320-338) -- and then average logits over the clip's frames (SANOnline:
    all-reduce of the per-frame logit sums, san.py:257) or classify CLIP crops (OpenVISOnline: all-gather of the crop logits for the per-query
    mean, openvis.py:130-138).  Two ranks (4 + 3 frames) against one: the same tracks, probabilities, top-10 and masks."""
    env = {"OVIS_SHARD_ARCH": arch}
    sp1 = _spawn(1, str(tmp_path / "single"), port, env)
    sp2 = _spawn(2, str(tmp_path / "two"), port + 1, env)
    spg = _spawn(2, str(tmp_path / "gather"), port + 2, dict(env, OVIS_GATHER_TO="1"))
    single, two, g = _collect(sp1)[0], _collect(sp2), _collect(spg)
    assert two[0]["range"] == [0, 4] and two[1]["range"] == [4, 7]
    for r in two + g:
        assert r["indices"] == single["indices"]                                      # replicated tracker on the gathered embeddings
        assert np.abs(np.array(r["probs"]) - np.array(single["probs"])).max() < 1e-5
        assert r["labels"] == single["labels"] and r["queries"] == single["queries"]
        assert np.abs(np.array(r["scores"]) - np.array(single["scores"])).max() < 1e-5
    assert [a + b for a, b in zip(two[0]["mask_sums"], two[1]["mask_sums"])] == single["mask_sums"]
    assert two[0]["mask_frames"] == [0, 4] and two[1]["mask_frames"] == [4, 7]
    assert g[1]["mask_shape"] == single["mask_shape"] and g[1]["frame_sums"] == single["frame_sums"] and g[0]["mask_shape"] == []
    # window inference on every rank (its 4 / 3 frames as windows of 2) in front of the same exchange
    w = _run(2, str(tmp_path / "windows"), port + 3, dict(env, OVIS_SHARD_WINDOWS="1"))
    for r in w:
        assert r["indices"] == single["indices"] and r["labels"] == single["labels"] and r["queries"] == single["queries"]
        assert np.abs(np.array(r["probs"]) - np.array(single["probs"])).max() < 1e-5
    # (a window of 2 frames reaches other GEMM tilings than a block of 4: f32 summation order, a handful of pixels at |logit| ~ 1e-6)
    assert max(abs(a + b - c) for a, b, c in zip(w[0]["mask_sums"], w[1]["mask_sums"], single["mask_sums"])) <= 16


@pytest.mark.parametrize("model,scaling", [("openvis", "weak"), ("brivis", "strong")])
def test_bench_multi_rank_control_flow(model, scaling, tmp_path):
    """bench.py with WORLD_SIZE=2 (test rig: both ranks on cuda:0, gloo): rank 0 prints ONE JSON line with n_gpus = 2."""
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29651",
                   OVIS_BENCH_TEST_RIG="1", OMP_NUM_THREADS="8")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--model", model, "--frames", "6" if model == "brivis" else "0", "--sharded-frames", "6", "--split-frames", "3"]
                                      + (["--gather-masks"] if model == "brivis" else []), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append([l for l in o.splitlines() if l.startswith("{")])
    assert len(outs[0]) == 1 and len(outs[1]) == 0
    d = json.loads(outs[0][0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and "roofline" in d
    if model == "brivis":        # frame-sharded: the exchange steps of every rank stand on the line
        cm = d["collective_ms"]
        assert len(cm["per_rank"]) == 2 and all(set(r) >= {"all_gather_wait", "linker", "temporal_resampler", "logit_all_reduce", "mask_gather"}
                                                for r in cm["per_rank"][:1])
        assert {"all_gather_wait", "linker", "logit_all_reduce"} <= set(cm["per_rank"][1]) and all(v >= 0 for v in cm["max_over_ranks"].values())
        assert d["frame_sharded"] is None
    else:
        # the default multi-rank command also measures north_star's split: ONE BriVIS clip frame-sharded over the same ranks
        assert d["collective_ms"] is None
        fs = d["frame_sharded"]
        assert fs["value"] > 0 and fs["unit"] == "frames/s" and fs["scaling"] == "strong" and fs["ms_per_step"] > 0
        assert fs["frames_per_rank"] == [3, 3] and fs["world_size_seen"] == 2
        # ... and the OpenVIS row of SURVEY.md 8e: ONE OpenVIS clip over the same ranks (split-KV decoder), with rank 0's un-split run beside it
        sc = d["split_clip"]
        assert sc["value"] > 0 and sc["scaling"] == "strong" and sc["frames_per_rank"] == [2, 1] and sc["unsplit_on_one_gpu"]["ms_per_step"] > 0
        assert {"partial_all_gather", "logit_all_gather"} <= set(sc["collective_ms"]["per_rank"][0])
        cm = fs["collective_ms"]
        assert len(cm["per_rank"]) == 2 and all({"all_gather_wait", "linker", "temporal_resampler", "logit_all_reduce"} <= set(r) for r in cm["per_rank"])
        assert all(v >= 0 for v in cm["max_over_ranks"].values())


def test_rccl_collectives_of_the_frame_sharded_path():
    """One rank per GPU over RCCL (backend "nccl"): the async all-gather on the side stream, the device all-reduce and the mask
    gather -- the branches of openvis_amd/distributed.py that the gloo rigs never execute.  Needs >= 2 GPUs on the box (the
    driver's 8-GPU node; a 1-GPU box skips: RCCL refuses two ranks on one device)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("RCCL test needs >= 2 GPUs (this box has %d)" % n)
    world = min(n, 8)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OVIS_BENCH_TEST_RIG", None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                        "127.0.0.1", "--master-port", "29671", os.path.join(ROOT, "tests", "_rccl_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and f"RCCL_OK world={world}" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_one_rank_rccl_group_runs_the_frame_sharded_path(tmp_path):
    """A ONE-rank "nccl" (= RCCL) process group on the box's single GPU, created in a fresh child process before anything touches the GPU:
    the async all_gather_into_tensor on the side stream, the device all-reduce (with the fp16x2 range flag riding on it) and dist.gather of
    device tensors -- the branches no gloo rig takes -- run under BriVIS(frame_range=(0, T), gather_masks_to=0) and give the un-sharded
    forward's outputs; an out-of-memory error behind the all-gather re-raises without a second collective (tests/_rccl_one_rank_worker.py)."""
    out = str(tmp_path / "rccl1.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29678")
    for k in ("OVIS_BENCH_TEST_RIG", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_one_rank_worker.py"), out], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "RCCL1_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    r = json.load(open(out))
    assert r["ok"] and r["backend"] == "nccl" and r["world"] == 1 and r["librccl"], r


def test_bench_brivis_under_a_one_rank_rccl_group():
    """`bench.py --model brivis --process-group` : the frame-sharded bench control flow (warm_up, all-gather on the side stream, logit
    all-reduce, mask gather, collective_ms) over a 1-rank RCCL group -- what `--gpus 8` runs, minus the peers."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29679")
    for k in ("OVIS_BENCH_TEST_RIG", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "brivis", "--process-group", "--frames", "6", "--steps", "2", "--warmup", "1",
                        "--gather-masks", "--no-alt-splits", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["world_size_seen"] == 1 and line["process_group"] == "nccl" and line["scaling"] == "strong"
    cm = line["collective_ms"]
    assert {"all_gather_wait", "linker", "temporal_resampler", "logit_all_reduce", "mask_gather"} <= set(cm["per_rank"][0])
    assert line["value"] > 0 and line["frames_per_rank"] == [6]
    # stdout is the ONE JSON line: RCCL's version banner (printed to the C stdout at the first communicator) goes to stderr
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-1500:]
    assert len([ln for ln in p.stdout.splitlines() if ln.strip()]) == 1


@pytest.mark.parametrize("model,spans", [("san_online", {"all_gather_wait", "linker", "logit_all_reduce", "mask_gather"}),
                                         ("openvis_online", {"all_gather_wait", "linker", "logit_all_gather", "mask_gather"})])
def test_bench_online_models_frame_sharded_under_a_one_rank_rccl_group(model, spans):
    """`bench.py --model san_online|openvis_online --frame-sharded --process-group`: the other two frame-shardable architectures through the
    bench's frame-sharded control flow over RCCL (one rank)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29682")
    for k in ("OVIS_BENCH_TEST_RIG", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", model, "--frame-sharded", "--process-group", "--frames", "4", "--steps", "2",
                        "--warmup", "1", "--gather-masks", "--no-alt-splits", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["process_group"] == "nccl" and line["scaling"] == "strong" and line["value"] > 0 and line["frames_per_rank"] == [4]
    assert spans <= set(line["collective_ms"]["per_rank"][0]), line["collective_ms"]


def test_bench_openvis_split_clip_under_a_one_rank_rccl_group():
    """`bench.py --process-group` (default model): the headline as always plus `split_clip` -- ONE OpenVIS clip through the split-KV decoder's
    exchange (9 all-gathers of flash partials, 1 of crop logits, mask gather) over a 1-rank RCCL group, with the un-split forward beside it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29680")
    for k in ("OVIS_BENCH_TEST_RIG", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--process-group", "--steps", "5", "--warmup", "1", "--split-frames", "6", "--gather-masks",
                        "--no-alt-splits", "--no-cpu-baseline", "--no-other-configs", "--no-in-flight"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["process_group"] == "nccl" and line["scaling"] == "weak" and line["value"] > 0
    sc = line["split_clip"]
    assert sc["frames_per_rank"] == [6] and sc["value"] > 0 and sc["unsplit_on_one_gpu"]["value"] > 0
    assert {"partial_all_gather", "logit_all_gather", "mask_gather"} <= set(sc["collective_ms"]["per_rank"][0])
    # one rank: the split path is the un-split forward plus the exchange bookkeeping -- within 25 % of it
    assert sc["ms_per_step"] <= 1.25 * sc["unsplit_on_one_gpu"]["ms_per_step"], sc


# ---- ONE OpenVIS clip over several ranks: split-KV offline decoder (OpenVIS row) -------------------------------------------
SPLIT = "_openvis_split_worker.py"


def _masks(out, world):
    return np.concatenate([np.load(f"{out}.{r}.masks.npy") for r in range(world)], axis=1)       # [Q, T, h, w] in frame order


def test_openvis_clip_split_over_ranks_equals_one_rank(tmp_path):
    """5 frames, 192x256, exact-f32 policy: 2 ranks (3 + 2 frames) and 5 ranks (one frame each; OVIS_TEST_ALL_WORLDS=1 adds 3 ranks, 2 + 2 + 1) against the one-GPU forward.  The split changes
    only the order in which the cross-attention's softmax sums are merged (f32 rounding): mask logits to 1e-3 of their scale, class
    probabilities to 1e-5, the same top-10, and the masks' bits up to logits that sit within rounding of zero."""
    # every group of ranks is started at once (independent rendezvous ports, all on cuda:0): the test's time is one model build, not six
    cases = [(2, 29663, [[0, 3], [3, 5]]), (5, 29667, [[0, 1], [1, 2], [2, 3], [3, 4], [4, 5]])]      # ragged blocks; ONE frame per rank
    if os.environ.get("OVIS_TEST_ALL_WORLDS") == "1":
        cases.insert(1, (3, 29664, [[0, 2], [2, 4], [4, 5]]))                                      # 2 + 2 + 1 (measured like the others)
    sp_one = _spawn(1, str(tmp_path / "one"), 29661, {"OVIS_SPLIT_OFF": "1"}, SPLIT)
    # R = 1 through the partial / all-gather / merge path, in a ONE-rank RCCL group (the only way to run the RCCL branches of the exchange
    # on a one-GPU box): the same partials merged by the same arithmetic
    sp_solo = _spawn(1, str(tmp_path / "solo"), 29662, {"OVIS_SPLIT_BACKEND": "nccl", "OVIS_GATHER_TO": "0"}, SPLIT)
    sp_cases = [_spawn(world, str(tmp_path / f"w{world}"), port, None, SPLIT) for world, port, _ in cases]
    sp_gather = _spawn(2, str(tmp_path / "gather"), 29665, {"OVIS_GATHER_TO": "0", "OVIS_SPLIT_CROP_LIST": "device"}, SPLIT)
    one = _collect(sp_one)[0]
    m1 = _masks(str(tmp_path / "one"), 1)
    solo = _collect(sp_solo)[0]
    ms = _masks(str(tmp_path / "solo"), 1)
    assert solo["backend"] == "nccl" and one["backend"] == "gloo"
    assert np.abs(ms - m1).max() <= 1e-4 * np.abs(m1).max() and solo["labels"] == one["labels"] and solo["queries"] == one["queries"]
    assert np.abs(np.array(solo["mask_sums"]) - np.array(one["mask_sums"])).max() <= 8 and np.abs(np.array(solo["probs"]) - np.array(one["probs"])).max() <= 2e-4
    assert {"partial_all_gather", "logit_all_gather", "mask_gather"} <= set(solo["spans"])
    for (world, port, ranges), sp in zip(cases, sp_cases):
        rs = _collect(sp)
        assert [r["range"] for r in rs] == ranges
        mw = _masks(str(tmp_path / f"w{world}"), world)
        assert mw.shape == m1.shape
        err = np.abs(mw - m1).max() / np.abs(m1).max()
        flips = int(((mw > 0) != (m1 > 0)).sum())
        print("OpenVIS clip over %d ranks: mask logits within %.2e of their scale, %d of %d mask bits differ" % (world, err, flips, m1.size))
        assert err <= 1e-3 and flips <= 1e-5 * m1.size + 8
        dp = max(np.abs(np.array(r["probs"]) - np.array(one["probs"])).max() for r in rs)
        dl = max(np.abs(np.array(r["pred_logits"]) - np.array(one["pred_logits"])).max() for r in rs)
        print("    class probabilities within %.2e (peaked-attention tower: it amplifies what the masks move by), query logits within %.2e" % (dp, dl))
        for r in rs:
            assert np.abs(np.array(r["probs"]) - np.array(one["probs"])).max() <= 1e-3                 # every rank: the clip's probabilities
            assert np.abs(np.array(r["pred_logits"]) - np.array(one["pred_logits"])).max() <= 1e-3
            assert r["probs"] == rs[0]["probs"] and r["labels"] == rs[0]["labels"] and r["queries"] == rs[0]["queries"]   # replicated bit for bit
            assert sorted(zip(r["queries"], r["labels"])) == sorted(zip(one["queries"], one["labels"]))
            assert np.abs(np.sort(r["scores"]) - np.sort(one["scores"])).max() <= 1e-3
            assert r["mask_frames"] == r["range"] and r["mask_shape"][0] == r["range"][1] - r["range"][0]
        # the ranks' output masks are the one-GPU masks of their frames (pixel counts per instance add up, a few boundary pixels aside)
        order = {k: i for i, k in enumerate(zip(one["queries"], one["labels"]))}
        tot = np.zeros(len(order), np.int64)
        for r in rs:
            for q, l, n in zip(r["queries"], r["labels"], r["mask_sums"]):
                tot[order[(q, l)]] += n
        assert np.abs(tot - np.array(one["mask_sums"])).max() <= 16, (tot.tolist(), one["mask_sums"])
    g = _collect(sp_gather)
    assert g[0]["mask_shape"] == one["mask_shape"] and g[1]["mask_shape"] == [] and g[1]["labels"] == g[0]["labels"]
    assert np.abs(np.array(g[0]["frame_sums"]) - np.array(one["frame_sums"])).max() <= 16
    assert np.abs(np.array(g[0]["probs"]) - np.array(one["probs"])).max() <= 1e-3                     # device crop list on the ranks, same result


def test_openvis_c2_split_over_two_ranks_against_the_oracle_golden(tmp_path):
    """BASELINE configs[1] at full size (5 frames, 720p, 482 classes) with the clip on TWO ranks (3 + 2 frames), held against the ORACLE's
    values of tests/golden/c2_sharp_classes.npz exactly as the one-GPU forward is (test_c2_720p_gpu.py, separated label space): the EXACT
    top-10 (query, label) set, scores to 1e-3, class probabilities to 3e-3, output-mask pixel counts to 2e-3."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2_sharp_classes.npz"))
    rs = _run(2, str(tmp_path / "c2s"), 29666, {"OVIS_SPLIT_CASE": "c2s", "OVIS_GATHER_TO": "0"}, SPLIT)
    rows = g["rows"].tolist()
    sr = {(rows[r], int(l)): float(s) for r, l, s in zip(g["top_rows"], g["top_labels"], g["top_scores"])}
    ref_counts = {(rows[r], int(l)): int(n) for r, l, n in zip(g["top_rows"], g["top_labels"], g["top_mask_counts"])}
    for r in rs:
        sg = {(q, l): s for q, l, s in zip(r["queries"], r["labels"], r["scores"])}
        assert set(sg) == set(sr), (sorted(sg), sorted(sr))
        ds = max(abs(sg[k] - sr[k]) for k in sg)
        dp = np.abs(np.array(r["probs"])[rows] - g["probs"]).max()
        assert ds <= 1e-3 and dp <= 3e-3, (ds, dp)
    assert rs[0]["mask_shape"] == [5, 720, 1280] and rs[1]["mask_shape"] == []
    counts = {(q, l): n for q, l, n in zip(rs[0]["queries"], rs[0]["labels"], rs[0]["mask_sums"])}
    dc = max(abs(counts[k] - ref_counts[k]) / max(ref_counts[k], 1) for k in counts)
    print("C2 over two ranks vs the oracle: top-10 sets equal (labels %s), max score diff %.2e, max probability diff %.2e, pixel counts within %.2e"
          % (sorted({l for _, l in sr}), ds, dp, dc))
    assert dc <= 2e-3


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_bench_line_survives_a_rank_that_fails_inside_a_side_measurement(fail_rank):
    """The multi-rank side measurements run collectives; a rank that raises inside one leaves its peer waiting in a collective nobody will
    complete.  bench.py runs them LAST under a deadline (_SideGuard): rank 0 prints the complete headline line with an error in that field
    and every rank exits 0 -- whichever rank failed."""
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29691 + fail_rank),
                   OVIS_BENCH_TEST_RIG="1", OMP_NUM_THREADS="8", OVIS_BENCH_FAIL_SIDE_RANK=str(fail_rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--sharded-frames", "0",
                                       "--split-frames", "2", "--side-timeout", "25", "--no-alt-splits"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append([ln for ln in o.splitlines() if ln.strip()])
    assert len(outs[0]) == 1 and outs[1] == []
    d = json.loads(outs[0][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "roofline" in d and d["frame_sharded"] is None
    assert "error" in d["split_clip"], d["split_clip"]


def test_bench_eight_rank_control_flow_on_one_gpu():
    """The command the driver's scaling run issues at N = 8, on the rig (eight ranks on cuda:0, gloo): one JSON line from rank 0 with the
    clip-replica headline, `frame_sharded` (36 BriVIS frames -> 5,5,5,5,4,4,4,4) and `split_clip` (8 OpenVIS frames, one per rank: the
    decoder's partials merged from EIGHT blocks).  The numbers mean nothing here (one GPU, host-staged collectives); the control flow at the
    world size of the real run is what is under test."""
    W = 8
    procs = []
    for r in range(W):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(W), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29699",
                   OVIS_BENCH_TEST_RIG="1", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--steps", "2", "--warmup", "1", "--no-alt-splits",
                                       "--gather-masks"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=1200)
        assert p.returncode == 0, e[-3000:]
        outs.append([ln for ln in o.splitlines() if ln.strip()])
    assert [len(o) for o in outs] == [1] + [0] * (W - 1)
    d = json.loads(outs[0][0])
    assert d["n_gpus"] == W and d["world_size_seen"] == W and d["scaling"] == "weak" and d["value"] > 0 and d["frames_per_rank"] == [5] * W
    fs, sc = d["frame_sharded"], d["split_clip"]
    assert fs["frames_per_rank"] == [5, 5, 5, 5, 4, 4, 4, 4] and fs["value"] > 0 and fs["world_size_seen"] == W
    assert sc["frames_per_rank"] == [1] * W and sc["value"] > 0 and sc["unsplit_on_one_gpu"]["ms_per_step"] > 0
    assert len(sc["collective_ms"]["per_rank"]) == W and {"partial_all_gather", "logit_all_gather", "mask_gather"} <= set(sc["collective_ms"]["per_rank"][0])
