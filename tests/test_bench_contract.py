"""bench.py prints ONE JSON line with the agreed keys (checked on the GPU box; the values are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_line_has_the_agreed_shape():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mpaths/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["paths_per_step"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    ro = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in ro, key
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4 and 0.02 < ro["frac"] < 1.0
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert d["value"] > 100 * cb["value"]  # sanity: three orders of magnitude are expected, two must be there


@pytest.mark.gpu
def test_two_rank_step_gives_the_single_render(rt):
    """bench.py --gpus 2 as the driver starts it (torch.distributed.run, one process per rank), with both ranks on the
    GPUs that are there and the reduce over gloo (BENCH_REHEARSE=1): the step bench.py times — each rank's sample
    range, the sum-reduce to rank 0, rank 0's finalise — must produce the image of one 200-sample render."""
    import socket

    import numpy as np

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["paths_per_step"] == 1920 * 1080 * 200 and "REHEARSAL" in d["data"]
    scene = rt.Scene(rt.gen_scene(1))
    _, q = scene.render(rt.default_camera(1920, 1080), 200, depth=50, seed=0, want_sum=False)
    # two 100-sample sums added instead of one 200-sample sum: f32 rounding may move a channel by one step
    assert abs(d["config"]["image_mean_u8"] - float(np.mean(q))) < 0.01


@pytest.mark.gpu
def test_two_rank_strong_scaling_step_of_the_4k_configuration(rt):
    """bench.py --config c3 --gpus 2 (3840x2160, the samples of ONE render split over the ranks), at a reduced sample
    count and as a rehearsal on the GPUs that are there: "scaling" is strong, the paths of a step do not grow with the
    ranks, and the finalised image is the single render's."""
    import socket

    import numpy as np

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--config", "c3", "--spp", "7"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["paths_per_step"] == 3840 * 2160 * 7
    assert d["roofline"]["traffic"] is None
    scene = rt.Scene(rt.gen_scene(1))
    _, q = scene.render(rt.default_camera(3840, 2160), 7, depth=50, seed=0, want_sum=False)
    assert abs(d["config"]["image_mean_u8"] - float(np.mean(q))) < 0.01
