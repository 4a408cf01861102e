"""bench.py's launch paths on one GPU: the plain run, the run with torch.distributed (RCCL) forced on for one rank,
and the path `python bench.py --gpus N` takes without a launcher (fresh child ranks), each as a child process whose
one JSON line is parsed."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*flags):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--workload", "cfg1", "--no-cpu-baseline"] + list(flags)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_plain_forced_dist_and_spawned_ranks_agree():
    plain = _bench()
    forced = _bench("--force-dist")
    spawned = _bench("--spawn")
    for out in (plain, forced, spawned):
        assert out["n_gpus"] == 1 and out["steps"] == 3 and out["unit"] == "points/s" and out["value"] > 0
        assert out["dtype"] == "f32" and "fp16" in out["arithmetic"]
        assert out["roofline"]["frac"] <= 1.0
        assert all(k["roof_frac"] is None or k["roof_frac"] <= 1.0 for k in out["kernels"].values())
    # the same seeded model and points: the first step's loss is the same number on every path
    assert plain["loss_first_step"] == forced["loss_first_step"] == spawned["loss_first_step"]
