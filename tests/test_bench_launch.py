"""bench.py --gpus N: the self-launching multi-rank path.

CPU: the launcher starts exactly N children with a consistent rendezvous environment and never
touches a GPU itself.  GPU (one card): the 2-rank path end to end over gloo (both ranks share the
card): n_gpus = 2 in the line, disjoint ordered seed shards, every environment's return gathered."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_spawns_one_rank_per_gpu(monkeypatch):
    import bench
    started = []

    class FakeProc(object):
        def __init__(self, cmd, env):
            started.append((cmd, env))
            self.returncode = 0

        def wait(self):
            return 0
    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env: FakeProc(cmd, env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    assert len(started) == 4
    ports = {e["MASTER_PORT"] for _, e in started}
    assert len(ports) == 1
    for r, (cmd, e) in enumerate(started):
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"]) == (str(r), str(r), "4")
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_amortised_rate():
    import bench
    # 256 envs, 100 steps in 0.07 s, a 0.02 s reset per 1000 steps
    v = bench.amortised(256, 100, 0.07, 0.02, 1000)
    assert abs(v - 256 * 100 / (0.07 + 0.1 * 0.02)) < 1e-6


@pytest.mark.gpu
def test_two_rank_bench_on_one_gpu():
    env = dict(os.environ, AOMARL_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--envs", "8", "--config", "production_sh_10x10_2m", "--no-cpu-baseline", "--no-side-configs"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    sh = sorted(out["shards"], key=lambda d: d["rank"])
    assert [d["rank"] for d in sh] == [0, 1]
    seeds = [d["first_seed"] + 16 * i for d in sh for i in range(d["envs"])]
    assert seeds == [1234 + 16 * i for i in range(16)]       # one global sequence, disjoint shards
    assert out["gathered"]["n"] == 16
    assert out["value"] > 0 and out["value"] <= out["value_no_reset"]
