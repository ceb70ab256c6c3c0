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

        def poll(self):
            return 0
    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env, stderr=None: FakeProc(cmd, env))
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


def test_launcher_stops_the_other_ranks_when_one_fails(monkeypatch, capsys):
    """a dead rank must not leave its siblings in a barrier until the group timeout."""
    import bench
    procs = []

    class FakeProc(object):
        def __init__(self, rank, errfile):
            self.rank, self.terminated, self.polls = rank, False, 0
            if rank == 1:
                errfile.write("RuntimeError: no HIP device\n")
                errfile.flush()

        def poll(self):
            self.polls += 1
            if self.rank == 1:
                return 3 if self.polls > 2 else None     # dies after a moment
            return -15 if self.terminated else None       # the others would run forever

        def terminate(self):
            self.terminated = True

        def wait(self, timeout=None):
            return -15

        def kill(self):
            self.terminated = True

    def popen(cmd, env, stderr=None):
        p = FakeProc(int(env["RANK"]), stderr)
        procs.append(p)
        return p
    monkeypatch.setattr(bench.subprocess, "Popen", popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 3
    assert [p.terminated for p in procs] == [True, False, True]
    err = capsys.readouterr().err
    assert "rank 1" in err and "no HIP device" in err


def test_dtype_string_names_what_was_launched():
    """the bench line's dtype is built from the launch counters, not typed in."""
    from ao_marl_amd import libaomarl as la
    f32_only = {"frame_kernel_dft:f32_mfma": 20, "gemm:f32_mfma": 120, "actor:f32_mfma": 20,
                "frame_kernel_dft:split_f16_mfma": 0, "gemm:split_f16_mfma": 0}
    assert la.dtype_string(f32_only) == "f32"
    mixed = dict(f32_only, **{"gemm:split_f16_mfma": 6})
    d = la.dtype_string(mixed)
    assert d != "f32" and "split" in d and "gemm" in d and "frame_kernel_dft" not in d.split("; f32 in:")[0]
    fast = {"frame_kernel_dft:split_f16_mfma": 20, "gemm:split_f16_mfma": 120, "denoiser:split_f16_mfma": 20,
            "actor:f32_mfma": 20}
    d = la.dtype_string(fast)
    head = d.split("; f32 in:")[0]
    assert all(k in head for k in ("frame_kernel_dft", "gemm", "denoiser")) and "actor" in d.split("; f32 in:")[1]


def test_precision_mode_default_is_the_references_fp32():
    """Host-only calls of the library (no GPU needed): f32 is the default, the fast mode is opt-in and
    the families the counters know are the ones include/aomarl.h documents."""
    from ao_marl_amd import libaomarl as la
    L = la.load()
    if not os.environ.get("AOMARL_PRECISION"):
        assert la.get_precision() == "f32"
    keep = la.get_precision()
    try:
        la.set_precision("split_f16")
        assert L.aomarl_get_precision() == la.PRECISION_SPLIT_F16
        la.set_precision("f32")
        assert L.aomarl_get_precision() == la.PRECISION_F32
        with pytest.raises(ValueError):
            la.set_precision("bf16")
        assert L.aomarl_set_precision(7) != 0
    finally:
        la.set_precision(keep)
    fam = set(la.arith_launches())
    assert {"frame_kernel_dft:f32_mfma", "frame_kernel_dft:split_f16_mfma", "gemm:f32_mfma", "gemm:split_f16_mfma",
            "denoiser:f32_mfma", "denoiser:split_f16_mfma", "actor:f32_mfma"} == fam


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
    assert out["dtype"] == "f32" and not any("split" in k for k in out["launched"])
    assert all(d["ms_per_step"] > 0 for d in sh)


@pytest.mark.gpu
def test_five_rank_bench_on_one_gpu():
    """More ranks than any box of the pool has cards, on ONE card over gloo: five processes through the calibration
    cache's lock (one calibrates, four load), five shards of one global seed sequence, the port hand-out of
    `spawn_ranks`, the gathers at world size 5.  (Five, not eight: a GPU box admits six processes of one user on its
    card at a time, and the test runner is one of them.)"""
    import tempfile
    env = dict(os.environ, AOMARL_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory(prefix="aomarl_calib_5rank_") as cache:
        env["AOMARL_CALIB_CACHE"] = cache       # empty: exactly one rank must calibrate
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--steps", "4", "--warmup", "1", "--settle", "4",
               "--envs", "8", "--config", "production_sh_10x10_2m", "--no-cpu-baseline", "--no-side-configs",
               "--episode-len", "20"]
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        assert len([f for f in os.listdir(cache) if f.endswith(".npz")]) == 1
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 5 and out["scaling"] == "weak" and out["gathered"]["n"] == 40
    sh = sorted(out["shards"], key=lambda d: d["rank"])
    assert [d["rank"] for d in sh] == list(range(5))
    seeds = [d["first_seed"] + 16 * i for d in sh for i in range(d["envs"])]
    assert seeds == [1234 + 16 * i for i in range(40)]
    stats = [d["calibration_cache"] for d in sh]
    assert sum(c["miss"] for c in stats) == 1 and sum(c["hit_disk"] for c in stats) == 4, stats
    assert out["value"] > 0 and all(d["ms_per_step"] > 0 for d in sh)


@pytest.mark.gpu
def test_two_rank_bench_over_rccl():
    """One rank per GPU over RCCL (backend "nccl"), as the driver launches the scaling runs; needs two
    cards (the one-GPU boxes skip it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d)" % torch.cuda.device_count())
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AOMARL_DIST_BACKEND"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--envs", "8", "--config", "production_sh_10x10_2m", "--no-cpu-baseline", "--no-side-configs"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["gathered"]["n"] == 16
    sh = sorted(out["shards"], key=lambda d: d["rank"])
    assert [d["rank"] for d in sh] == [0, 1] and all(d["ms_per_step"] > 0 for d in sh)


@pytest.mark.gpu
def test_one_rank_bench_over_rccl():
    """The RCCL leg of bench.py on ONE card: AOMARL_DIST_FORCE=1 makes a one-rank run initialise the "nccl" process
    group and take the multi-rank path (barriers around the timed region, MAX over ranks of the elapsed time on a
    device tensor, gather of shards and episode returns) -- what the scaling runs execute per rank, minus the peers."""
    env = dict(os.environ, AOMARL_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("AOMARL_DIST_BACKEND", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
           "--envs", "8", "--config", "production_sh_10x10_2m", "--no-cpu-baseline", "--no-side-configs"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["gathered"]["n"] == 8 and out["value"] > 0
    assert [d["rank"] for d in out["shards"]] == [0]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "split_f16"])
def test_bench_dtype_follows_the_precision_mode(mode):
    """The label comes from the launch counters of the timed region: an all-fp32 pass launches no
    split-fp16 kernel of any family, the fast mode names every family it used."""
    import bench
    import torch
    from ao_marl_amd import libaomarl as la
    keep = la.get_precision()
    la.set_precision(mode)
    try:
        w = bench.Workload("production_sh_10x10_2m", 8, 0, 1, "cuda:0")
        # the general chain and the extrusion rounds (on this small system the default is the one-launch move and the
        # two-kernel chain, fp32 vector arithmetic in both modes: no internal GEMM runs at all)
        w.sim.set_option("small_chain", 0)
        w.sim.set_option("small_move", 0)
        w.reset()
        w.timed(3, 1)
        launched = {k: v for k, v in w.launched.items() if v}
        d = la.dtype_string(w.launched)
        if mode == "f32":
            assert d == "f32" and not any("split" in k for k in launched), launched
            assert w.sim.frame_kernel_name().endswith("false>")
        else:
            assert "frame_kernel_dft" in d and "gemm" in d and d != "f32"
            assert launched.get("frame_kernel_dft:split_f16_mfma") == 3 and launched.get("gemm:split_f16_mfma", 0) > 0
            assert w.sim.frame_kernel_name().endswith("true>")
        assert launched.get("actor:f32_mfma") == 3
        del w
        torch.cuda.synchronize()
    finally:
        la.set_precision(keep)
