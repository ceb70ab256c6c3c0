#!/usr/bin/env python3
"""Headline benchmark: env steps/sec (AO frames/sec) of the per-timestep hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, one
process per GPU, before anything in this process touches a GPU.

One step = one pass of the hot path over one batch of synthetic environments per GPU:
    batched 14-agent SAC actor forward  ->  next_part_two (Btt correction, delay, DM shapes,
    Strehl)  ->  per-agent rewards  ->  next_part_one (phase-screen extrusion, target trace + PSF,
    WFS trace + spot images + COG, integrator)  ->  state assembly.
Workload (BASELINE.json configs[2]): production_sh_40x40_8m_3layers, 256 atmosphere seeds per
GPU, 14 agents (13 x 98 Btt modes + tip-tilt), windowed states (w = 20).  Environments are
independent, so N GPUs run N x 256 seeds with no data-path collective (weak scaling); the
collectives are the MAX over ranks of the timed region and one all_gather of per-environment
returns in the epilogue.

What the ONE JSON line (rank 0) says:
  value            env steps/s with the metric's episode structure: one env.reset() per
                   `--episode-len` (1000) frames, timed in this run, amortised into the K timed steps:
                   N_env * K / (T_K + K / 1000 * T_reset).  `value_no_reset` is N_env * K / T_K.
  dtype            built from what the timed region LAUNCHED (aomarl_arith_launches): "f32" when every
                   kernel family ran its fp32 form -- the library's default and the reference's
                   arithmetic (Rtc_FFF, shesha/sutra_wrap.py:49) -- otherwise every split-fp16 family by
                   name.  `value` is the all-fp32 pass.  `fast_mode` repeats the loop under
                   aomarl_set_precision(SPLIT_F16) (fp16 operand pairs, 22-bit mantissa, fp32 accumulation
                   in the frame kernel's DFTs and the internal GEMMs) with its own dtype / roofline.
  roofline         the one-pass frame kernel, from its launch duration measured inside the timed region by
                   HIP event pairs the library attaches to that dispatch on its own stream
                   (aomarl_frame_kernel_time).  fp32 instantiation: bound = "mfma" -- algorithmic flops
                    against the 157.3 TFLOP/s dense fp32 matrix peak (it is
                   compute-bound: fp32 matrix and vector instructions share the issue slots); `hbm_frac`
                   is the same launch against the 8 TB/s HBM peak.  split-fp16 instantiation
                   (`fast_mode.roofline`): bound = "hbm", algorithmic bytes / duration.  `traffic` =
                   HBM bytes per launch from rocprofv3 --pmc passes kept under profiles/ -- refused
                   (null) unless that file names the kernel instantiation this run launched.
  configs          side figures of BASELINE configs[1] (10x10, 64 envs, 2 agents) and configs[4]
                   (noise + denoiser) from the same process (not `value`).
  cpu_baseline     the CPU oracle (C restatement of the COMPASS frame, oracle/aoref.c) on this host:
                   1 thread and all cores, bounded samples.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime initialises: see ao_marl_amd/__init__.py

WORKLOAD = "production_sh_40x40_8m_3layers"
SMALL = "production_sh_10x10_2m"
NOISY = "production_sh_40x40_8m_3layers_d0_noise"
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PMC_FILE = None              # default: the newest profiles/r*_pmc_frame_kernel.json (by round tag)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps right in front of the timed region")
    ap.add_argument("--settle", type=int, default=100,
                    help="episode position of the warm-up: untimed steps behind the reset in front of it.  The first ~40 "
                         "frames behind a reset run 3-10 %% slower (the loop is still closing); they are timed on their "
                         "own and their excess over the steady rate is amortised with the reset (`post_reset_transient_ms`)")
    ap.add_argument("--envs", type=int, default=256, help="environments per GPU")
    ap.add_argument("--config", default=WORKLOAD)
    ap.add_argument("--episode-len", type=int, default=1000,
                    help="frames per episode: one reset is amortised over this many steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the configs[1] / configs[4] side figures and the fp32 pass")
    ap.add_argument("--unfused", action="store_true",
                    help="separate science / WFS passes instead of the one-pass frame kernel")
    ap.add_argument("--denoiser", default=None,
                    help="WFS-image denoiser in the loop: 'shipped' (the reference's trained network, "
                         "ao_marl_amd/data) or a state_dict file: BASELINE configs[4] as the workload")
    ap.add_argument("--residual-shortcut", action="store_true",
                    help="residual modes from one product with v2m.cmat instead of do_control + volts2modes "
                         "(VecAoEnv.residual_shortcut, an opt-in of the package: this bench's default together with the frame pipeline)")
    ap.add_argument("--no-residual-shortcut", action="store_true",
                    help="the reference's order (do_control, then v2m . err) also with the frame pipeline")
    ap.add_argument("--frame-pipeline-always", action="store_true",
                    help="VecAoEnv(frame_pipeline=True): pipelined whenever eligible, without the package's probe of "
                         "both call orders behind the first reset (the default, frame_pipeline='auto')")
    ap.add_argument("--no-whole-episode", action="store_true",
                    help="skip the measured whole episode (reset + episode_len steps) behind the timed region")
    ap.add_argument("--reset-prefetch", action="store_true",
                    help="(the default since round 6; kept for old command lines)")
    ap.add_argument("--no-reset-prefetch", action="store_true",
                    help="resets in the open.  Default: VecAoEnv.throughput_mode(reset_prefetch='same') -- the next episode's "
                         "screens grow beside the running episode (aomarl_reset_prefetch_*, bit-identical to the reset in "
                         "the open), as sac.train_agent runs its episodes; the timed steps carry their share of those rounds")
    ap.add_argument("--timed-only", action="store_true",
                    help="nothing but warm-up + the timed region on the GPU (profiling): no plain-order pass, no stage split")
    ap.add_argument("--no-frame-pipeline", action="store_true",
                    help="plain call order: every frame behind the control / agent chain of the previous one")
    ap.add_argument("--graph-step", action="store_true",
                    help="aomarl_env_step replayed as HIP graphs (with --no-prefetch: ONE stream, a linear graph)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="move the atmosphere in front of the image kernels (no side stream)")
    ap.add_argument("--no-defer", action="store_true",
                    help="materialise the stack-array DM shapes instead of evaluating them from the "
                         "voltages inside the frame kernel")
    ap.add_argument("--precision", default="f32", choices=("f32", "split_f16"),
                    help="arithmetic of the MAIN timed pass (libaomarl.set_precision); f32 is the reference's")
    ap.add_argument("--pmc", default=PMC_FILE,
                    help="profiles/<file> with the HBM bytes per launch from rocprofv3 --pmc passes (default: the newest "
                         "profiles/r*_pmc_frame_kernel.json)")
    ap.add_argument("--random-actor", action="store_true",
                    help="actors with a random last layer (rounds 1-4's bench policy: state-dependent actions at full scale, "
                         "the loop does not stay closed) instead of SURVEY 8(d)'s policy (Xavier, seed 1234, last layer zero)")
    ap.add_argument("--no-episode-end-to-end", action="store_true",
                    help="skip the measured training episode (reset + episode_len steps + episode_len SAC updates)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------ launcher
def spawn_ranks(args):
    """`--gpus N` without a launcher: start N ranks of this script, one per GPU, and pass rank 0's
    line through.  Runs before this process has touched a GPU (no HIP call, no torch.cuda query);
    children are fresh processes -- nothing is re-exec'd.  All children are polled together: the
    first one that fails takes its siblings down (they would otherwise sit in a barrier until the
    process-group timeout) and its stderr tail is shown."""
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ef = tempfile.TemporaryFile(mode="w+")
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stderr=ef))
    rc, failed = 0, None
    live = set(range(len(procs)))
    while live and failed is None:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc, failed = code, r
                break
        if live and failed is None:
            time.sleep(0.05)
    if failed is not None:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    for r, ef in enumerate(errs):                # children's stderr: everything of a failed rank, else as is
        ef.seek(0)
        txt = ef.read()
        if txt:
            sys.stderr.write(("[rank %d stderr]\n" % r if failed is not None else "") + txt[-8000:])
        ef.close()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d; the other ranks were stopped\n" % (failed, rc))
    sys.exit(rc)


# ------------------------------------------------------------------------------------ models
def frame_kernel_model(s, nenv):
    """Algorithmic bytes / flops of one launch of the one-pass frame kernel:
    every lit 16x16 tile of the pupil grid is read once per layer (1 KiB), 2 slopes per sub-aperture
    and 16 complex PSF-row values per pupil row are written; tip-tilt planes, masks and twiddles are
    shared by all environments (L2 / Infinity-Cache resident)."""
    nl = s.nscreens
    lit = int((s.spupil.reshape(s.pupdiam // 16, 16, s.pupdiam // 16, 16).sum(axis=(1, 3)) > 0).sum()) \
        if s.pupdiam % 16 == 0 else 0
    fft_flops = 16 * 5 * 64 * 6 + 32 * 5 * 64 * 6        # pruned radix-2 count: 16 rows -> 32x32 of 64^2
    byts = nenv * (lit * nl * 1024.0 + s.nvalid * 8.0 + s.pupdiam * 16 * 8.0)
    flops = nenv * (s.nvalid * float(fft_flops + 256 * 20) + lit * 256 * 16 * 8.0)
    # what the fp32 slopes-only instantiation EXECUTES on the matrix pipe since round 4's quadratic-form centre of
    # gravity: per sub-aperture 28 v_mfma_f32_16x16x4_f32 (48 with the pruned transform), per lit tile 8 for the PSF
    # rows and 2 for the stack-array lattice; 2048 flop each
    executed = nenv * (s.nvalid * 28 + lit * 10) * 2048.0
    return dict(bytes=byts, flops=flops, lit_tiles=lit, executed_matrix_flops=executed)


def usable_cores(host_cores):
    """Cores this process may actually run on: the scheduler affinity and the cgroup CPU quota of the
    box's share (a 256-core host hands a 1-GPU job 16 of them; 256 OpenMP threads on 16 cores ran 5x
    SLOWER than one thread)."""
    n = host_cores
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    if "OMP_NUM_THREADS" in os.environ:
        try:
            n = min(n, int(os.environ["OMP_NUM_THREADS"]))
        except ValueError:
            pass
    # the GPU pool's rule of thumb is 16 host cores per GPU: never spin more threads than that
    return max(1, min(n, 16))


def cpu_baseline(s, budget_s=8.0):
    """The CPU oracle stepping ONE environment of the workload (same calibrated command matrix) on
    this host: single thread, then every core; each bounded to ~budget_s seconds of frames."""
    import numpy as np
    from oracle import aoref
    L = aoref.lib()

    class Sim(aoref.OracleSim):
        def reset(self, seed):          # short refresh: frame cost does not depend on content
            self.seed, self.frame = int(seed), 0
            self.accumx = np.zeros(s.nscreens, dtype=np.float32)
            self.accumy = np.zeros(s.nscreens, dtype=np.float32)
            self.ext_count = [0] * s.nscreens
            for l in range(s.nscreens):
                for _ in range(8):
                    self._extrude(l, 1 if s.deltax[l] > 0 else -1)
            self._alloc_ctrl()
            self.reset_strehl()

    o = Sim(s, seed=1234)

    def run(threads):
        got = L.aoref_set_threads(threads)
        o.next_part_two(None)
        o.next_part_one()                   # warm caches / thread pool
        t0, frames = time.time(), 0
        while True:
            o.next_part_two(None)
            o.next_part_one()
            frames += 1
            dt = time.time() - t0
            if dt > budget_s or frames >= 200:
                break
        return got, frames, dt
    ncores = usable_cores(L.aoref_max_threads())
    t1, f1, d1 = run(1)
    tn, fn, dn = run(ncores)
    return {"value": fn / dn, "unit": "env steps/s", "cores": tn, "kind": "port",
            "value_1_thread": f1 / d1, "host_cores": L.aoref_max_threads(),
            "sample": "integrator frames of 1 environment of %s (extrusion, 2 raytraces, %d zero-padded "
                      "64x64 FFT spots, COG, cmat GEMV, delay, DM shapes, %d^2 FFT PSF; SAC actor not "
                      "included): %d frames in %.1f s on 1 thread, %d frames in %.1f s on %d OpenMP threads"
                      % (s.name, s.nvalid, s.npsf, f1, d1, fn, dn, tn)}


def sac_update_rate(layout, device, n_updates=100, batch=256, rows=20000):
    """Secondary figure (not `value`): wall time of one aomarl_sac_update of every agent on a batch of
    256 replay rows per agent (the learner side of the path)."""
    import torch
    from ao_marl_amd.sac import BatchedSAC
    sac = BatchedSAC(layout, dict(memory_size=rows), device=device)
    g = torch.Generator(device=device).manual_seed(1)
    sac.memory.push(torch.randn(rows, layout.state_dim, generator=g, device=device),
                    torch.rand(rows, layout.action_dim, generator=g, device=device) * 2 - 1,
                    -torch.rand(rows, layout.n_agents, generator=g, device=device),
                    torch.randn(rows, layout.state_dim, generator=g, device=device), 1.0)
    for _ in range(5):
        sac.update_from_memory(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_updates):
        sac.update_from_memory(batch)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_updates * 1e3
    return {"ms_per_update": ms, "host_enqueue_ms_per_update": t_enq / n_updates * 1e3, "agents": layout.n_agents,
            "batch_per_agent": batch, "updates_per_s": 1e3 / ms, "kernel": "aomarl_sac_update"}


def episode_end_to_end(env, layout, device, episode_len, batch=256):
    """The episode a trainer sees, as the reference runs and prints it (train_rpc.py:503-549: env.reset, max_steps x
    (choose_action, env_step, replay bookkeeping), then update_all_agents = `updates_per_episode_rpc` SAC updates per
    agent, :1084-1133; wall time printed at :546-549): ao_marl_amd.sac.run_episode on this environment -- reset +
    episode_len steps with sampled actions and the delayed-MDP replay writes + episode_len updates of every agent on
    batches of `batch` -- between two device synchronisations."""
    import torch
    from ao_marl_amd.sac import BatchedSAC, run_episode
    rows = env.nenv * episode_len
    sac = BatchedSAC(layout, dict(memory_size=rows), device=device)
    run_episode(env, sac, max_steps=12, train=True, n_updates=12, batch_size=batch)     # allocations, first launches
    gc.collect()
    gc.disable()
    try:
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run_episode(env, sac, max_steps=episode_len, train=True, n_updates=episode_len, batch_size=batch, timing=tm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    res = {"seconds": dt, "steps": episode_len, "updates": int(out["updates"]), "envs": env.nenv,
           "agents": layout.n_agents, "batch_per_agent": batch,
           "reset_and_steps_s": tm.get("steps_s"), "updates_s": tm.get("updates_s"),
           "env_steps_per_s": env.nenv * episode_len / dt,
           "mean_strehl_le": float(out["sr_le"].mean()), "mean_return": float(out["r_total"].mean()),
           "what": "reset + %d x (actor forward with sampled actions, env step, delayed-MDP replay write) + %d SAC updates "
                   "of %d agents (batch %d each) between two synchronisations: the reference's printed episode time "
                   "(train_rpc.py:503-549, 1084-1133)" % (episode_len, episode_len, layout.n_agents, batch)}
    del sac
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------ one workload
class Workload(object):
    """A VecAoEnv + random-init batched SAC actors for one BASELINE configuration."""

    def __init__(self, config, envs, rank, world, device, denoiser=None, prefetch=True, pipeline="auto",
                 reset_prefetch=None, agents=None, random_actor=False, shortcut=True):
        import torch
        from ao_marl_amd.agents import BatchedGaussianPolicy
        from ao_marl_amd.env import VecAoEnv, load_norm
        self.config, self.envs, self.device = config, envs, device
        if "10x10" in config:
            rl, n_modal = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5), 1
        elif agents == 43:              # the reference's published layout (README.md:116-119): 42 x 30 modes + tip-tilt
            rl, n_modal = dict(n_zernike_start_end=[0, 1260], n_reverse_filtered_from_cmat=5), 42
        else:
            rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5,
                      window_n_zernike=20, include_tip_tilt_windowed=True)
            n_modal = 13
        autoencoder = None
        if denoiser:
            from ao_marl_amd.denoiser import SubapDenoiser
            autoencoder = SubapDenoiser.load(None if denoiser == "shipped" else denoiser, device=device)
        norm_kw = {}
        try:
            load_norm(config)
        except FileNotFoundError:
            # no recorded statistics for this configuration: borrow the closest one's (synthetic
            # benchmark data; the arithmetic per step is identical)
            nrm, zn = load_norm("production_sh_40x40_8m_3layers_d1_noise" if "noise" in config
                                else WORKLOAD)
            norm_kw = dict(norm=nrm, zn_norm=zn)
        # independent shards: rank r owns seeds [r*envs, (r+1)*envs) of the global seed sequence
        from ao_marl_amd.dist import shard_seeds
        self.first_seed = shard_seeds(1234, envs, rank, stride=16)
        self.env = VecAoEnv(config, envs, rl, initial_seed=self.first_seed, seed_stride=16,
                            n_agents_modal=n_modal, device=device, autoencoder=autoencoder,
                            prefetch_atmos=prefetch, frame_pipeline=pipeline if prefetch else False,
                            **norm_kw)
        # the loop sac.train_agent runs (VecAoEnv.throughput_mode: frame pipeline where eligible and not slower, residual
        # modes from one product, the next reset's screens grown beside the episode) -- bench and trainer time the same
        # steps; every piece has its switch on the command line
        if pipeline == "auto" and prefetch:
            self.env.throughput_mode(reset_prefetch=reset_prefetch)
        elif reset_prefetch is not None and prefetch:
            self.env.supervisor.reset_prefetch = reset_prefetch
        self.env.residual_shortcut = bool(shortcut)
        self.layout = self.env.layout
        # SURVEY 8(d)'s policy = the reference's at the start of training: Xavier-uniform weights, seed 1234, last
        # layer zero (model_rpc.py:10-14,103-106, `initialize_last_layer_0: True`), actions SAMPLED as the trainer
        # does (mean 0, log_std 0: a = tanh(N(0, 1)) x freedom vector) -- exploration noise on top of the closed
        # integrator loop.  random_actor: the last layer random too (rounds 1-4), state-dependent actions at full scale
        self.init_s = 0.0
        self.policy = BatchedGaussianPolicy(self.layout, last_layer_zero=not random_actor, seed=1234 + rank,
                                            device=device)
        self.policy_name = ("random-init actors, last layer random (--random-actor)" if random_actor else
                            "Xavier-uniform, seed 1234, last layer zero, sampled actions (the reference's policy at the "
                            "start of training, model_rpc.py:10-14,103-106)")
        self.sim = self.env.supervisor.sim
        self.state = None
        self.torch = torch
        from ao_marl_amd import libaomarl
        self.lib = libaomarl
        self.launched, self.local_elapsed = {}, None

    def one_step(self):
        # choose_action + env_step as ONE library call (aomarl_policy_env_step): the launches of policy.select_action
        # followed by env.step, one host round trip
        _, self.state, self.last_r, _, _ = self.env.policy_step(self.policy, self.state)

    def one_step_integrator(self):
        # TrainerRPC.env_step(a=None, linear_control=True): the integrator alone (train_rpc.py:577-578)
        self.state, self.last_r, _, _ = self.env.step(None, linear_control=True)

    def reset(self):
        self.state = self.env.reset()

    def timed(self, steps, warmup, dist=None, backend="nccl", time_frame=True, settle=0, step=None):
        """`settle` untimed steps (timed on their own: self.settle_s), W untimed warm-up steps, then exactly K timed
        steps between barriers + synchronisations.  Returns (elapsed s -- MAX over ranks --, host enqueue s,
        frame-kernel ms per launch from the library's events)."""
        torch = self.torch
        one_step = step or self.one_step
        # no cyclic-garbage collection inside the timed region: a collection that frees an earlier
        # configuration's device buffers (hipFree synchronises the device) would be charged to this one
        gc.collect()
        gc.disable()
        try:
            self.settle_s, self.settle_steps = 0.0, settle
            if settle:
                torch.cuda.synchronize()
                ts = time.perf_counter()
                for _ in range(settle):
                    one_step()
                torch.cuda.synchronize()
                self.settle_s = time.perf_counter() - ts
            for _ in range(warmup):
                one_step()
            torch.cuda.synchronize()
            if time_frame:
                self.sim.set_option("time_frame_kernel", steps)
            self.lib.arith_launches(reset=True)     # `dtype` = what the timed steps launch
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        finally:
            gc.enable()
        self.launched = self.lib.arith_launches()
        self.local_elapsed = elapsed
        fk_ms = None
        if time_frame:
            tot, n = self.sim.frame_kernel_time()
            self.sim.set_option("time_frame_kernel", 0)
            fk_ms = tot / n if n else None
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device=self.device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, t_enq, fk_ms

    def transient_excess(self, steps, elapsed):
        """What the `settle` steps behind the reset took beyond the steady rate of the timed region (s, >= 0): the
        post-reset transient, amortised over the episode like the reset itself."""
        if not self.settle_steps:
            return 0.0
        return max(0.0, self.settle_s - self.settle_steps * elapsed / steps)

    def time_episode(self, episode_len, dist=None, backend="nccl", split_reset=False):
        """One WHOLE episode, measured: reset + episode_len steps between synchronisations (MAX over ranks).
        split_reset: one more synchronisation behind the reset; returns (episode s, reset s)."""
        torch = self.torch
        gc.collect()
        gc.disable()
        try:
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            self.reset()
            t_reset = None
            if split_reset:
                torch.cuda.synchronize()
                t_reset = time.perf_counter() - t0
            for _ in range(episode_len):
                self.one_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        if dist is not None:
            t = torch.tensor([dt, t_reset or 0.0], dtype=torch.float64, device=self.device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, t_reset = float(t[0].item()), (float(t[1].item()) if split_reset else None)
        return (dt, t_reset) if split_reset else dt

    def time_reset(self, dist=None, backend="nccl"):
        """One env.reset() (RlSupervisor.reset: 2n extrusions per layer + the first frame), MAX over ranks."""
        torch = self.torch
        gc.collect()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        self.reset()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt


def amortised(envs_total, steps, elapsed, reset_s, episode_len):
    return envs_total * steps / (elapsed + steps / float(episode_len) * reset_s)


def roofline_block(model, fk_ms, kernel_name, args_pmc, envs, config):
    """The frame kernel against the roof that bounds the instantiation launched: the split-fp16 one is
    HBM-bound (algorithmic bytes / duration against 8 TB/s), the fp32 one compute-bound on the fp32
    matrix pipe (algorithmic flops / duration against 157.3 TFLOP/s; its HBM fraction rides along)."""
    if not fk_ms:
        return None
    gbs = model["bytes"] / (fk_ms * 1e-3) * 1e-9
    tfl = model["flops"] / (fk_ms * 1e-3) * 1e-12
    split = kernel_name.endswith("true>")
    common = {"kernel": kernel_name, "traffic": pmc_traffic(args_pmc, kernel_name, envs, config),
              "avg_launch_ms": fk_ms, "algorithmic_bytes_per_launch": model["bytes"],
              "algorithmic_flops_per_launch": model["flops"],
              "timing": "HIP event pair attached to each k_frame_wave dispatch (start / stop events of "
                        "hipExtLaunchKernelGGL) on its stream, inside the timed region"}
    if split:
        r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
             "algorithmic_tflops": tfl}
    else:
        r = {"bound": "mfma", "achieved": tfl, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
             "frac": tfl / FP32_MFMA_PEAK_TF, "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
        if kernel_name.endswith("false, false, false>") and "executed_matrix_flops" in model:
            # `achieved` keeps the algorithmic count of rounds 1-4 (the reference's transform, pruned: SURVEY 8d); the
            # slopes-only kernel now reaches the same slopes with fewer matrix instructions: what the pipe really does
            ex = model["executed_matrix_flops"] / (fk_ms * 1e-3) * 1e-12
            r["executed_matrix_flops_per_launch"] = model["executed_matrix_flops"]
            r["executed_matrix_tflops"] = ex
            r["executed_matrix_frac"] = ex / FP32_MFMA_PEAK_TF
    r.update(common)
    return r


# the denoiser of one spot image on 16 x 16 x K matrix instructions (csrc/aomarl_denoise.hip): L1 16 tiles x 3
# (K = 9 taps padded to 12), L2 4 x 2 x 9 x 4, L3 1 x 4 x 9 x 8, D1 4 classes x 2 x 4 taps x 16, D2 4 x 4 x 4 x 8
DENOISER_MFMA_PER_IMAGE = 48 + 288 + 288 + 512 + 512
DENOISER_MAC_PER_IMAGE = 1712128


def denoiser_roofline(w, f32, reps=5):
    """The denoiser launch of one step by itself (a copy of the step's spot cube, torch events on the launch stream):
    fp32: matrix-issue bound -- the time its v_mfma_f32_16x16x4_f32 instructions (32 cycles each, one SIMD per wave)
    need on 1024 SIMDs at the clock of the quoted fp32 matrix peak against the launch; fast mode: the duration only."""
    import torch
    sup = w.env.supervisor
    cube = sup.sim.t["bincube"].clone()
    dn = sup.autoencoder
    for _ in range(2):
        dn.denoise_bincube_(cube, f32=f32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dn.denoise_bincube_(cube, f32=f32)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nimg = cube.shape[0] * cube.shape[1]
    ghz = FP32_MFMA_PEAK_TF * 1e12 / (256 * 256.0) * 1e-9        # the clock the fp32 matrix peak is quoted at (2.4 GHz)
    out = {"kernel": "k_denoise4" if f32 else "k_denoise4c", "images": nimg, "ms": ms,
           "tflops": 2.0 * DENOISER_MAC_PER_IMAGE * nimg / (ms * 1e-3) * 1e-12}
    if f32:
        floor_ms = nimg * DENOISER_MFMA_PER_IMAGE * 32 / (1024 * ghz * 1e9) * 1e3
        out.update({"bound": "mfma issue", "matrix_instructions_per_image": DENOISER_MFMA_PER_IMAGE,
                    "matrix_floor_ms": floor_ms, "frac": floor_ms / ms, "clock_ghz": ghz})
    return out


def side_config(config, envs, device, steps, warmup, episode_len, denoiser=None, settle=100, agents=None,
                reset_prefetch=None):
    """A BASELINE configuration other than the headline one, same loop, same accounting (1 GPU): the
    all-fp32 pass is the figure, the split-fp16 pass rides along as `fast_mode`.  reset_prefetch = "same": the
    headline's prefetched reset (the timed steps carry their share of the next reset's rounds, `reset_ms` is the
    adopting reset behind a whole episode)."""
    from ao_marl_amd import libaomarl
    w = Workload(config, envs, 0, 1, device, denoiser=denoiser, agents=agents, reset_prefetch=reset_prefetch)
    rp_on = w.env.supervisor.reset_prefetch is not None
    out = {"workload": config + (" + shipped denoiser" if denoiser else ""), "envs": envs,
           "agents": w.layout.n_agents, "steps": steps, "reset_prefetch": rp_on}
    for mode in ("f32", "split_f16"):
        libaomarl.set_precision(mode)
        try:
            w.reset()
            for _ in range(3):
                w.one_step()                    # (one-time costs of the first episode, see main)
            reset_s = w.time_reset()
            elapsed, _, fk = w.timed(steps, warmup, time_frame=True, settle=settle)
            transient_s = w.transient_excess(steps, elapsed)
            whole = None
            if rp_on:
                w.time_episode(episode_len)                     # (an episode whose prefetch runs to its end)
                ep_s, reset_s = w.time_episode(episode_len, split_reset=True)
                whole = envs * episode_len / ep_s
            rec = {"dtype": libaomarl.dtype_string(w.launched),
                   "value": amortised(envs, steps, elapsed, reset_s + transient_s, episode_len),
                   "value_no_reset": envs * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
                   "reset_ms": reset_s * 1e3, "whole_episode_value": whole, "frame_kernel_ms": fk,
                   "frame_kernel": w.sim.frame_kernel_name(),
                   "mean_strehl_le": float(w.sim.strehl[:, 1].mean())}
            if denoiser:
                w.env.supervisor.autoencoder.check_range()
                rec["denoiser_kernel"] = denoiser_roofline(w, mode == "f32")
        finally:
            libaomarl.set_precision("f32")
        if mode == "f32":
            out.update(rec)
        else:
            out["fast_mode"] = rec
    import torch
    del w
    gc.collect()                # the HIP context of this configuration goes NOW (hipFree synchronises the
    torch.cuda.synchronize()    # device), not at some later collection inside the next timed region
    return out


# ------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                   # does not return

    import datetime
    import numpy as np  # noqa: F401
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    # one rank per GPU over RCCL (backend "nccl").  AOMARL_DIST_BACKEND=gloo lets the multi-rank path
    # be rehearsed on a box with fewer GPUs than ranks (ranks then share devices round-robin).
    backend = os.environ.get("AOMARL_DIST_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if backend == "nccl" and world > ndev:
        raise SystemExit("--gpus %d but only %d GPU(s) visible (AOMARL_DIST_BACKEND=gloo shares devices "
                         "for rehearsals)" % (world, ndev))
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    device = "cuda:%d" % dev_index
    torch.cuda.set_device(device)
    # AOMARL_DIST_FORCE=1: a one-rank run goes through the process group too (the RCCL leg -- barrier, MAX over ranks,
    # the gathers -- executed on a one-GPU box; tests/test_bench_launch.py)
    if world > 1 or os.environ.get("AOMARL_DIST_FORCE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:         # (a one-rank forced process group: any free port, like spawn_ranks)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        tmo = datetime.timedelta(seconds=int(os.environ.get("AOMARL_DIST_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo,
                                    device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    # everything runs on a stream of its own, not on the null stream: launches on the legacy default stream cost the
    # host noticeably more on this runtime (10x10 / 64 environments, host-bound: 0.116 against 0.158 ms per step,
    # profiles/r03_graph_step_probe.txt); the timed regions are bracketed by device-wide synchronisations either way
    main_stream = torch.cuda.Stream(device=torch.device(device))
    torch.cuda.set_stream(main_stream)

    from ao_marl_amd import libaomarl
    from ao_marl_amd.dist import gather_episode_returns
    denoiser = args.denoiser
    if denoiser == "golden":                # the old spelling
        denoiser = "shipped"
    libaomarl.set_precision(args.precision)
    t_init = time.perf_counter()
    w = Workload(args.config, args.envs, rank, world, device, denoiser=denoiser,
                 prefetch=not args.no_prefetch,
                 pipeline=False if args.no_frame_pipeline else (True if args.frame_pipeline_always else "auto"),
                 reset_prefetch=None if (args.no_reset_prefetch or args.no_prefetch) else "same",
                 random_actor=args.random_actor,
                 shortcut=not args.no_residual_shortcut and (not args.no_frame_pipeline or args.residual_shortcut))
    torch.cuda.synchronize()
    init_s = time.perf_counter() - t_init
    env, sim, layout = w.env, w.sim, w.layout
    # (throughput configuration = VecAoEnv.throughput_mode, what sac.train_agent steps with: frame pipeline, residual
    # shortcut -- states within 3e-3 relative of the reference order's asserted, 1.3e-5 measured; the composition itself
    # against the oracle: tests/test_gpu_env_step_large.py `bench` cases --, prefetched reset)
    if args.unfused:
        sim.set_option("force_unfused_frame", 1)
    if args.no_defer:
        sim.defer_shape = False
    if args.graph_step:
        sim.set_option("graph_step", 1)
        w.policy.out_ring = 6                       # stable output addresses for the replays

    w.reset()                                       # (the first reset: VecAoEnv probes both call orders behind it)
    order_probe = env.order_probe
    for _ in range(3):                              # one-time costs of a process's first episode (allocations: 11 ms in
        w.one_step()                                # the first step) are not a per-episode transient
    w.reset()
    rp_on = env.supervisor.reset_prefetch is not None
    # One full reset of this rank's batch in the open, timed (with the reset prefetch on: the rounds the prefetch begun
    # a moment ago has not run yet -- all of them -- are run by this call: the same figure)
    reset_open_s = w.time_reset(dist, backend)
    elapsed, t_enq, fk_ms = w.timed(args.steps, args.warmup, dist, backend, settle=args.settle)
    envs_total = args.envs * world
    transient_s = w.transient_excess(args.steps, elapsed)
    # What a reset costs an episode: in the open, all of it; prefetched, the timed steps already carry their share
    # of the next reset's rounds (dealt out evenly over the episode) and the reset call itself adopts the screens --
    # measured behind a whole episode below (reset_s is replaced by that figure).
    reset_s = reset_open_s
    kernel_name = sim.frame_kernel_name()
    launched = dict(w.launched)
    sr = float(sim.strehl[:, 1].mean())

    # epilogue collective of the path: per-environment returns of every rank, in global seed order
    # (here: the reward of the last step summed over agents, and the long-exposure Strehl)
    ret_all = gather_episode_returns(w.last_r.sum(dim=1))
    sr_all = gather_episode_returns(sim.strehl[:, 1].contiguous())
    from ao_marl_amd import modal as _modal
    shards = [dict(rank=rank, first_seed=int(w.first_seed), envs=args.envs,
                   ms_per_step=w.local_elapsed / args.steps * 1e3, frame_kernel_ms=fk_ms,
                   device=torch.cuda.get_device_name(dev_index),
                   # start-up of this rank: geometry + calibration (memoised on disk under a lock: of N ranks one
                   # calibrates, the others load) + simulator state
                   init_s=init_s, calibration_s=float(getattr(w.env.supervisor, "calibration_seconds", 0.0)),
                   calibration_cache=dict(_modal.cache_stats))]
    if dist is not None:
        got = [None] * world
        dist.all_gather_object(got, shards[0])
        shards = got

    pipe_state = sim.frame_pipeline_state()
    # one whole episode, measured (reset + episode_len steps): what `value` amortises, without the amortisation
    whole = None
    if not args.timed_only and not args.no_whole_episode:
        if rp_on:
            w.time_episode(args.episode_len, dist, backend)      # (an episode whose prefetch runs to its end)
        ep_s, rs = w.time_episode(args.episode_len, dist, backend, split_reset=True)
        whole = {"steps": args.episode_len, "seconds": ep_s, "value": envs_total * args.episode_len / ep_s,
                 "ms_per_step": ep_s / args.episode_len * 1e3, "reset_ms": rs * 1e3,
                 "what": "reset + episode_len steps of the same loop between two synchronisations (one more behind the "
                         "reset), behind the timed region%s" % (" and behind one untimed whole episode: the reset ADOPTS the "
                                                                "screens grown beside that episode" if rp_on else "")}
        if rp_on:
            reset_s = rs
    # (reset prefetch on but no whole episode measured: the prefetched reset's cost is unknown -- amortise the reset in
    # the open, an upper bound, rather than nothing)
    value = amortised(envs_total, args.steps, elapsed, reset_s + transient_s, args.episode_len)
    # the same steps in the plain call order (frame kernel alone on the GPU, the chains behind it): what the
    # pipeline buys, and the frame kernel's duration without the chains' kernels beside it
    plain = None
    do_plain = bool(pipe_state[0])
    if dist is not None:                    # a collective decision: the pass has barriers inside (ranks probe their own
        t = torch.tensor([int(do_plain)], dtype=torch.int32,      # call order and may differ)
                         device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        do_plain = bool(int(t.item()))
    if do_plain and not args.timed_only:
        w.reset()
        sim.set_option("frame_pipeline", 0)
        e_p, _, fk_p = w.timed(min(args.steps, 40), min(args.warmup, 5), dist, backend, settle=args.settle)
        plain = {"ms_per_step_no_reset": e_p / min(args.steps, 40) * 1e3, "frame_kernel_ms": fk_p}
        w.reset()
        sim.set_option("frame_pipeline", 1)
        if not (plain["frame_kernel_ms"] and pipe_state[0]):
            plain = None                    # (this rank ran in the plain order anyway)
    # the integrator alone (a=None, linear_control=True: train_rpc.py:577-578; the reference's evaluation baseline and
    # its normalisation runs), plain call order (the call-by-call step), from a reset: a closed loop's Strehl beside
    # the policy's
    integ = None
    if not args.timed_only and denoiser is None:
        if pipe_state[0]:
            sim.set_option("frame_pipeline", 0)
        try:
            w.reset()
            k_i = min(args.steps, 40)
            e_i, _, fk_i = w.timed(k_i, min(args.warmup, 5), dist, backend, settle=args.settle, step=w.one_step_integrator)
            integ = {"ms_per_step_no_reset": e_i / k_i * 1e3, "value_no_reset": envs_total * k_i / e_i, "steps": k_i,
                     "frame_kernel_ms": fk_i, "mean_strehl_le": float(sim.strehl[:, 1].mean()),
                     "mean_strehl_se": float(sim.strehl[:, 0].mean()),
                     "frames_behind_reset": args.settle + min(args.warmup, 5) + k_i,
                     "what": "TrainerRPC.env_step(a=None, linear_control=True): no actor, no rl_control; call-by-call "
                             "(plain order); Strehl: long exposure over all frames since the reset (the closing "
                             "transient included) and the last frame's short exposure"}
        except Exception as e:                      # side figure
            integ = {"error": str(e)[:200]}
        finally:
            w.reset()
            if pipe_state[0]:
                sim.set_option("frame_pipeline", 1)
    # side figure: the same whole episode with the NEXT reset's rounds hidden beside it (single-GPU runs)
    rp_side = None
    if whole is not None and not rp_on and dist is None and not args.no_prefetch:
        try:
            env.supervisor.reset_prefetch = "same"
            w.time_episode(args.episode_len)             # (begins the prefetch, runs it to its end)
            ep2, rs2 = w.time_episode(args.episode_len, split_reset=True)
            rp_side = {"whole_episode_value": envs_total * args.episode_len / ep2, "ms_per_step": ep2 / args.episode_len * 1e3,
                       "reset_ms": rs2 * 1e3, "prefetched_resets": int(getattr(sim, "prefetched_resets", 0))}
        finally:
            env.supervisor.reset_prefetch = None
            w.reset()
    # diagnostic pass (outside `value`): every stage with its own event pair (call by call: behind a reset
    # when the timed steps left a pipelined frame in flight)
    if pipe_state[0] and args.timed_only:
        w.reset()
    stage_diag = None if args.timed_only else stage_split(w, min(20, args.steps))

    out = None
    if rank == 0:
        s = env.supervisor.s
        model = frame_kernel_model(s, args.envs)
        roof = roofline_block(model, fk_ms, kernel_name, args.pmc, args.envs, args.config)
        if roof is not None and pipe_state[0] and os.environ.get("AOMARL_FW_LDS_PAD") != "0":
            roof["in_the_timed_region"] = ("pipelined launches of this instantiation ask for 54.7 KB of LDS: TWO frame workgroups per "
                                           "CU (three when the kernel runs alone) so that a workgroup of the chains' products or "
                                           "actors fits beside them at any time -- the frame kernel 0.39 -> 0.44 ms, the step 0.482 "
                                           "-> 0.473 ms (profiles/r06_overlap_experiments.txt); `alone` is the kernel by itself")
        if plain is not None and plain["frame_kernel_ms"]:
            # `achieved` / `frac` above: the launches of the timed region, which share the GPU with the kernels of
            # the control / agent and extrusion chains (the frame pipeline).  The kernel by itself:
            alone = roofline_block(model, plain["frame_kernel_ms"], kernel_name, args.pmc, args.envs, args.config)
            roof["alone"] = {"avg_launch_ms": alone["avg_launch_ms"], "achieved": alone["achieved"], "frac": alone["frac"],
                             "how": "the same launches in the plain call order (no other kernel beside the frame kernel), "
                                    "%d steps behind the timed region" % min(args.steps, 40)}
        out = {
            "metric": "env steps/sec (AO frames/sec)", "value": value, "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": args.envs * world / value * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": libaomarl.dtype_string(launched),
            "data": "synthetic",
            "config": {"workload": args.config + (" + denoiser" if denoiser else ""),
                       "envs_per_gpu": args.envs, "agents": layout.n_agents,
                       "state_dims": layout.state_shapes()[:1] + layout.state_shapes()[-1:],
                       "action_dim": layout.action_dim, "episode_len": args.episode_len,
                       "policy": w.policy_name,
                       "parallelism": "independent env shards x%d" % world},
            "value_no_reset": envs_total * args.steps / elapsed,
            "ms_per_step_no_reset": elapsed / args.steps * 1e3,
            "reset_ms": reset_s * 1e3,
            "reset_in_the_open_ms": reset_open_s * 1e3,
            "reset_prefetch": {"on": rp_on, "prefetched_resets": int(getattr(sim, "prefetched_resets", 0)), "side_figure": rp_side,
                               "what": "the next episode's 2 x 648 extrusion rounds per layer run in a shadow state beside this "
                                       "episode's steps (dealt out evenly: the timed steps carry their share), reset() adopts the "
                                       "screens: bit-identical to the reset in the open (aomarl_reset_prefetch_*)"},
            "post_reset_transient_ms": transient_s * 1e3,
            "episode_position": {"settle": args.settle, "warmup": args.warmup,
                                 "what": "the timed steps are frames [settle + warmup, settle + warmup + steps) of an episode; "
                                         "`value` = steps / (timed seconds + steps / episode_len x (reset + transient excess "
                                         "of the first `settle` frames)); `whole_episode` is the same thing measured"},
            "whole_episode": whole,
            "roofline": roof,
            "launched": {k: v for k, v in launched.items() if v},
            "stage_ms": stage_diag, "atmos_prefetch": bool(env.supervisor.prefetch_atmos),
            "frame_pipeline": {"on": bool(pipe_state[0]), "pipelined_steps": pipe_state[2], "moves_beside_frame": pipe_state[3],
                               "plain_order_ms_per_step_no_reset": plain["ms_per_step_no_reset"] if plain else None,
                               "order_probe_ms_per_step": order_probe,
                               "what": "frame t+1 launched before frame t is reduced (loop delay = 1 frame): same kernels, "
                                       "same values, frame kernels back to back with the control / agent and extrusion "
                                       "chains beside them (aomarl_set_frame_pipeline)"},
            "residual_shortcut": bool(env.residual_shortcut),
            "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "mean_strehl_le": sr,
            "integrator_only": integ,
            "gathered": {"n": int(ret_all.numel()), "mean_last_step_reward": float(ret_all.mean()),
                         "mean_strehl_le": float(sr_all.mean())},
            "shards": shards,
        }
    if dist is not None:
        dist.barrier()

    # ---- everything below is single-GPU side information (rank 0 of an N = 1 run)
    if rank == 0 and world == 1:
        main_is_headline = args.config == WORKLOAD and not denoiser
        if denoiser:
            out["denoiser_kernel"] = denoiser_roofline(w, args.precision == "f32")
        if not args.no_side_configs and main_is_headline and args.precision == "f32":
            try:        # the same loop in the fast mode: split-fp16 operand pairs in the DFTs and the GEMMs
                libaomarl.set_precision("split_f16")
                w.reset()
                rs16 = w.time_reset()
                e16, _, fk16 = w.timed(args.steps, args.warmup, settle=args.settle)
                if rp_on:       # (the prefetched reset, like the headline: what an adopting reset costs behind a whole episode)
                    w.time_episode(args.episode_len)
                    _, rs16 = w.time_episode(args.episode_len, split_reset=True)
                out["fast_mode"] = {"dtype": libaomarl.dtype_string(w.launched),
                                    "value": amortised(args.envs, args.steps, e16, rs16 + w.transient_excess(args.steps, e16), args.episode_len),
                                    "value_no_reset": args.envs * args.steps / e16,
                                    "ms_per_step_no_reset": e16 / args.steps * 1e3, "reset_ms": rs16 * 1e3,
                                    "launched": {k: v for k, v in w.launched.items() if v},
                                    "roofline": roofline_block(model, fk16, sim.frame_kernel_name(), args.pmc,
                                                               args.envs, args.config)}
            except Exception as e:
                out["fast_mode"] = {"error": str(e)[:200]}
            finally:
                libaomarl.set_precision("f32")
        try:
            out["sac_update"] = sac_update_rate(layout, device) if main_is_headline else None
        except Exception as e:                      # secondary figure: never fail the bench line
            out["sac_update"] = {"error": str(e)[:200]}
        if main_is_headline and not args.no_episode_end_to_end and not args.timed_only:
            try:
                out["episode_end_to_end"] = episode_end_to_end(env, layout, device, args.episode_len)
            except Exception as e:
                out["episode_end_to_end"] = {"error": str(e)[:300]}
        s_main = env.supervisor.s
        del w, env, sim
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if not args.no_side_configs and main_is_headline:
            out["configs"] = {}
            # (+ the configuration the reference publishes, README.md:116-119: `_d1_noise` + the shipped autoencoder,
            # 43 agents = 42 x 30 modes + tip-tilt)
            for key, cfg, ne, dn, st, ag in (("configs[1]", SMALL, 64, None, 200, None),
                                             ("configs[4]", NOISY, args.envs, "shipped", 30, None),
                                             ("published_43_agents", NOISY.replace("_d0_", "_d1_"), args.envs, "shipped", 30, 43)):
                try:
                    # (configs[1]: a reset in the open is 7 % of its episode -- prefetched like the headline's; the
                    # noisy configurations' is 0.6 %: in the open)
                    out["configs"][key] = side_config(cfg, ne, device, st, 40 if cfg == SMALL else 5, args.episode_len, dn,
                                                      agents=ag, reset_prefetch="same" if cfg == SMALL and rp_on else None)
                except Exception as e:
                    out["configs"][key] = {"error": str(e)[:300]}
                torch.cuda.empty_cache()
        out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(s_main)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def stage_split(w, steps):
    """Per-stage durations (ms) from event pairs around the supervisor's stage entry points; a
    diagnostic pass AFTER the timed region (sixteen event records per step cost 7 % of a step)."""
    import numpy as np
    torch = w.torch
    sim, pairs = w.sim, {}
    names = (("move_atmos", "move_atmos"), ("target_psf", "target_psf"), ("comp_image", "wfs_spot_cog"),
             ("frame_fused", "frame_fused"), ("do_control", "do_control"),
             ("slopes2modes", "residual_modes"), ("rl_control", "rl_control"),
             ("apply_control", "dm_shape"), ("comp_strehl", "strehl_commit"))
    saved = {}
    for name, label in names:
        fn = getattr(sim, name)
        saved[name] = fn

        def timed(*a, _fn=fn, _label=label, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = _fn(*a, **k)
            e1.record()
            pairs.setdefault(_label, []).append((e0, e1))
            return r
        setattr(sim, name, timed)
    sup = w.env.supervisor
    native, split = w.env.native_step, sup.next_part_one_split
    w.env.native_step, sup.next_part_one_split = False, True     # stage by stage (same kernels, unfused tail)
    for _ in range(steps):
        w.one_step()
    torch.cuda.synchronize()
    w.env.native_step, sup.next_part_one_split = native, split
    for name in saved:
        delattr(sim, name)                  # back to the class methods
    return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in pairs.items()}


def pmc_traffic(fname, kernel_name, envs, config):
    """HBM bytes per launch of the frame kernel from the rocprofv3 --pmc passes summarised in
    profiles/<fname> (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md; collected by
    tools/fw_pmc.sh + tools/fw_pmc.py).  None unless the file was measured on the kernel instantiation and the
    configuration this run launched."""
    try:
        if not fname:
            import glob
            import re
            cands = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_frame_kernel.json"))
            if not cands:
                return None
            tag = lambda f: re.match(r"r(\d+)([a-z]*)_", os.path.basename(f))      # noqa: E731
            fname = os.path.basename(max(cands, key=lambda f: (int(tag(f).group(1)), tag(f).group(2))))
        pmc = json.load(open(os.path.join(ROOT, "profiles", fname)))
        if pmc.get("_config") != config:
            return None
        recs = pmc.get("kernels") or {pmc["frame_fused"]["kernel"]: pmc["frame_fused"]}
        rec = recs.get(kernel_name)
        if rec is None:
            return None
        return rec["hbm_traffic_bytes_per_launch"] * envs / pmc["_envs"]
    except Exception:
        return None


if __name__ == "__main__":
    main()
