#!/usr/bin/env python3
"""Headline benchmark: env steps/sec (AO frames/sec) of the per-timestep hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic environments per GPU:
    batched 14-agent SAC actor forward  ->  next_part_two (Btt correction, delay, DM shapes,
    Strehl)  ->  per-agent rewards  ->  next_part_one (phase-screen extrusion, target trace + PSF,
    WFS trace + spot images + COG, integrator)  ->  state assembly.
Workload (BASELINE.json configs[2]): production_sh_40x40_8m_3layers, 256 atmosphere seeds per
GPU, 14 agents (13 x 98 Btt modes + tip-tilt), windowed states (w = 20).  Environments are
independent, so N GPUs run N x 256 seeds with no data-path collective (weak scaling); the only
collective is the MAX over ranks of the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step, timed live
with HIP events on the launch stream; `cpu_baseline` is the CPU oracle (a C restatement of the
COMPASS frame the reference drives, oracle/aoref.c) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOAD = "production_sh_40x40_8m_3layers"
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak


def stage_models(s, nenv, nmodes, nact):
    """Algorithmic bytes / flops per launch of each stage (DESIGN.md section 4 derives them)."""
    n2, p2 = s.n * s.n, s.pupdiam * s.pupdiam
    nl = s.nscreens
    shape_px = sum(d.dim * d.dim for d in s.dms)
    # A4+A3+A5 fused: phase of every layer + DM planes in, 2 slopes per sub-aperture out;
    # flops: pruned radix-2 FFT count for 16 non-zero rows -> 32x32 kept outputs of a 64^2 grid
    fft_flops = 16 * 5 * 64 * 6 + 32 * 5 * 64 * 6
    spot = dict(bound="mfma", unit="TFLOP/s", peak=FP32_MFMA_PEAK_TF,
                work=nenv * s.nvalid * float(fft_flops + 256 * 20),
                bytes=nenv * s.nvalid * 2048.0)
    dm = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
              work=nenv * (s.nactu * 4.0 + shape_px * 4.0))
    tgt = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
               work=nenv * ((nl + len(s.dms)) * p2 * 4.0))
    # one-pass frame kernel: every lit 16x16 tile of the pupil grid is read once (layers + stack
    # array) and feeds the spot DFT (valid sub-apertures) and the 16-column PSF row DFT (16 kx x
    # 256 pixels x 8 flop); the tip-tilt planes and the mask are shared by all environments
    lit = int((s.spupil.reshape(s.pupdiam // 16, 16, s.pupdiam // 16, 16).sum(axis=(1, 3)) > 0).sum()) \
        if s.pupdiam % 16 == 0 else 0
    # with both DFTs on split-fp16 MFMAs the matrix roof is an order of magnitude away; the kernel
    # is bounded by getting the phase in: roofline against HBM (algorithmic bytes), the flop view is
    # kept in `image_kernel`
    fused = dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
                 work=nenv * (lit * nl * 1024.0 + s.nvalid * 8.0 + s.pupdiam * 16 * 8.0),
                 flops=nenv * (s.nvalid * float(fft_flops + 256 * 20) + lit * 256 * 16 * 8.0),
                 bytes=nenv * (lit * nl * 1024.0 + s.nvalid * 8.0 + s.pupdiam * 16 * 8.0))
    return {"wfs_spot_cog": spot, "dm_shape": dm, "target_psf": tgt, "frame_fused": fused}


class StageTimer(object):
    """HIP events around stage entry points.  Only the labels in `live` are timed (an event pair is
    not free: sixteen of them per step cost 7 % of the step); the timed region keeps the image
    kernel's pair, the per-stage split comes from a short diagnostic pass after it."""

    def __init__(self):
        self.pairs = {}
        self.live = None                   # None: every wrapped stage

    def wrap(self, obj, name, label):
        fn = getattr(obj, name)

        def timed(*a, **k):
            if self.live is not None and label not in self.live:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.pairs.setdefault(label, []).append((e0, e1))
            return r

        setattr(obj, name, timed)

    def mean_ms(self):
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.pairs.items()}

    def clear(self):
        self.pairs = {}


def cpu_baseline(env, budget_s=15.0):
    """The CPU oracle stepping ONE environment of the same configuration (same calibrated
    command matrix) on this host; bounded to ~budget_s seconds of frames."""
    from oracle import aoref
    s = env.supervisor.s

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
    o.next_part_two(None)
    o.next_part_one()                   # warm caches / OpenMP pool
    t0, frames = time.time(), 0
    while True:
        o.next_part_two(None)
        o.next_part_one()
        frames += 1
        dt = time.time() - t0
        if dt > budget_s or frames >= 200:
            break
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": frames / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d integrator frames of 1 environment of %s (extrusion, 2 raytraces, "
                      "1200 zero-padded 64x64 FFT spots, COG, cmat GEMV, delay, DM shapes, "
                      "2048^2 FFT PSF) in %.1f s, OpenMP over rows/sub-apertures; SAC actor not "
                      "included" % (frames, s.name, dt)}


def sac_update_rate(layout, device, n_updates=100, batch=256, rows=20000):
    """Secondary figure (not `value`): wall time of one aomarl_sac_update of every agent on a batch of
    256 replay rows per agent (the learner side of the path)."""
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
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_updates * 1e3
    return {"ms_per_update": ms, "agents": layout.n_agents, "batch_per_agent": batch,
            "updates_per_s": 1e3 / ms, "kernel": "aomarl_sac_update"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs", type=int, default=256, help="environments per GPU")
    ap.add_argument("--config", default=WORKLOAD)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="separate science / WFS passes instead of the one-pass frame kernel")
    ap.add_argument("--denoiser", default=None,
                    help="WFS-image denoiser weights (a state_dict file, or 'golden' for the shipped "
                         "network kept in tests/golden/host_denoiser.pt): BASELINE configs[4]")
    ap.add_argument("--residual-shortcut", action="store_true",
                    help="residual modes from one product with v2m.cmat instead of do_control + "
                         "volts2modes (VecAoEnv.residual_shortcut; off in the product default)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="move the atmosphere in front of the image kernels (no side stream)")
    ap.add_argument("--no-defer", action="store_true",
                    help="materialise the stack-array DM shapes instead of evaluating them from the "
                         "voltages inside the frame kernel")
    ap.add_argument("--pmc", default="r01g_pmc_counters_256env.json",
                    help="profiles/<file> with the HBM bytes per launch from rocprofv3 --pmc passes")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    # one rank per GPU over RCCL (backend "nccl").  AOMARL_DIST_BACKEND=gloo lets the multi-rank path
    # be rehearsed on a box with fewer GPUs than ranks (ranks then share devices round-robin).
    backend = os.environ.get("AOMARL_DIST_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    device = "cuda:%d" % dev_index
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ao_marl_amd.agents import BatchedGaussianPolicy
    from ao_marl_amd.env import VecAoEnv

    small = "10x10" in args.config
    if small:
        rl = dict(n_zernike_start_end=[0, 80], n_reverse_filtered_from_cmat=5)
        n_modal = 1
    else:
        rl = dict(n_zernike_start_end=[0, 1274], n_reverse_filtered_from_cmat=5,
                  window_n_zernike=20, include_tip_tilt_windowed=True)
        n_modal = 13
    # independent shards: rank r owns seeds [r*envs, (r+1)*envs) of the global seed sequence
    autoencoder = None
    if args.denoiser:
        from ao_marl_amd.denoiser import SubapDenoiser
        path = os.path.join(ROOT, "tests", "golden", "host_denoiser.pt") if args.denoiser == "golden" \
            else args.denoiser
        sd = torch.load(path, map_location="cpu", weights_only=True)
        autoencoder = SubapDenoiser(sd.get("state_dict", sd), device=device)
    norm_kw = {}
    try:
        from ao_marl_amd.env import load_norm
        load_norm(args.config)
    except FileNotFoundError:
        # no recorded statistics for this configuration: borrow the closest one's (synthetic
        # benchmark data; the arithmetic per step is identical)
        nrm, zn = load_norm("production_sh_40x40_8m_3layers_d1_noise" if "noise" in args.config
                            else WORKLOAD)
        norm_kw = dict(norm=nrm, zn_norm=zn)
    env = VecAoEnv(args.config, args.envs, rl, initial_seed=1234 + 16 * args.envs * rank,
                   seed_stride=16, n_agents_modal=n_modal, device=device, autoencoder=autoencoder,
                   prefetch_atmos=not args.no_prefetch, **norm_kw)
    env.residual_shortcut = bool(args.residual_shortcut)
    layout = env.layout
    policy = BatchedGaussianPolicy(layout, last_layer_zero=False, seed=1234 + rank, device=device)
    sim = env.supervisor.sim

    timer = StageTimer()
    # stage-by-stage call order of the supervisor (same kernels as the composite entry point), so
    # that the image kernel can be bracketed by its own event pair
    env.supervisor.next_part_one_split = True

    if args.unfused:
        sim.set_option("force_unfused_frame", 1)
    if args.no_defer:
        sim.defer_shape = False

    for name, label in (("move_atmos", "move_atmos"), ("target_psf", "target_psf"),
                        ("comp_image", "wfs_spot_cog"), ("frame_fused", "frame_fused"),
                        ("do_control", "do_control"), ("slopes2modes", "residual_modes"),
                        ("rl_control", "rl_control"), ("apply_control", "dm_shape"),
                        ("comp_strehl", "strehl_commit")):
        timer.wrap(sim, name, label)

    state = env.reset()

    def one_step(st):
        a, _ = policy.select_action(st)
        s_next, r, _, _ = env.step(a)
        return s_next

    # the timed region carries the event pair of the image kernel(s) only (-> roofline)
    timer.live = {"frame_fused", "wfs_spot_cog", "target_psf"}
    for _ in range(args.warmup):
        state = one_step(state)
    timer.clear()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = one_step(state)
    t_enq = time.perf_counter() - t0           # host time to enqueue the K steps (diagnostic)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage_ms = timer.mean_ms()
    sr = sim.strehl[:, 1].mean().item()
    # diagnostic pass (outside `value`): every stage with its own event pair
    timer.clear()
    timer.live = None
    for _ in range(min(20, args.steps)):
        state = one_step(state)
    torch.cuda.synchronize()
    stage_diag = timer.mean_ms()
    stage_diag.update(stage_ms)            # the image kernel keeps its timed-region figure
    if rank == 0:
        models = stage_models(env.supervisor.s, args.envs, env.nmodes, layout.action_dim)
        if not any(k in models for k in stage_ms):       # denoiser run: no per-stage split
            stage_ms["frame_fused"] = float("nan")
        dom = max((k for k in stage_ms if k in models), key=lambda k: (stage_ms[k] == stage_ms[k], stage_ms[k]))
        m, ms = models[dom], stage_ms[dom]
        scale = 1e-12 if m["unit"] == "TFLOP/s" else 1e-9
        achieved = m["work"] / (ms * 1e-3) * scale
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
        # MI355X_MICROARCH.md), measured off-line at 256 envs and scaled to this batch
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", args.pmc)))
            if dom in pmc and args.config == WORKLOAD:
                traffic = pmc[dom]["hbm_traffic_bytes_per_launch"] * args.envs / pmc["_envs"]
        except Exception:
            traffic = None
        roof = {"kernel": dom, "bound": m["bound"], "achieved": achieved, "peak": m["peak"],
                "unit": m["unit"], "frac": achieved / m["peak"], "traffic": traffic,
                "avg_launch_ms": ms}
        img = "frame_fused" if "frame_fused" in stage_ms else "wfs_spot_cog"
        spot_ms = stage_ms[img]
        sp = models[img]
        out = {
            "metric": "env steps/sec (AO frames/sec)", "value": args.envs * world * args.steps / elapsed,
            "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "envs_per_gpu": args.envs,
                       "agents": layout.n_agents, "state_dims": layout.state_shapes()[:1] +
                       layout.state_shapes()[-1:], "action_dim": layout.action_dim,
                       "parallelism": "independent env shards x%d" % world},
            "roofline": roof,
            "image_kernel": {"kernel": img, "avg_launch_ms": spot_ms,
                            "algorithmic_tflops": sp.get("flops", sp["work"]) / (spot_ms * 1e-3) * 1e-12,
                            "frac_fp32_mfma_peak": sp.get("flops", sp["work"]) / (spot_ms * 1e-3) * 1e-12 / FP32_MFMA_PEAK_TF,
                            "algorithmic_gbs": sp["bytes"] / (spot_ms * 1e-3) * 1e-9,
                            "frac_hbm_peak": sp["bytes"] / (spot_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS},
            "stage_ms": stage_diag, "atmos_prefetch": bool(env.supervisor.prefetch_atmos),
            "host_enqueue_ms_per_step": t_enq / args.steps * 1e3,
            "mean_strehl_le": sr,
        }
        try:
            out["sac_update"] = sac_update_rate(layout, device) if (world == 1 and args.config == WORKLOAD) else None
        except Exception as e:                      # secondary figure: never fail the bench line
            out["sac_update"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(env)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
