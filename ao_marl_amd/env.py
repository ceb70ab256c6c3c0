"""Vectorised AO environment: the reference's RlSupervisor + AoEnv + the environment half of
TrainerRPC.env_step, for `nenv` independent atmosphere seeds resident on one GPU.

Mirrors (same names, argument meaning, ordering, error behaviour):
  RlSupervisor.reset / next_part_one / next_part_two / rl_control    rlSupervisor.py:236-246,
                                                                      1015-1051, 900-947, 713-733
  AoEnv.reset / linear_step / rl_step / calculate_reward             ao_env.py:316-359, 871-939,
                                                                      585-860
  TrainerRPC.env_step / divide_rewards_for_agents                    train_rpc.py:633-648, 402-416
Every per-frame arithmetic step runs in libaomarl_hip.so (ao_marl_amd/sim.py); this file is
sequencing + state bookkeeping on device tensors.  States are float32 (the reference concatenates
float64 NumPy vectors, ao_env.py:909).
"""
import ctypes
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import geometry as G
from . import modal, params, system
from .agents import AgentLayout
from .sim import HipSim

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# defaults of the reference's RL configuration (config/parameters.cfg:6-46, GlobalConfig.py:86-131)
DEFAULT_ENV_RL = dict(
        reward_type="avg_squared_modes_1000", max_steps_per_episode=1000, basis="zernike_space",
        level="correction", state_dm_before_linear=True, state_dm_after_linear=False,
        state_wfs=False, state_dm_residual=True, number_of_previous_dm=2,
        number_of_previous_wfs=0, number_of_previous_dm_residuals=0, include_tip_tilt=True,
        normalization_std_inside_environment=1.0, normalization_mean_inside_environment=0.0,
        norm_scale_zernike_actions=10.0, modification_online=False,
        n_zernike_start_end=[-1, -1], n_reverse_filtered_from_cmat=0, window_n_zernike=-1,
        include_tip_tilt_windowed=False, tt_treated_as_mode=False, delayed_assignment=1)


def load_norm(name):
    """Normalisation statistics + action bounds recorded by the reference from real COMPASS."""
    if name.endswith(".py"):
        name = name[:-3]
    path = os.path.join(DATA_DIR, "norm_%s.npz" % name)
    if not os.path.exists(path):
        raise FileNotFoundError(
                "no normalisation data for %r (%s); generate it with "
                "VecAoEnv.obtain_normalization() or tools/import_norm_data.py" % (name, path))
    z = np.load(path)
    norm = {k: {st: z["%s_%s" % (k, st)] for st in ("mean", "std", "max", "min")}
            for k in ("wfs", "dm", "dm_residual")}
    return norm, z["zn_norm"].copy()


class VecAtmos(object):
    """AtmosCompass's run-time surface for the whole batch: set_wind / set_r0 (atmosCompass.py:79-135), what the
    trainer's non-stationary experiments call between episodes (train_rpc.py:429-450).  The new values hold for every
    move PLANNED after the call.  A move that is already issued -- the next frame's, under prefetch_atmos or with a
    frame in flight -- keeps the atmosphere it was planned with: the reference would have moved it with the new one,
    so stepping on from there raises; call these right before reset() (the trainer's use), or build the environment
    with prefetch_atmos=False to change the atmosphere in mid-episode."""

    def __init__(self, sup):
        self._sup = sup
        a = sup.config.p_atmos
        self.windspeed = np.array(a.windspeed, dtype=np.float32)      # (PATMOS.py: float32 arrays)
        self.winddir = np.array(a.winddir, dtype=np.float32)
        self.r0 = float(a.r0)

    def _changed(self, blocked, dropped):
        sup = self._sup
        if blocked:
            sup._atmos_changed_behind = blocked
        if dropped:
            sup._rp_left = 0                       # the prefetched reset was grown with the old atmosphere: the next reset runs in the open

    def deltas(self, screen_index):
        """(deltax, deltay) in pixels per frame of a layer from its wind speed and direction (atmos_init.py:99-102,
        atmosCompass.py:118-123)."""
        # (the expression of geometry.atmos_geometry, same types in the same order: a layer set to the wind of a
        # parameter file moves exactly like the layer of a system built from that file)
        sup, ps = self._sup, self._sup.config
        cz = np.cos(ps.p_geom.zenithangle * G.DEG2RAD)
        lin_delta = sup.sysm.geom.pupdiam / ps.p_tel.diam * self.windspeed * cz * ps.p_loop.ittime
        deltax = np.asarray(lin_delta * np.sin(G.DEG2RAD * self.winddir + np.pi), dtype=np.float32)
        deltay = np.asarray(lin_delta * np.cos(G.DEG2RAD * self.winddir + np.pi), dtype=np.float32)
        return deltax[screen_index], deltay[screen_index]

    def set_wind(self, screen_index, *, windspeed=None, winddir=None):
        """atmosCompass.py:103-135: new speed [m/s] and / or direction [deg] of one layer, every environment."""
        sup = self._sup
        k = int(screen_index)
        if not 0 <= k < sup.s.nscreens:
            raise IndexError("screen_index %d of %d" % (k, sup.s.nscreens))
        if windspeed is not None:
            self.windspeed[k] = float(windspeed)
        if winddir is not None:
            self.winddir[k] = float(winddir)
        dx, dy = self.deltas(k)
        blocked = sup.sim.atmos_change_blocked()
        dropped = sup.sim.set_wind(k, dx, dy)
        sup.config.p_atmos.windspeed[k], sup.config.p_atmos.winddir[k] = self.windspeed[k], self.winddir[k]
        self._changed(blocked, dropped)

    def amplitudes(self, r0):
        """Per-layer noise amplitude [um] of the extrusion for a global r0 @ 0.5 um (atmos_init.py:115,
        iterkolmo.py:278; ao_marl_amd/system.py)."""
        sup = self._sup
        atm = sup.sysm.atm
        fr = np.asarray(sup.config.p_atmos.frac)
        frac = fr / np.sum(fr)                              # geometry.atmos_geometry, same types in the same order
        r0_layers = np.asarray(r0 / (frac**(3. / 5.) * atm.pupixsize), dtype=np.float32)
        return (r0_layers.astype(np.float64)**(-5. / 6.) * 0.5 / (2 * np.pi)).astype(np.float32)

    def set_r0(self, r0, *, reset_seed=-1):
        """atmosCompass.py:79-101: r0 @ 0.5 um for all layers; the screens as they stand are kept (reset_seed = -1)."""
        if reset_seed != -1:
            raise NotImplementedError("set_r0(reset_seed=...): re-seeding belongs to the episode here -- "
                                      "set_sim_seed(seed) and reset() (every environment owns a block of seeds)")
        sup = self._sup
        blocked = sup.sim.atmos_change_blocked()
        dropped = sup.sim.set_amplitudes(self.amplitudes(r0))
        self.r0 = float(r0)
        sup.config.p_atmos.r0 = float(r0)
        self._changed(blocked, dropped)


class VecRlSupervisor(object):
    """Batched counterpart of shesha's RlSupervisor for the integrator (+RL correction) path."""

    def __init__(self, config, config_rl, nenv, *, initial_seed=1234, seed_stride=16,
                 device="cuda:0", strehl_halfwin=8, keep_bincube=False, sim_factory=None,
                 autoencoder=None, geo=False, prefetch_atmos=True):
        # autoencoder: a denoiser.SubapDenoiser (rlSupervisor.py:147, :977-978) or None
        self.autoencoder = autoencoder
        self.config = config if not isinstance(config, str) else params.builtin(config)
        self.config_rl = dict(DEFAULT_ENV_RL)
        self.config_rl.update(config_rl or {})
        if self.config_rl["level"] != "correction" or self.config_rl["basis"] != "zernike_space":
            raise NotImplementedError                       # rlSupervisor.py:728-731, 831-834
        # modification_online = the reference's `pure_delay_0` (rlSupervisor.py:145): the science target is ray-traced
        # behind apply_control in next_part_two (:938-939) -- it sees THIS frame's atmosphere with the command just
        # applied -- and not in next_part_one (:964-965).  Call-by-call order, screens on the current frame.
        # (the geometric controller's path, next_part_one_geo, is the same in both orders: rlSupervisor.py:989-1013 does
        # not read pure_delay_0; the denoiser sits between image formation and centroiding in both, :975-984)
        self.pure_delay_0 = bool(self.config_rl["modification_online"])
        if self.pure_delay_0:
            prefetch_atmos = False
        self.nenv, self.device = nenv, torch.device(device)
        self.sysm = G.build_system(self.config)
        self.s = system.from_system(self.sysm, ncontrol=0, strehl_halfwin=strehl_halfwin)
        # `sim_factory(s, nenv, device=..., **kw)` builds the simulator; the product default is
        # the HIP one.  (tests/ inject a CPU-oracle-backed stand-in to pin this file's host logic
        # against the reference's traces on a GPU-less box.)
        make = sim_factory if sim_factory is not None else HipSim
        # calibration through the backend (imat_geom, correct_dm, imat, Btt, filtered cmat)
        # (memoised per geometry / nfilt / backend arithmetic, in the process and on disk: modal.calibrate; the
        # calibration simulator is only built when the calibration really runs)
        self.n_reverse_filtered_from_cmat = int(self.config_rl["n_reverse_filtered_from_cmat"])
        t_cal = time.perf_counter()
        if sim_factory is None:
            from .sim import hip_calibration_id
            cal_id = hip_calibration_id()
        else:
            cal_id = getattr(sim_factory, "calibration_id", None)
            cal_id = cal_id() if callable(cal_id) else None
        self.cal = modal.calibrate(self.s, self.sysm,
                                   lambda: make(self.s, nenv=min(512, 2048), device=device, keep_phase=True),
                                   nfilt=max(self.n_reverse_filtered_from_cmat, 0), backend_id=cal_id)
        self.calibration_seconds = time.perf_counter() - t_cal
        self.modes2volts, self.volts2modes = self.cal.modes2volts, self.cal.volts2modes
        self.nmodes = self.volts2modes.shape[0]
        self.n_modes_start_end = list(self.config_rl["n_zernike_start_end"])
        self.include_tip_tilt = bool(self.config_rl["include_tip_tilt"])
        self.sim = make(self.s, nenv=nenv, device=device, keep_bincube=keep_bincube)
        if autoencoder is not None and hasattr(autoencoder, "set_input_bound"):
            # brightest possible pixel: every photon of the best-lit sub-aperture in one pixel, plus
            # six sigma of photon and read-out noise -> picks the denoiser kernel (fp16 pairs / fp32)
            tot = float(self.s.nphot) * float(np.max(self.s.flux))
            autoencoder.set_input_bound(tot + 6.0 * np.sqrt(max(tot, 0.0)) + 6.0 * max(float(self.s.noise), 0.0))
        self.freedom_vector = None
        self.gain, self._env_gains = float(self.s.gain), False
        # next_part_one(defer_control=True) images a frame without do_control; the product with the
        # command matrix then runs only if somebody needs err / com in actuator space (see
        # VecAoEnv._linear_step_fused): _control_pending = slopes measured, integrator not yet run;
        # _err_stale = the command was rebuilt from modal coordinates, err never formed
        self._control_pending, self._err_stale, self._s2m_ok = False, False, False
        self.next_part_one_split = False        # True: always the stage-by-stage call order (bench.py times the stages)
        self.last_modes = None
        self._push_modal()
        # controller 1 of the non-noise parameter files: the geometric reference controller with
        # its own DM pair and target (rlSupervisor.py:989-1013); off by default -- it changes
        # nothing controller 0 sees and is only read by evaluation episodes
        self.geo = self.sim.geo_twin(self.cal.IF) if geo else None
        # move_atmos of frame t+1 runs on a side stream right behind the image kernels of frame t,
        # beside do_control / the agents / next_part_two (same results, bit for bit; between frames
        # the screens are one frame ahead).  Not with the GEO twin, which reads the same screens.
        self.prefetch_atmos = bool(prefetch_atmos) and self.geo is None and \
            hasattr(self.sim, "prefetch_atmos")
        if self.prefetch_atmos:
            self.sim.set_option("prefetch_atmos", 1)
        self.initial_seed, self.seed_stride = int(initial_seed), int(seed_stride)
        self.current_seed = int(initial_seed)
        self.iter = 0
        # supervisor.atmos.set_wind / set_r0 (atmosCompass.py:79-135)
        self.atmos = VecAtmos(self)

    # a change of wind / r0 made while the next frame's atmosphere was already moved (see VecAtmos): why, or None
    _atmos_changed_behind = None

    def _check_atmos_change(self):
        if self._atmos_changed_behind:
            raise RuntimeError("atmos.set_wind / set_r0 was called while %s: that frame keeps the old atmosphere, the "
                               "reference's would have the new one.  Call them right before reset(), or build the "
                               "environment with prefetch_atmos=False" % self._atmos_changed_behind)

    # ---------------------------------------------------------------- configuration
    def obtain_action_range_modal(self):
        """rlSupervisor.py:677-691"""
        lo, hi = self.n_modes_start_end
        if lo >= 0:
            rng = list(range(lo, hi))
            if self.include_tip_tilt:
                rng += [self.nmodes - 2, self.nmodes - 1]
            return np.asarray(rng)
        return np.arange(self.nmodes)

    def load_freedom_vector(self, zn_norm):
        """rlSupervisor.py:255-282: action bound per mode = zn_norm / norm_scale."""
        fv = np.asarray(zn_norm, dtype=np.float32) / \
            np.float32(self.config_rl["norm_scale_zernike_actions"])
        if fv.shape != (self.nmodes,):
            raise ValueError("freedom vector has %s entries, system has %d modes" %
                             (fv.shape, self.nmodes))
        self.freedom_vector = fv
        self._push_modal()

    def _push_modal(self):
        self.action_range = self.obtain_action_range_modal()
        self.sim.set_modal(self.volts2modes, self.modes2volts, self.freedom_vector,
                           self.action_range)

    def ensure_slopes2modes(self):
        """Push s2m = v2m . cmat (fp64 product) for `sim.slopes2modes` if it is not there yet."""
        if not self._s2m_ok:
            s2m = self.volts2modes.astype(np.float64) @ np.asarray(self.s.cmat, dtype=np.float64)
            self.sim.set_slopes2modes(s2m.astype(np.float32))
            self._s2m_ok = True

    def materialize_control(self):
        """Run the deferred do_control (err = -cmat . slopes, com += gain err) on the slopes of the
        last frame: exactly what the plain order would have done right after the frame."""
        if self._control_pending:
            self.sim.do_control()
            self._control_pending = False

    def _refresh_err(self):
        """err of the last frame when the command was rebuilt from modal coordinates: the product
        with the command matrix without integrating it into the (already final) command."""
        if self._err_stale:
            if self._env_gains:
                raise RuntimeError("err is not available with per-environment gains here")
            self.sim.set_gain(0.0)
            self.sim.do_control()
            self.sim.set_gain(self.gain)
            self._err_stale = False

    def set_sim_seed(self, seed):
        """rlSupervisor.py:207-213: the seed the NEXT reset starts from (environment e gets
        seed + seed_stride * e, its layer k seed + seed_stride * e + k)."""
        self.current_seed = int(seed)

    def env_seeds(self):
        return self.current_seed + self.seed_stride * np.arange(self.nenv)

    def seed_block(self):
        """Seeds one reset of this batch consumes: [current_seed, current_seed + seed_block())."""
        return self.seed_stride * self.nenv

    def next_seed_block(self, world_size=1):
        """The batched counterpart of the trainer's `seed += 1; set_sim_seed(seed)` after every
        episode (train_rpc.py:486-487, 495-496): every environment of every rank moves on to
        seeds nobody has used.  The reference's +1 would not do here: with a stride of 16 between
        environments, environment e of episode k + 16 would replay environment e + 1 of episode k.
        Ranks own consecutive blocks (dist.shard_seeds), so one episode of the whole job consumes
        world_size blocks."""
        self.set_sim_seed(self.current_seed + self.seed_block() * int(world_size))
        return self.current_seed

    # ---------------------------------------------------------------- loop
    # The NEXT reset's screens grown beside the running episode (aomarl_reset_prefetch_*, sim.prefetch_reset_*).
    # reset_prefetch: None (off); "same" (the next reset repeats this one's seeds: benchmarks, evaluation on fixed
    # seeds); an int w (the next reset follows next_seed_block(w): train_agent).  The rounds are dealt out evenly over
    # the first `reset_prefetch_span` steps of the episode (default: 95 % of max_steps_per_episode, so that every step
    # carries the same share); `step_done()` (called by VecAoEnv.step) issues them; a reset that comes earlier runs
    # what is left.
    reset_prefetch = None
    reset_prefetch_span = None

    def _begin_reset_prefetch(self):
        sim = self.sim
        if self.reset_prefetch is None or not hasattr(sim, "prefetch_reset_begin"):
            return
        if self.reset_prefetch == "same":
            nxt = self.env_seeds()
        else:
            nxt = self.env_seeds() + self.seed_block() * int(self.reset_prefetch)
        sim.prefetch_reset_begin(nxt)
        total = 2 * max(self.s.screen_dim) if hasattr(self.s, "screen_dim") else 0
        span = self.reset_prefetch_span
        if span is None:                        # evenly over (95 % of) the episode: every step carries the same share
            span = 0.95 * float(self.config_rl.get("max_steps_per_episode", 1000))
        self._rp_rate, self._rp_acc = total / max(1.0, float(span)), 0.0

    def step_done(self):
        """Once per environment step: the prefetched reset's share of rounds (no-op when none is pending)."""
        if self.reset_prefetch is not None and getattr(self, "_rp_left", 0):
            self._rp_acc += self._rp_rate
            k = int(self._rp_acc)
            if k:
                self._rp_acc -= k
                self._rp_left = self.sim.prefetch_reset_advance(k)

    def reset(self):
        """rlSupervisor.py:236-246 for every environment (seed e: current_seed + stride*e)."""
        if self.autoencoder is not None and hasattr(self.autoencoder, "check_range"):
            self.autoencoder.check_range()      # a saturated fp16 launch of the last episode is an error
        self.check_range()
        self.sim.reset(self.env_seeds())
        self._atmos_changed_behind = None
        self._rp_left = 0
        if self.reset_prefetch is not None:
            self._begin_reset_prefetch()
            self._rp_left = -1
        if self.geo is not None:
            self.geo.reset()
        self._control_pending, self._err_stale = False, False
        self._tar_image_se, self._tar_image_se_geo, self._le_count = None, None, 0      # target.reset_strehl: the long exposure starts over
        for t in (self._tar_image_le_sum, self._tar_image_le_sum_geo):
            if t is not None:
                t.zero_()
        self.iter = 0

    def check_range(self):
        """Fast mode (libaomarl.set_precision("split_f16")): raise if a split-fp16 GEMM of the finished
        episode clipped an operand at the fp16 range (aomarl_gemm_saturated) -- screens, commands or
        states of that episode are wrong.  Synchronises; called at episode boundaries."""
        if not hasattr(self.sim, "lib"):
            return
        from . import libaomarl as la
        import ctypes as C
        n = la.gemm_saturated(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if n:
            raise FloatingPointError(
                    "split-fp16 GEMM: operands left the fp16 range in %d kernel threads since the last check "
                    "(a diverging loop or policy?); results of that episode are wrong.  The counter is ONE per "
                    "device for the whole process and is cleared by whoever reads it: with several live "
                    "environments the clipping may have happened in another one.  Run in the default "
                    "precision (libaomarl.set_precision('f32'))" % n)

    def rl_control(self, action):
        """rlSupervisor.py:713-733 (+ correction_modal_basis :784-818) on the device."""
        std = self.config_rl["normalization_std_inside_environment"]
        mean = self.config_rl["normalization_mean_inside_environment"]
        if std != 1.0 or mean != 0.0:
            action = action * std + mean
        if self.freedom_vector is None:
            raise RuntimeError("freedom vector not loaded (load_freedom_vector)")
        self.sim.rl_control(action)

    def set_gain(self, gain):
        """rtc._rtc.d_control[0].set_gain (ao_env.py:950-958).  A scalar, or one gain per
        environment ([nenv], integrator control only: the gain scan runs its candidates as one
        batch)."""
        g = np.asarray(gain, dtype=np.float32).reshape(-1)
        if g.size == 1:
            self.gain = float(g[0])
            if self._env_gains:
                self.sim.set_env_gains(None)
                self._env_gains = False
            self.sim.set_gain(self.gain)
        else:
            self.sim.set_env_gains(g)
            self.gain, self._env_gains = None, True

    def obtain_and_set_cmat_filtered(self, modes_filtered):
        """Command matrix through the Btt basis without its last `modes_filtered` (non-TT) modes
        (rlSupervisor.py:215-234 -> basis.compute_cmat_with_Btt)."""
        self.n_reverse_filtered_from_cmat = int(modes_filtered)
        cmat = modal.cmat_with_btt(self.cal.imat, self.cal.Btt, max(int(modes_filtered), 0))
        self._proj_w2m = None
        self.cal.cmat = cmat
        self.s.cmat = np.ascontiguousarray(cmat)
        self.sim.set_cmat(self.s.cmat)
        self._s2m_ok = False

    @property
    def projector_wfs2modes(self):
        """rlSupervisor.py:191-194 (built when a reward type asks for it)."""
        if getattr(self, "_proj_w2m", None) is None:
            self._proj_w2m = modal.projector_wfs2modes(
                self.cal.imat, self.cal.Btt, max(int(getattr(self, "n_reverse_filtered_from_cmat", 0)), 0)).astype(np.float32)
        return self._proj_w2m

    def next_part_two(self, action, linear_control=False, apply_control=True,
                      compute_tar_psf=True, modes_pair=None, modes_out=None):
        """rlSupervisor.py:900-947.  modes_pair = (v2m . com_before, v2m . err) of the last
        integrator frame: the Btt coordinates of the current command follow by linearity
        (aomarl_rl_control_modes) and the coordinates after the action come back in
        `self.last_modes` (written into `modes_out` when given)."""
        self.last_modes = None
        snap = (self.keep_tar_image or self.keep_le_image) and compute_tar_psf
        if snap and not self.pure_delay_0:
            # comp_tar_image (:945-946) forms the image of the phase raytrace_target left in next_part_one: the
            # atmosphere of the frame + the mirrors BEFORE this call's apply_control -- the state as it stands now
            self._snap_tar_image()
        if not linear_control and modes_pair is not None and hasattr(self.sim, "rl_control_modes"):
            std = self.config_rl["normalization_std_inside_environment"]
            mean = self.config_rl["normalization_mean_inside_environment"]
            if std != 1.0 or mean != 0.0:
                action = action * std + mean
            if self.freedom_vector is None:
                raise RuntimeError("freedom vector not loaded (load_freedom_vector)")
            if self.gain is None:
                raise RuntimeError("per-environment gains are for integrator-only runs")
            self.last_modes = self.sim.rl_control_modes(modes_pair[0], modes_pair[1], self.gain, action,
                                                        out=modes_out)
            # the command is final without the integrator ever running in actuator space
            self._err_stale, self._control_pending = self._control_pending or self._err_stale, False
        else:
            self.materialize_control()
            if not linear_control:
                self.rl_control(action)
        if apply_control:
            # the stack-array shapes are left to the one-pass frame kernel when it can evaluate
            # them from the voltages (any other consumer materialises them on demand)
            self.sim.apply_control(defer_shape=getattr(self.sim, "defer_shape", False) and not self.pure_delay_0)
            if self.pure_delay_0:
                self.sim.target_psf()           # raytrace_target behind apply_control (rlSupervisor.py:938-939)
        if snap and self.pure_delay_0:
            self._snap_tar_image()              # ... and the image is that trace's
        if compute_tar_psf:
            self.sim.comp_strehl()
            if self.geo is not None:
                self.geo.comp_strehl()              # tar_trace covers every target (:943-946)

    def next_part_one(self, move_atmos=True, do_control=True, defer_control=False):
        """rlSupervisor.py:1015-1051 -> next_part_one_integrator :954-987.  defer_control: image the
        frame, leave do_control to `materialize_control` (run on demand)."""
        self._check_atmos_change()
        self.materialize_control()              # a frame still waiting for its do_control
        self._err_stale = False
        if defer_control and do_control:
            do_control = False
            self._control_pending = True
        if self.autoencoder is not None:
            # rlSupervisor.py:975-984 with the denoiser between image formation and centroiding;
            # the bincube never leaves the device (the reference copies it to the host and back)
            self._move_or_keep(move_atmos)
            if self.pure_delay_0:               # no target trace here (:964-965): the sensor's path alone
                self.sim.comp_image(noise=True, write_bincube=True, cog=False)
            else:
                self._target_and_image(write_bincube=True, cog=False)
            self.autoencoder.denoise_bincube_(self.sim.t["bincube"])
            if self.prefetch_atmos and move_atmos:      # beside centroids / control, not the denoiser
                self.sim.prefetch_atmos()
            self.sim.do_centroids()
            if do_control:
                self.sim.do_control()
        elif self.pure_delay_0:
            # no target trace here (:964-965): the sensor's path alone
            self._move_or_keep(move_atmos)
            self.sim.comp_image(noise=True, write_bincube=False, cog=True)
            if do_control:
                self.sim.do_control()
        elif move_atmos and do_control and self.geo is None and not self.next_part_one_split:
            self.sim.next_part_one()
        else:
            self._move_or_keep(move_atmos)
            self._target_and_image(write_bincube=False, cog=True)
            if self.prefetch_atmos and move_atmos:
                self.sim.prefetch_atmos()
            if do_control:
                self.sim.do_control()
        if self.geo is not None and do_control:
            self.geo.next_part_one_geo()            # next_part_one_geo, after controller 0 (:1038-1049)
        if self.keep_wfs_phase:
            self._snap_wfs_phase()
        self.iter += 1

    # The phase the sensor saw in the frame just imaged = what COMPASS leaves in d_gs.d_phase behind next_part_one's
    # raytrace (atmosphere of frame t + the mirrors as the PREVIOUS next_part_two left them); next_part_two only
    # re-traces the target (rlSupervisor.py:900-947), so wfs.get_wfs_phase(0) read by the rewards behind it
    # (ao_env.py:736-760) is still that phase.  The one-pass frame kernel never materialises it: with keep_wfs_phase
    # it is ray-traced once more right behind the frame (screens and mirror shapes are still the frame's) and kept.
    keep_wfs_phase = False
    _wfs_phase_frame = None

    def _snap_wfs_phase(self):
        if self.prefetch_atmos or getattr(self.sim, "pending_atmos", False):
            raise RuntimeError("keep_wfs_phase: the screens run one frame ahead (prefetch_atmos); build the "
                               "supervisor with prefetch_atmos=False")
        self.sim.raytrace_wfs(atm=True, dms=True, reset=True)
        ph = self.sim.t["wfs_phase"]
        if self._wfs_phase_frame is None or self._wfs_phase_frame.shape != ph.shape:
            self._wfs_phase_frame = torch.empty_like(ph)
        self._wfs_phase_frame.copy_(ph)

    def _move_or_keep(self, move_atmos):
        if move_atmos:
            self.sim.move_atmos()
        elif getattr(self.sim, "pending_atmos", False):
            raise RuntimeError("the next frame's atmosphere is already moved (prefetch_atmos): "
                               "build the supervisor with prefetch_atmos=False to image the same "
                               "atmosphere twice")

    def _target_and_image(self, write_bincube, cog):
        """raytrace_target + PSF, raytrace_wfs + comp_image (rlSupervisor.py:829-843)."""
        sim = self.sim
        if getattr(sim, "frame_fused_available", lambda: False)():
            sim.frame_fused(noise=True, write_bincube=write_bincube, cog=cog)
        else:
            sim.target_psf()
            sim.comp_image(noise=True, write_bincube=write_bincube, cog=cog)

    # ---------------------------------------------------------------- getters (device tensors)
    def get_command(self, ncontrol=0):
        self.materialize_control()
        if ncontrol == 1:
            if self.geo is None:
                raise RuntimeError("no geometric controller (VecRlSupervisor(..., geo=True))")
            return self.geo.com
        return self._get_command0()

    def _get_command0(self):
        return self.sim.com

    def get_slopes(self):
        return self.sim.slopes

    def get_err(self):
        self.materialize_control()
        self._refresh_err()
        return self.sim.err

    def get_voltages(self):
        return self.sim.voltage

    # Full-frame target images (targetCompass.py:71-92).  The hot path forms the PSF on the Strehl window only; with
    # keep_tar_image every next_part_two also forms the whole npsf x npsf short exposure of the phase the reference's
    # comp_tar_image sees (aomarl_target_image: two DFT passes per environment), with keep_le_image it is summed into
    # the long exposure too (d_image_le; [nenv, npsf, npsf] floats: 4.3 GB for 256 environments of the 40x40 system).
    keep_tar_image = False
    keep_le_image = False
    _tar_image_se, _tar_image_le_sum, _le_count = None, None, 0
    _tar_image_se_geo, _tar_image_le_sum_geo = None, None

    def _snap_tar_image(self):
        if self.prefetch_atmos or getattr(self.sim, "pending_atmos", False):
            raise RuntimeError("keep_tar_image / keep_le_image: the screens run one frame ahead (prefetch_atmos); build "
                               "the supervisor with prefetch_atmos=False")
        img = self.sim.target_image()
        self._tar_image_se = img
        # target 1 (the geometric controller's, comp_tar_image loops over every target: rlSupervisor.py:943-946): the
        # image of the phase next_part_one_geo left -- the frame's atmosphere + its own mirrors
        geo_img = self.geo.target_image() if (self.geo is not None and hasattr(self.geo, "target_image")) else None
        self._tar_image_se_geo = geo_img
        if self.keep_le_image:
            if self._tar_image_le_sum is None or self._tar_image_le_sum.shape != img.shape:
                self._tar_image_le_sum = torch.zeros_like(img)
                self._tar_image_le_sum_geo = torch.zeros_like(img) if geo_img is not None else None
                self._le_count = 0
            self._tar_image_le_sum += img
            if geo_img is not None and self._tar_image_le_sum_geo is not None:
                self._tar_image_le_sum_geo += geo_img
            self._le_count += 1

    def get_tar_image(self, tar_index=0, expo_type="se"):
        """targetCompass.py:71-92, [nenv, npsf, npsf], centred.  "se": d_image_se, the image comp_tar_image formed in
        the last next_part_two (of the phase next_part_one's raytrace_target left: a command applied in that
        next_part_two is not in it; with modification_online it is).  "le": d_image_le / strehl_counter, the mean of
        those images since the reset.  Needs keep_tar_image (VecAoEnv sets it for the rewards that read the image) /
        keep_le_image = True before the frames in question."""
        if tar_index not in (0, 1):
            raise IndexError("get_tar_image: target %d (0: the loop's, 1: the geometric controller's)" % tar_index)
        if tar_index == 1 and self.geo is None:
            raise RuntimeError("no geometric controller (VecRlSupervisor(..., geo=True)): target 1 is its target")
        se = self._tar_image_se if tar_index == 0 else self._tar_image_se_geo
        le = self._tar_image_le_sum if tar_index == 0 else self._tar_image_le_sum_geo
        if expo_type == "se":
            if not (self.keep_tar_image or self.keep_le_image) or se is None:
                raise RuntimeError("get_tar_image: no image was kept (set supervisor.keep_tar_image = True before "
                                   "next_part_two; VecAoEnv does for the rewards that read it)")
            return se
        if expo_type == "le":
            if not self.keep_le_image or not self._le_count or le is None:
                raise RuntimeError("get_tar_image(expo_type='le'): the full-frame long exposure is accumulated only when "
                                   "asked (set supervisor.keep_le_image = True before the episode)")
            return le / float(self._le_count)
        raise ValueError("Unknown exposure type")

    def get_wfs_phase(self):
        """wfsCompass.get_wfs_phase(0) (wfsCompass.py:366-372): the phase the sensor saw in the last next_part_one --
        atmosphere of that frame + the mirrors as they stood THEN, [nenv, n, n].  A command applied since
        (next_part_two) is not in it, exactly as in the reference, whose next_part_two re-traces the target only.
        Needs `keep_wfs_phase = True` before the frame (VecAoEnv sets it for the rewards that read it)."""
        if not self.keep_wfs_phase or self._wfs_phase_frame is None:
            raise RuntimeError("get_wfs_phase: the frame's sensor phase was not kept (set supervisor.keep_wfs_phase "
                               "= True before next_part_one; VecAoEnv does for the projection rewards)")
        return self._wfs_phase_frame

    _projector_phase2modes = None

    @property
    def projector_phase2modes(self):
        """get_projector_wfsphase2modes (helper_functions/utils/utils.py:88-160, mode 2): every Btt mode poked on the
        mirrors (atmosphere off), the WFS phase on the pupil pixels recorded; least-squares projectors of the
        stack-array modes and of the two tip-tilt modes, stacked: [nmodes, pupil pixels].  Built on first use (it
        pokes the mirrors of this simulator: call it before an episode, the next reset clears them)."""
        if self._projector_phase2modes is None:
            m2v, sim = np.asarray(self.modes2volts, dtype=np.float32), self.sim
            pup = np.flatnonzero(np.asarray(self.s.mpupil).reshape(-1) != 0)
            nm = m2v.shape[1]
            resp = np.zeros((nm, pup.size), dtype=np.float64)
            for k0 in range(0, nm, sim.nenv):
                n = min(sim.nenv, nm - k0)
                cmd = np.zeros((n, m2v.shape[0]), dtype=np.float32)
                for i in range(n):                  # (a mode drives the stack array OR the tip-tilt mirror: :120-150)
                    mode = k0 + i
                    if mode < nm - 2:
                        cmd[i, :-2] = m2v[:-2, mode]
                    else:
                        cmd[i, -2:] = m2v[-2:, mode]
                sim.comp_dm_shape(torch.from_numpy(cmd).to(sim.device), 0, n)
                sim.raytrace_wfs(atm=False, dms=True, reset=True, env_begin=0, env_count=n)
                resp[k0:k0 + n] = sim.t["wfs_phase"][:n].reshape(n, -1)[:, torch.as_tensor(pup, device=sim.device)].double().cpu().numpy()
            a, t = resp[:nm - 2], resp[nm - 2:]
            proj = np.concatenate([np.linalg.inv(a @ a.T) @ a, np.linalg.inv(t @ t.T) @ t])
            self._projector_phase2modes = proj.astype(np.float32)
        return self._projector_phase2modes

    def get_strehl(self, tar_index=0, do_fit=True):
        """targetCompass.py:139-159: [SR SE, SR LE, phase variance SE, phase variance LE] per environment; do_fit
        (default True, like the reference): the PSF peak fitted by two 1-D sincs."""
        if tar_index == 1:
            if self.geo is None:
                raise RuntimeError("no geometric controller (VecRlSupervisor(..., geo=True))")
            src = self.geo
        else:
            src = self.sim
        return src.strehl_fit if (do_fit and hasattr(src, "strehl_fit")) else src.strehl


class VecAoEnv(object):
    """Batched AoEnv (gym-like).  step(action) == TrainerRPC.env_step: rl_step(a), per-agent
    rewards from the residual measured BEFORE the action reached the DM, then linear_step()."""

    def __init__(self, parameters_telescope, nenv, config_rl=None, *, normalization_bool=True,
                 initial_seed=1234, seed_stride=16, n_agents_modal=None, device="cuda:0",
                 strehl_halfwin=8, norm=None, zn_norm=None, sim_factory=None, autoencoder=None,
                 geo=False, prefetch_atmos=True, frame_pipeline=False, dead_columns="mask",
                 reset_prefetch=None):
        # frame_pipeline: False (default: the reference's call order; every call-by-call piece -- rl_step,
        # linear_step, step(linear_control=True), the supervisor's getters -- can be mixed with step() freely),
        # True (a frame in flight wherever the loop is eligible), or "auto" (True unless a probe of both orders
        # behind the first reset finds it slower on this process's streams).  bench.py and throughput-minded
        # trainers opt in; while a frame is in flight the state accepts step() and a full reset() only.
        cfg = dict(DEFAULT_ENV_RL)
        cfg.update(config_rl or {})
        self.config_rl = cfg
        self.normalization_bool = normalization_bool
        if isinstance(parameters_telescope, params.ParamSet):
            config, name = parameters_telescope, parameters_telescope.simul_name
        else:
            name = parameters_telescope[:-3] if parameters_telescope.endswith(".py") else \
                parameters_telescope
            config = name
        from . import rewards as _R
        self._order_pinned = False
        if cfg["reward_type"] in _R.NEEDS_CURRENT_SCREENS or cfg["modification_online"]:
            # these rewards read the image / phase of THIS frame, and the pure-delay-0 order traces the target behind
            # apply_control: the screens must not run ahead
            prefetch_atmos, frame_pipeline = False, False
            self._order_pinned = True
        self.supervisor = VecRlSupervisor(config, cfg, nenv, initial_seed=initial_seed,
                                          seed_stride=seed_stride, device=device,
                                          strehl_halfwin=strehl_halfwin, sim_factory=sim_factory,
                                          autoencoder=autoencoder, geo=geo,
                                          prefetch_atmos=prefetch_atmos)
        sup = self.supervisor
        if cfg["reward_type"] in _R.PROJECTION:
            sup.keep_wfs_phase = True                # these read wfs.get_wfs_phase(0): the frame's sensor phase
        if cfg["reward_type"] in _R.IMAGE:
            sup.keep_tar_image = True                # these read target.get_tar_image(0): comp_tar_image's frame
        # the next reset's screens grown beside the running episode: None, "same" or the number of seed blocks the
        # trainer moves on by per episode (VecRlSupervisor.reset_prefetch; train_agent sets it)
        sup.reset_prefetch = reset_prefetch
        self.nenv, self.device = nenv, sup.device
        self.nmodes = sup.nmodes
        if normalization_bool:
            if norm is None or zn_norm is None:
                norm, zn_norm = load_norm(name)
            sup.load_freedom_vector(zn_norm)
        self.windowed = cfg["window_n_zernike"] > -1 or cfg["tt_treated_as_mode"]
        ar = sup.obtain_action_range_modal()
        self._sel = None if (self.windowed or cfg["n_zernike_start_end"][0] < 0) else \
            torch.as_tensor(ar % self.nmodes, dtype=torch.long, device=self.device)
        self.dm_dim = self.nmodes if self._sel is None else int(self._sel.numel())
        self.action_dim = len(ar)
        self.wfs_dim = sup.s.nslope
        # standardisation vectors (ao_env.py:470-480), sub-selected like load_norm_parameters does
        self.norm = None
        # Degenerate statistics: a state column whose recorded standard deviation is ~0 (the modes filtered out of the
        # command matrix: 1e-10; and, in the reference's own 10x10 file, 10 modes at 4e-9 .. 9e-9) turns round-off
        # into 1e+3 .. 1e+9 when the state is standardised -- the reference feeds its agents exactly that, and a SAC
        # trained on it here diverges (alpha -> inf, NaN; profiles/r03_learning_acceptance.txt).  dead_columns:
        # "mask" (default) standardises such columns to 0 and warns, naming the file; "raise"; "keep" = the
        # reference's division as it stands.
        if dead_columns not in ("mask", "raise", "keep"):
            raise ValueError("dead_columns: 'mask', 'raise' or 'keep'")
        self.dead_columns = {}
        if normalization_bool:
            self.norm = {}
            for k in ("wfs", "dm", "dm_residual"):
                m, sd = norm[k]["mean"], norm[k]["std"]
                if k != "wfs" and self._sel is not None:
                    m, sd = m[ar], sd[ar]
                sd = np.asarray(sd, dtype=np.float64)
                pos = sd[sd > 0]
                dead = sd < 1e-6 * (np.median(pos) if pos.size else 1.0)
                if dead.any() and dead_columns != "keep":
                    idx = np.flatnonzero(dead)
                    msg = ("norm_%s.npz: %d column(s) of %r have a recorded standard deviation below 1e-6 x the median "
                           "(%.1e .. %.1e; columns %s%s)" % (name, idx.size, k, sd[dead].min(), sd[dead].max(),
                                                            idx[:12].tolist(), " ..." if idx.size > 12 else ""))
                    if dead_columns == "raise":
                        raise ValueError(msg + ": standardising them amplifies round-off by 1 / std "
                                         "(dead_columns='mask' zeroes them, 'keep' divides like the reference)")
                    import warnings
                    warnings.warn(msg + ": standardised to 0 in the states (pass dead_columns='keep' for the "
                                        "reference's division as it stands, 'raise' to refuse)")
                    sd = np.where(dead, np.inf, sd)
                    self.dead_columns[k] = idx
                self.norm[k] = (torch.as_tensor(m, dtype=torch.float32, device=self.device),
                                torch.as_tensor(sd, dtype=torch.float32, device=self.device))
        # state layout
        keys = []
        for i in range(cfg["number_of_previous_wfs"], 0, -1):
            keys.append(("wfs_history-%d" % i, self.wfs_dim))
        if cfg["state_wfs"]:
            keys.append(("wfs", self.wfs_dim))
        for i in range(cfg["number_of_previous_dm"], 0, -1):
            keys.append(("dm_history_%d" % i, self.dm_dim))
        if cfg["state_dm_after_linear"]:
            keys.append(("dm_after_linear", self.dm_dim))
        if cfg["state_dm_before_linear"]:
            keys.append(("dm_before_linear", self.dm_dim))
        for i in range(cfg["number_of_previous_dm_residuals"], 0, -1):
            keys.append(("dm_residual_history_%d" % i, self.dm_dim))
        if cfg["state_dm_residual"]:
            keys.append(("dm_residual", self.dm_dim))
        self.state_keys = OrderedDict(keys)
        self.state_dim = sum(self.state_keys.values())
        self.reward_type = cfg["reward_type"]
        # agents
        self.layout = None
        if n_agents_modal is not None:
            self.set_agents(n_agents_modal)
        self._hist_dm, self._hist_wfs, self._hist_res = [], [], []
        self._last_res_modes = None
        # fused glue kernels of the library (GPU + the HIP simulator only)
        self._native_glue = self.device.type == "cuda" and hasattr(sup.sim, "lib")
        # Btt coordinates of the command carried from frame to frame by linearity instead of two
        # v2m GEMMs per step (aomarl_rl_control_modes); fp32 round-off apart, the same numbers
        self.modal_shortcut = True
        # v2m . err straight from the slopes: ONE product with v2m . cmat instead of the reference's do_control
        # (cmat . s on the device, integrate) + v2m . err -- same mathematics, another order of the fp32 sums.  Bounds
        # the tests enforce: states within 3e-3 relative of the reference order's over 10 steps of the 40x40 system
        # (1.3e-5 measured; tests/test_gpu_glue.py::test_residual_shortcut_inside_the_one_call_step), the oracle
        # environment in the reference's order at the usual 2e-3 / 1e-4 arcsec (tests/test_gpu_env_step_large.py,
        # `bench` cases), the trace of the reference's own Python at the GPU tolerance (tests/
        # test_env_vs_reference_trace.py).  Off by default (the reference's order; the pipelined step then IS the
        # plain step bit for bit); `throughput_mode()` -- bench.py, train_agent -- switches it on: one product and two
        # launches less in the control chain.  Works inside the one-call step (aomarl_env_step, "residual_shortcut")
        # and call by call; err / com in actuator space appear on demand.
        self.residual_shortcut = False
        # one library call per environment step (aomarl_env_step) when the configuration is the one it
        # covers (see _native_step_ok); the same launches in the same order as the call-by-call path
        self.native_step = True
        self.fused_tail = True       # ... with the reductions folded into their consumers (see include/aomarl.h)
        # frames one step ahead of the chains (aomarl_set_frame_pipeline; loop delay of one frame, noise-free
        # sensor -- the library takes the plain order whenever a step is not eligible).  Between two resets a
        # pipelined environment takes step() calls only (no call-by-call pieces, no linear_control steps: they
        # raise, naming this argument).  "auto" (default): pipelined when eligible AND not slower on this
        # process's streams (one probe of both orders behind the first reset, see _probe_order); True: whenever
        # eligible, no probe; False: plain call order.
        if frame_pipeline not in (True, False, "auto"):
            raise ValueError("frame_pipeline: True, False or 'auto'")
        self.frame_pipeline = frame_pipeline
        self.order_probe = None      # {"pipelined": ms/step, "plain": ms/step, "chosen": ...} once probed
        # Btt coordinates of the last number_of_previous_dm + 1 commands, newest in slot _ring_pos
        self._ring, self._ring_pos, self._ring_next_valid = None, 0, False
        self._res_modes, self._glue, self._glue_keep = None, None, None
        self._modal_valid = False
        self._default_state_layout = (
            list(self.state_keys) == ["dm_history_%d" % i for i in
                                      range(cfg["number_of_previous_dm"], 0, -1)] +
            ["dm_before_linear", "dm_residual"] and cfg["number_of_previous_dm"] > 0)

    def throughput_mode(self, reset_prefetch=None):
        """The configuration of the package's own loops -- bench.py's timed steps, `sac.train_agent`, the learning
        acceptance tools: everything that leaves the numbers where the tests pin them and makes the step faster.
        The frame pipeline where the loop is eligible and not slower on this process's streams ("auto": one probe
        behind the next reset; bit-identical to the plain order), the residual modes from one product
        (`residual_shortcut`: the reference's numbers to fp32 round-off, see __init__), and -- `reset_prefetch`
        "same" or the number of seed blocks the caller moves on by per episode -- the next reset's screens grown
        beside the episode (bit-identical to the reset in the open).  Call before reset(); returns self.  What it
        costs: between two resets a pipelined environment takes step() / policy_step() and a full reset() only."""
        if self.frame_pipeline is False and not self._order_pinned and self.supervisor.prefetch_atmos:
            self.frame_pipeline, self._pipe_checked = "auto", False
        self.residual_shortcut = True
        if reset_prefetch is not None and not self._order_pinned and self.supervisor.prefetch_atmos:
            self.supervisor.reset_prefetch = reset_prefetch
        return self

    # ------------------------------------------------------------------ agents
    def set_agents(self, n_agents_modal):
        cfg = self.config_rl
        self.layout = AgentLayout(self.nmodes, cfg["n_zernike_start_end"], n_agents_modal,
                                  include_tip_tilt=cfg["include_tip_tilt"],
                                  window_n_zernike=cfg["window_n_zernike"],
                                  include_tip_tilt_windowed=cfg["include_tip_tilt_windowed"],
                                  n_filtered=cfg["n_reverse_filtered_from_cmat"],
                                  state_keys=tuple(self.state_keys), state_block=self.dm_dim)
        assert self.layout.state_dim == self.state_dim or cfg["state_wfs"] or \
            cfg["number_of_previous_wfs"] > 0
        factor = float(self.reward_type.split("_")[-1])        # helper_rewards.py:18
        M = torch.zeros(self.nmodes, self.layout.n_agents, device=self.device)
        for j, (w, (a, b)) in enumerate(self.layout.agents.items()):
            M[a:b, j] = -factor / (b - a)                      # -factor * mean(reward[a:b])
        self._reward_mat = M
        self._reward_factor = factor
        self._lohi_i32 = torch.tensor([list(v) for v in self.layout.agents.values()],
                                      dtype=torch.int32, device=self.device)
        return self.layout

    # ------------------------------------------------------------------ helpers
    def _standardise(self, x, key):
        if not self.normalization_bool:
            return x
        m, sd = self.norm[key]
        return (x - m) / sd

    def transform_state_to_zernike(self, volts, return_reward=False):
        """ao_env.py:482-505"""
        m = self.supervisor.sim.volts2modes(volts)
        if (self.windowed and not return_reward) or self._sel is None:
            return m
        return m[:, self._sel]

    # ------------------------------------------------------------------ gym-like API
    def set_sim_seed(self, seed):
        """ao_env.py:454-459"""
        self.supervisor.set_sim_seed(seed)

    def next_seed_block(self, world_size=1):
        return self.supervisor.next_seed_block(world_size)

    def reset(self):
        """ao_env.py:316-359"""
        cfg = self.config_rl
        self.supervisor.reset()
        self._ring_next_valid, self._modal_valid = False, False
        z = lambda d: torch.zeros(self.nenv, d, device=self.device)  # noqa: E731
        self._hist_dm = [z(self.dm_dim) for _ in range(cfg["number_of_previous_dm"])]
        self._hist_wfs = [z(self.wfs_dim) for _ in range(cfg["number_of_previous_wfs"])]
        self._hist_res = [z(self.dm_dim) for _ in range(cfg["number_of_previous_dm_residuals"])]
        if self._native_glue and self._default_state_layout:
            if self._ring is None:
                self._ring = torch.zeros(cfg["number_of_previous_dm"] + 1, self.nenv, self.nmodes,
                                         device=self.device)
                self._res_modes = torch.zeros(self.nenv, self.nmodes, device=self.device)
                self._sel_i32 = None if self._sel is None else self._sel.to(torch.int32)
            else:
                self._ring.zero_()
            self._ring_pos, self._glue = 0, None
            self._out_pos = 0
        state = self.linear_step()
        if self.frame_pipeline == "auto" and not self._probing:
            if self._pipe_eligible():
                state = self._probe_order(state)
            else:
                self.frame_pipeline = False
        return state

    # call orders probed per (device, caller stream): the HIP runtime multiplexes streams onto hardware queues
    # (ao_marl_amd/__init__.py); a caller stream that shares one with the library's frame stream serialises the
    # pipelined order (1.0 against 0.6 ms per step seen).  One probe per process and stream pair.
    _ORDER_CACHE = {}
    _probing = False

    def _pipe_eligible(self):
        sup = self.supervisor
        return bool(self._native_glue and self._default_state_layout and self._ring is not None and
                    hasattr(sup.sim, "enable_frame_pipeline") and sup.autoencoder is None and
                    sup.s.delay == 1.0 and sup.s.noise < 0 and sup.prefetch_atmos and sup.geo is None and
                    not getattr(sup.sim, "graph_step", False))

    def _probe_order(self, state, steps=28, skip=6, margin=1.15, alias_ratio=1.3):
        """Behind the first reset of an eligible environment: `steps` steps with zero actions in the pipelined and
        in the plain call order, the period between the states of step `skip` and step `steps` becoming ready on the
        caller's stream (device events).  The pipelined order is kept unless it is more than `margin` times SLOWER
        than the plain one (aliased hardware queues) -- then a warning says so and the environment runs in the
        plain order.  The environment is reset again afterwards (same seeds: same episode).
        margin: what the probe guards against is a 1.6 x slowdown (1.0 against 0.6 ms per step with an aliased queue);
        22 steps right behind a reset scatter by several per cent, and at 1.03 one bench run in forty took the plain
        order on a box where the pipelined one is 10 % faster (0.532 against 0.478 ms per step)."""
        import warnings
        key = (str(self.device), int(torch.cuda.current_stream(self.device).cuda_stream), self.nenv,
               self.supervisor.s.name if hasattr(self.supervisor.s, "name") else "")
        cached = VecAoEnv._ORDER_CACHE.get(key)
        if cached is not None:
            self.order_probe = dict(cached, cached=True)
            self.frame_pipeline = cached["chosen"] == "pipelined"
            return state
        sim = self.supervisor.sim
        zero = torch.zeros(self.nenv, self.action_dim, device=self.device)
        res = {}
        self._probing = True
        keep_rp, self.supervisor.reset_prefetch = self.supervisor.reset_prefetch, None   # no shadow resets for the probe's resets
        def timed_pass(on):
            self.frame_pipeline, self._pipe_checked = on, False
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in range(steps):
                self.step(zero)
                if k == skip - 1:
                    e0.record()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / (steps - skip), self.reset()
        try:
            for name, on in (("plain", False), ("pipelined", True)):      # (the twin of the second pass stays)
                res[name], state = timed_pass(on)
            # The signature of a frame stream that shares a hardware queue with another stream of the step is a
            # pipelined order nearly twice as slow (1.02 against 0.47 ms per step at production size, one process in
            # fifty on this runtime): the library's frame stream is created anew -- the runtime deals queues out as
            # streams come -- and the pipelined pass repeated, twice at most, before the plain order is settled for.
            res["frame_stream_renewed"] = 0
            while (res["pipelined"] > alias_ratio * res["plain"] and res["frame_stream_renewed"] < 2 and
                   hasattr(sim, "renew_frame_stream")):
                sim.renew_frame_stream()
                res["frame_stream_renewed"] += 1
                res["pipelined_before_renewal"] = res.get("pipelined_before_renewal", res["pipelined"])
                res["pipelined"], state = timed_pass(True)
        finally:
            self._probing = False
            self.supervisor.reset_prefetch = keep_rp
        slower = res["pipelined"] > margin * res["plain"]
        res["chosen"] = "plain" if slower else "pipelined"
        if slower:
            warnings.warn("VecAoEnv: the pipelined call order is slower than the plain one here (%.3f against %.3f ms per "
                          "step) -- small batches gain nothing from a frame in flight; at production size the caller's "
                          "stream probably shares a hardware queue with the library's frame stream (GPU_MAX_HW_QUEUES=8 "
                          "before the first HIP call avoids that); running in the plain order"
                          % (res["pipelined"], res["plain"]))
        self.frame_pipeline = not slower
        if slower:
            sim.enable_frame_pipeline(False)               # (behind a reset: nothing in flight)
        VecAoEnv._ORDER_CACHE[key] = dict(res)
        self.order_probe = res
        return state

    def linear_step(self, return_dict=False):
        """ao_env.py:871-909"""
        cfg, sup = self.config_rl, self.supervisor
        if self._ring is not None and not return_dict:
            return self._linear_step_fused()
        pick = lambda m: m if (self.windowed or self._sel is None) else m[:, self._sel]  # noqa: E731
        if self._ring is not None:
            # the dictionary form of the default layout: same ring as the fused path, so the two can
            # be mixed freely within an episode
            R = self._ring.shape[0]
            nxt = (self._ring_pos + 1) % R
            if not self._ring_next_valid:
                sup.sim.volts2modes(sup.get_command(), out=self._ring[nxt])
            self._ring_next_valid = False
            self._hist_dm = [pick(self._ring_slot(h)) for h in range(R - 2, -1, -1)]
            self._ring_pos = nxt
            s_dm_before = pick(self._ring[nxt])
        else:
            s_dm_before = self.transform_state_to_zernike(sup.get_command())
        sup.next_part_one()
        s_dm_after = self.transform_state_to_zernike(sup.get_command()) \
            if cfg["state_dm_after_linear"] else None
        if self._ring is not None:
            res_full = sup.sim.volts2modes(sup.get_err(), out=self._res_modes)
            self._modal_valid = True
        else:
            res_full = sup.sim.volts2modes(sup.get_err())
        self._last_res_modes = res_full
        s_res = pick(res_full)
        s_wfs = sup.get_slopes()
        out = OrderedDict()
        # add_wfs_to_state (ao_env.py:563-583)
        n = len(self._hist_wfs)
        for i, h in enumerate(self._hist_wfs):
            out["wfs_history-%d" % (n - i)] = self._standardise(h, "wfs")
        if cfg["number_of_previous_wfs"] > 0:
            self._hist_wfs = self._hist_wfs[1:] + [s_wfs.clone()]
        if cfg["state_wfs"]:
            out["wfs"] = self._standardise(s_wfs, "wfs")
        # add_dm_to_state (ao_env.py:535-561)
        n = len(self._hist_dm)
        for i, h in enumerate(self._hist_dm):
            out["dm_history_%d" % (n - i)] = self._standardise(h, "dm")
        if cfg["number_of_previous_dm"] > 0:
            self._hist_dm = self._hist_dm[1:] + [s_dm_before]
        if cfg["state_dm_after_linear"]:
            out["dm_after_linear"] = self._standardise(s_dm_after, "dm")
        if cfg["state_dm_before_linear"]:
            out["dm_before_linear"] = self._standardise(s_dm_before, "dm")
        # add_s_dm_residual_to_state (ao_env.py:507-533)
        n = len(self._hist_res)
        for i, h in enumerate(self._hist_res):
            out["dm_residual_history_%d" % (n - i)] = self._standardise(h, "dm_residual")
        if cfg["number_of_previous_dm"] > 0 and cfg["number_of_previous_dm_residuals"] > 0:
            self._hist_res = self._hist_res[1:] + [s_res]
        if cfg["state_dm_residual"]:
            out["dm_residual"] = self._standardise(s_res, "dm_residual")
        if return_dict:
            if not self.normalization_bool:
                # without standardisation the blocks are views of the command ring / the residual buffer,
                # which later steps overwrite in place: hand out copies
                out = OrderedDict((k, v.clone()) for k, v in out.items())
            return out
        return torch.cat(list(out.values()), dim=1)

    def _ring_slot(self, back):
        """Btt coordinates of the command `back` steps before the newest one (a view of the ring)."""
        return self._ring[(self._ring_pos - back) % self._ring.shape[0]]

    def _linear_step_fused(self):
        """linear_step for the default state layout on the GPU: the blocks are sub-selected,
        standardised and concatenated by one kernel (aomarl_assemble_state_cols) instead of ~10 tensor
        operations.  The Btt coordinates of the commands live in a ring ([number_of_previous_dm + 1,
        nenv, nmodes]) shared with the one-call step (aomarl_env_step)."""
        from . import libaomarl as la
        sup = self.supervisor
        R = self._ring.shape[0]
        nxt = (self._ring_pos + 1) % R
        if self._ring_next_valid:               # v2m . com came back from rl_control_modes
            self._ring_next_valid = False
        else:
            sup.sim.volts2modes(sup.get_command(), out=self._ring[nxt])
        self._ring_pos = nxt
        if (self.residual_shortcut and self.modal_shortcut and sup.geo is None and sup.gain is not None
                and sup.autoencoder is None and hasattr(sup.sim, "slopes2modes")):
            # the next control step rebuilds the command from Btt coordinates (rl_step below), so
            # the frame needs neither err nor the integrated command in actuator space: one product
            # with v2m . cmat instead of do_control + volts2modes; do_control runs on demand
            sup.ensure_slopes2modes()
            sup.next_part_one(defer_control=True)
            self._res_modes.copy_(sup.sim.slopes2modes())
        else:
            sup.next_part_one()
            sup.sim.volts2modes(sup.get_err(), out=self._res_modes)
        self._last_res_modes = self._res_modes
        self._modal_valid = True
        blocks = [self._ring_slot(h) for h in range(R - 1, -1, -1)] + [self._res_modes]
        norms = None
        if self.normalization_bool:
            norms = [self.norm["dm"]] * R + [self.norm["dm_residual"]]
        return la.assemble_state(blocks, norms, sel_i32=self._sel_i32)

    # ------------------------------------------------------------------ one library call per step
    def _native_step_ok(self, linear_control):
        sup = self.supervisor
        return (self.native_step and self._native_glue and self._default_state_layout and
                self.modal_shortcut and self._modal_valid and not linear_control and
                sup.geo is None and sup.gain is not None and not sup.pure_delay_0 and
                not sup.keep_wfs_phase and not sup.keep_tar_image and not sup.keep_le_image and       # (the frame's sensor phase is snapped behind the Python next_part_one)
                sup.freedom_vector is not None and not sup.next_part_one_split and
                # (a do_control left pending by the residual shortcut is dropped by the step: its head rebuilds the
                # command from the Btt coordinates, the integrator included)
                (not sup._control_pending or (self.residual_shortcut and sup.autoencoder is None)) and
                hasattr(sup.sim, "env_step") and
                (sup.autoencoder is None or (getattr(sup.autoencoder, "use_native", False) and
                                             sup.autoencoder.input_bound is not None)))

    def _make_glue(self):
        from . import libaomarl as la
        sup, cfg = self.supervisor, self.config_rl
        g = la.EnvGlue()
        g.nmodes, g.dm_dim, g.nhist = self.nmodes, self.dm_dim, cfg["number_of_previous_dm"]
        g.sel = self._sel_i32.data_ptr() if self._sel_i32 is not None else None
        if self.normalization_bool:
            (g.mean_dm, g.std_dm), (g.mean_res, g.std_res) = \
                [tuple(t.data_ptr() for t in self.norm[k]) for k in ("dm", "dm_residual")]
        if self.layout is not None:
            g.n_agents, g.lohi, g.reward_factor = self.layout.n_agents, self._lohi_i32.data_ptr(), \
                self._reward_factor
        g.modes_ring, g.res_modes = self._ring.data_ptr(), self._res_modes.data_ptr()
        g.flags = 0 if self.fused_tail else la.ENV_STEP_UNFUSED
        ae = sup.autoencoder
        if ae is not None:
            g.denoiser, g.denoiser_f32 = ae._native().value, int(bool(ae.wants_f32()))
            sup.sim._need_bincube()
        self._glue = g

    OUT_RING = 6        # a multiple of the command ring's period (number_of_previous_dm + 1 = 3)
    _out_ring, _out_pos = None, 0
    _pipe_checked = False
    _native_shortcut = False

    def policy_step(self, policy, state, eps=None, out=None, action_out=None):
        """TrainerRPC.choose_action + TrainerRPC.env_step (train_rpc.py:650-675, 633-648) in ONE library call
        (aomarl_policy_env_step): a = policy.select_action(state) (sampled), then step(a).  Returns
        (action, s_next, per-agent reward, done, info): the same launches as select_action followed by step, issued by
        one call (a host-bound step of a small system saves a host round trip); whenever the one-call step does not
        apply, this IS select_action followed by step.  out = (state, reward) / action_out: the caller's buffers."""
        sup = self.supervisor
        std = sup.config_rl["normalization_std_inside_environment"]
        mean = sup.config_rl["normalization_mean_inside_environment"]
        ok = (self._native_step_ok(False) and std == 1.0 and mean == 0.0 and getattr(policy, "use_native", False) and
              getattr(policy, "native_forward", False) and not getattr(policy, "out_ring", 0) and
              isinstance(state, torch.Tensor) and state.dtype == torch.float32 and state.dim() == 2 and
              state.shape == (self.nenv, policy.layout.state_dim) and state.device == self.device and
              policy.layout.action_dim == self.action_dim and
              hasattr(sup.sim, "policy_env_step") and not getattr(sup.sim, "graph_step", False))
        if not ok:
            a, _ = policy.select_action(state, eps=eps, out=action_out)
            s_next, r, done, info = self.step(a, out=out)
            return a, s_next, r, done, info
        if not state.is_contiguous():
            state = state.contiguous()
        d = policy._actor_desc(self.nenv)
        policy._draws += 1
        a = action_out if action_out is not None else torch.empty(self.nenv, self.action_dim, dtype=torch.float32, device=self.device)
        if a.shape != (self.nenv, self.action_dim) or a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            raise ValueError("policy_step: action_out must be a contiguous float32 [nenv, action_dim] tensor on the environment's device")
        if self._mean_buf is None or self._mean_buf.shape != a.shape:
            self._mean_buf = torch.empty_like(a)
        if eps is not None:
            eps = eps.to(torch.float32).contiguous()
        s_next, r, done, info = self._step_native(a, out, policy=(d, state, eps, policy.seed, policy._draws, self._mean_buf))
        return a, s_next, r, done, info

    _mean_buf = None

    def _step_native(self, action, out=None, policy=None):
        sup = self.supervisor
        std = sup.config_rl["normalization_std_inside_environment"]
        mean = sup.config_rl["normalization_mean_inside_environment"]
        if not (isinstance(action, torch.Tensor) and action.dtype == torch.float32 and action.device == self.device):
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device)
        if policy is None and (std != 1.0 or mean != 0.0):
            action = action * std + mean
        if not action.is_contiguous():
            action = action.contiguous()
        if action.shape != (self.nenv, self.action_dim):
            raise ValueError("action must be [nenv, %d]" % self.action_dim)
        sup._check_atmos_change()
        if self._glue is None:
            self._make_glue()
        g = self._glue
        g.ring_pos = self._ring_pos
        if getattr(sup.sim, "graph_step", False):
            # graph replay needs the addresses it was captured with: the outputs cycle through a ring of
            # OUT_RING buffers (a returned state / reward stays valid for OUT_RING - 1 further steps; clone
            # what must live longer -- DelayedMDP holds a state for delay + 1 steps, the replay copies)
            if self._out_ring is None:
                self._out_ring = [(torch.empty(self.nenv, self.state_dim, dtype=torch.float32, device=self.device),
                                   torch.empty(self.nenv, self.layout.n_agents, dtype=torch.float32, device=self.device)
                                   if self.layout is not None else None) for _ in range(self.OUT_RING)]
                self._out_pos = 0
            state, r = self._out_ring[self._out_pos]
            self._out_pos = (self._out_pos + 1) % self.OUT_RING
        elif out is not None:
            state, r = out
            ok = lambda t, d: (t.shape == (self.nenv, d) and t.dtype == torch.float32 and t.is_contiguous() and  # noqa: E731
                               t.device == self.device)
            if not ok(state, self.state_dim) or (self.layout is not None and not ok(r, self.layout.n_agents)):
                raise ValueError("step: out = (state [nenv, state_dim], reward [nenv, n_agents]), contiguous float32 on the "
                                 "environment's device")
            if self.layout is None:
                r = None
        else:
            state = torch.empty(self.nenv, self.state_dim, dtype=torch.float32, device=self.device)
            r = None
            if self.layout is not None:
                r = torch.empty(self.nenv, self.layout.n_agents, dtype=torch.float32, device=self.device)
        ae = sup.autoencoder
        if ae is not None:
            ae._used_fp16 = ae._used_fp16 or not g.denoiser_f32
        if self.frame_pipeline is True and not self._pipe_checked:
            self._pipe_checked = True
            if self._pipe_eligible():
                sup.sim.enable_frame_pipeline()
        shortcut = bool(self.residual_shortcut) and ae is None
        if shortcut != self._native_shortcut:
            if shortcut:
                sup.ensure_slopes2modes()
            sup.sim.set_option("residual_shortcut", int(shortcut))
            self._native_shortcut = shortcut
        if policy is not None:      # (the action is an OUTPUT here: the actors run inside the call)
            d, s_in, eps, seed, counter, mean_buf = policy
            sup.sim.policy_env_step(g, d, s_in, eps, seed, counter, sup.gain, action, mean_buf, state, r)
        else:
            sup.sim.env_step(g, action, sup.gain, state, r)
        if shortcut:        # (a small system's tail kernel runs do_control itself: nothing is left pending there; asked
            #                  BEHIND the call: it is the call that switches the deferred mirror shapes on)
            shortcut = bool(sup.sim.lib.aomarl_env_step_shortcut(sup.sim.ctx, g._ref if hasattr(g, "_ref") else ctypes.byref(g)))
        if sup.reset_prefetch is not None:
            sup.step_done()
        self._ring_pos = g.ring_pos
        self._last_res_modes = self._res_modes
        sup.last_modes = None
        # (shortcut: the frame's slopes are measured, the integrator has not run in actuator space: do_control on demand)
        sup._err_stale, sup._control_pending = False, shortcut
        sup.iter += 1
        return state, r, False, ""

    def calculate_reward(self, reward_type=None):
        """ao_env.py:585-860, every branch that reads slopes, err, residual modes or the Strehl tuple
        (formulas: ao_marl_amd/rewards.py), the full-frame target image (on demand) or the phase projector; [nenv]."""
        from . import rewards as R
        sup = self.supervisor
        rt = self.reward_type if reward_type is None else reward_type
        if rt in R.STREHL:
            return R.strehl_reward(rt, sup.get_strehl())
        if rt in R.SLOPES:
            return R.slopes_reward(rt, sup.get_slopes())
        if rt == "residual_dm":                                                  # :603-605
            return -torch.linalg.vector_norm(sup.get_err(), dim=1)
        if rt == "avg_squared_modes_from_measurements":                          # :773-776
            proj = self._dev_matrix("projector_wfs2modes")                        # rows: the KEPT modes only
            rng = np.asarray(sup.obtain_action_range_modal())
            if rng.size and int(rng.max()) >= proj.shape[0]:                     # what NumPy raises in the reference
                raise IndexError("index %d is out of bounds for axis 0 with size %d" % (int(rng.max()), proj.shape[0]))
            m = sup.get_slopes() @ proj.T
            return -(m[:, self._action_range_t()] ** 2).sum(dim=1)
        if rt == "variance_actuators_filtered_from_modes":                       # :834-844
            m = sup.get_err() @ self._dev_matrix("volts2modes").T
            keep = torch.zeros(m.shape[1], dtype=torch.bool, device=m.device)
            keep[self._action_range_t()] = True
            c = (m * keep) @ self._dev_matrix("modes2volts").T
            return -c.var(dim=1, unbiased=False)
        if R.is_modes_reward(rt):
            return R.modes_reward(rt, self.transform_state_to_zernike(sup.get_err(), return_reward=True))
        if "counterfactual_rpc" in rt:                                           # :855-856: r = None
            return None
        if rt in R.IMAGE:                                                        # :621-623, :654-656
            return R.image_reward(rt, sup.get_tar_image(0))
        if rt in R.PROJECTION:                                                   # :736-760
            ph = sup.get_wfs_phase()                                             # [nenv, n, n], atmosphere + DMs
            ph = ph - ph.mean(dim=(1, 2), keepdim=True)
            pup = torch.as_tensor(sup.s.mpupil != 0, device=ph.device).reshape(-1)
            proj = -(ph.reshape(ph.shape[0], -1)[:, pup] @ self._dev_matrix("projector_phase2modes").T)
            cur = sup.get_voltages() @ self._dev_matrix("volts2modes").T
            fr = torch.as_tensor(sup.freedom_vector, dtype=torch.float32, device=ph.device) \
                if rt == "weighted_projection_comparison" else None
            return R.projection_reward(rt, proj, cur, self._action_range_t(), fr)
        raise NotImplementedError("This reward type not implemented")

    def _dev_matrix(self, name):
        """A host matrix of the supervisor (volts2modes, modes2volts, projector_wfs2modes) as a device
        tensor, uploaded on first use (only the non-default reward types read them)."""
        cache = self.__dict__.setdefault("_dev_mats", {})
        src = getattr(self.supervisor, name)
        if src is None:
            raise RuntimeError("%s is not available (obtain_and_set_cmat_filtered has not run)" % name)
        if name not in cache or cache[name][0] is not src:
            cache[name] = (src, torch.as_tensor(np.asarray(src, dtype=np.float32), device=self.device))
        return cache[name][1]

    def _action_range_t(self):
        return torch.as_tensor(np.asarray(self.supervisor.obtain_action_range_modal()), device=self.device)

    def rl_step(self, action, linear_control=False, apply_control=True, compute_tar_psf=True,
                compute_env_reward=False):
        """ao_env.py:911-939. The reference computes the env-level reward and the trainer throws
        it away (train_rpc.py:641); it is only evaluated here on request."""
        pair, out = None, None
        if (self.modal_shortcut and self._native_glue and self._default_state_layout and
                self._modal_valid and not linear_control):
            pair = (self._ring_slot(0), self._res_modes)
            out = self._ring_slot(-1)           # the slot the next linear_step makes the newest
        self.supervisor.next_part_two(action, linear_control=linear_control,
                                      apply_control=apply_control,
                                      compute_tar_psf=compute_tar_psf, modes_pair=pair, modes_out=out)
        self._ring_next_valid = self.supervisor.last_modes is not None
        self._modal_valid = False               # re-established by the next linear_step
        r = self.calculate_reward() if compute_env_reward else None
        return r, False, ""

    def divide_rewards_for_agents(self):
        """train_rpc.py:402-416 + helper_rewards.py:14-22: [nenv, n_agents].  get_err is read
        after next_part_two and before the next linear_step, i.e. it is the residual of the
        previous linear_step, whose modal projection is already at hand."""
        if self.layout is None:
            raise RuntimeError("no agent layout (set_agents)")
        r = self._last_res_modes
        if self._native_glue:
            from . import libaomarl as la
            return la.agent_rewards(r, self._lohi_i32, self._reward_factor)
        return (r * r) @ self._reward_mat

    def step(self, action, linear_control=False, out=None):
        """TrainerRPC.env_step (train_rpc.py:633-648): (s_next, per-agent reward, done, info).
        out = (state [nenv, state_dim], reward [nenv, n_agents]): contiguous float32 device tensors the step writes
        its results into (rows of a trajectory buffer: a training episode then needs no copy per step)."""
        if self._native_step_ok(linear_control):
            return self._step_native(action, out)
        _, done, info = self.rl_step(action, linear_control)
        r = self.divide_rewards_for_agents() if self.layout is not None else None
        s_next = self.linear_step()
        if self.supervisor.reset_prefetch is not None:
            self.supervisor.step_done()
        if out is not None:
            out[0].copy_(s_next)
            s_next = out[0]
            if r is not None and out[1] is not None:
                out[1].copy_(r)
                r = out[1]
        return s_next, r, done, info


class DelayedMDP(object):
    """Credit assignment under loop delay (environment/delayed_mdp.py:5-58): the tuple stored at
    step t is (s_{t-delay}, a_{t-delay}, s'_t); batched tensors instead of single vectors."""

    def __init__(self, delay, modification):
        self.delay = delay
        self.not_modification = int(not modification)
        self._n = delay + self.not_modification
        self.state_list, self.action_list, self.next_state_list = [], [], []

    def check_update_possibility(self):
        return len(self.action_list) >= self._n

    def save(self, s, a, s_next):
        for lst, v in ((self.state_list, s), (self.action_list, a), (self.next_state_list, s_next)):
            lst.append(v)
            if len(lst) > self._n:
                lst.pop(0)

    def credit_assignment(self):
        return self.state_list[0], self.action_list[0], self.next_state_list[-1]
