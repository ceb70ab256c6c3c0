"""HipSim: the batched, device-resident AO simulator (product path).

Owns one `aomarl_ctx` (static geometry on the GPU) and the per-environment state as torch tensors
whose `data_ptr()`s are handed to the C ABI (include/aomarl.h).  PyTorch is plumbing here: device
memory, streams -- every arithmetic step of the AO frame runs in the hand-written gfx950 kernels
of libaomarl_hip.so.  No CPU fallback exists: without the library or a GPU this raises.

Method names mirror the native objects the reference drives:
reset / move_atmos / raytrace_* / comp_image / do_centroids / do_control / set_com / rl_control /
apply_control / target_psf + comp_strehl, plus the two composites next_part_one / next_part_two
(rlSupervisor.py:1015-1051, 900-947).
"""
import ctypes as C

import numpy as np
import torch

from . import libaomarl as la
from . import system


def _ld4(n):
    return (n + 3) & ~3


def hip_calibration_id():
    """What names the arithmetic of a calibration through this backend (modal.calibrate's cache key): the library
    file (path, size, modification time) and its precision mode."""
    import os
    st = os.stat(la.LIB_PATH)
    return ("hip", os.path.abspath(la.LIB_PATH), int(st.st_size), int(st.st_mtime_ns), str(la.get_precision()))


class HipSim(object):
    def __init__(self, s, nenv, device="cuda:0", keep_bincube=False, keep_phase=False):
        if not torch.cuda.is_available():
            raise la.AomarlError("HipSim needs a GPU (torch.cuda.is_available() is False); the "
                                 "product path has no CPU fallback")
        self.lib = la.load()
        self.s = s
        self.nenv = int(nenv)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.keep_bincube, self.keep_phase = keep_bincube, keep_phase
        self.overlap_target = False
        self._side = None
        # stack-array DM phase straight from the commands inside the one-pass frame kernel when
        # the library can (next_part_two then skips materialising the shapes; any other consumer
        # of the shapes triggers _ensure_shape() first)
        self.defer_shape = True
        self.prefetch, self.pending_atmos = False, False     # see prefetch_atmos()
        self.graph_step = False      # aomarl_env_step replays HIP graphs (set_option("graph_step", 1))
        self._pending_range = (0, 0)
        self._s2m_rows = 0                                   # see set_slopes2modes()
        self._defer_on = False       # the ctx option as currently set
        self._stale = False          # st.dm_shape's stack-array planes are older than st.voltage
        self.ctx = C.c_void_p()
        self._create_ctx()
        self._alloc()
        if s.cmat is not None:
            self.set_cmat(s.cmat)

    # ------------------------------------------------------------------ plumbing
    def _create_ctx(self):
        desc, keep = la.make_desc(self.s)
        ctx = C.c_void_p()
        la.check(self.lib.aomarl_create(C.byref(desc), C.byref(ctx)))
        if self.ctx:
            self.lib.aomarl_destroy(self.ctx)
        self.ctx = ctx
        self._defer_on, self._stale = False, False
        del keep

    def _alloc(self):
        s, n, dev = self.s, self.nenv, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.ld_actu = _ld4(s.nactu)
        self.screen_stride = int(self.lib.aomarl_screen_stride(self.ctx))
        self.shape_stride = int(self.lib.aomarl_dmshape_stride(self.ctx))
        W = 2 * s.strehl_halfwin
        t = {}
        t["screens"] = torch.zeros(n, self.screen_stride, **f32)
        t["origin"] = torch.zeros(n, max(s.nscreens, 1), 2, **i32)
        t["seeds"] = torch.zeros(n, **i32)
        t["ext_count"] = torch.zeros(n, max(s.nscreens, 1), **i32)
        for k in ("com", "com1", "com2", "err", "voltage"):
            t[k] = torch.zeros(n, self.ld_actu, **f32)
        t["slopes"] = torch.zeros(n, s.nslope, **f32)
        t["dm_shape"] = torch.zeros(n, self.shape_stride, **f32)
        t["bincube"] = torch.zeros(n, s.nvalid, s.npix * s.npix, **f32) if self.keep_bincube \
            else None
        t["wfs_phase"] = torch.zeros(n, s.n, s.n, **f32) if self.keep_phase else None
        t["tar_phase"] = torch.zeros(n, s.pupdiam, s.pupdiam, **f32) if self.keep_phase else None
        t["strehl"] = torch.zeros(n, 8, **f32)
        t["le_img"] = torch.zeros(n, W * W, **f32)
        t["frame"] = torch.zeros(n, **i32)
        t["work"] = torch.zeros(int(self.lib.aomarl_workspace_floats(self.ctx, n)), **f32)
        self.t = t
        st = la.State()
        st.nenv, st.ld_actu = n, self.ld_actu
        for k, v in t.items():
            setattr(st, k, v.data_ptr() if v is not None else None)
        self.st = st
        self._st_ref = C.byref(st)
        self.accumx = np.zeros((n, max(s.nscreens, 1)), dtype=np.float32)
        self.accumy = np.zeros((n, max(s.nscreens, 1)), dtype=np.float32)

    def _stream(self):
        return la.raw_stream(self.device)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                torch.cuda.synchronize(self.device)
                self.lib.aomarl_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def _range(self, env_begin, env_count):
        if env_count is None:
            env_count = self.nenv - env_begin
        return env_begin, env_count

    # ------------------------------------------------------------------ views
    @property
    def com(self):
        return self.t["com"][:, :self.s.nactu]

    @property
    def err(self):
        return self.t["err"][:, :self.s.nactu]

    @property
    def voltage(self):
        return self._frame_buf("voltage")[:, :self.s.nactu]

    @property
    def slopes(self):
        return self._frame_buf("slopes")

    # ------------------------------------------------------------------ frame pipeline
    def enable_frame_pipeline(self, on=True):
        """aomarl_set_frame_pipeline: hand the library a twin of the state (own slopes / voltage / dm_shape /
        work, everything else shared) so that aomarl_env_step keeps one frame in flight -- frame t+1 beside
        the control / agent chain of frame t (loop delay of one frame only; the library takes the plain
        path whenever a step is not eligible).  While a frame is in flight only env_step and reset are
        accepted; `slopes` / `voltage` return the buffers of the last REDUCED frame."""
        if not on:
            la.check(self.lib.aomarl_set_frame_pipeline(self.ctx, C.byref(self.st), None))
            self._twin, self._twin_st = None, None
            return
        if getattr(self, "_twin", None) is not None:
            return
        f32 = dict(dtype=torch.float32, device=self.device)
        tw = {"voltage": torch.zeros(self.nenv, self.ld_actu, **f32),
              "slopes": torch.zeros(self.nenv, self.s.nslope, **f32),
              # only the tip-tilt slot of every environment is ever touched (stack-array DM from the voltages)
              "dm_shape": torch.empty(self.nenv, self.shape_stride, **f32),
              "work": torch.zeros(self.t["work"].numel(), **f32)}
        st2 = la.State()
        for name, _ in la.State._fields_:
            setattr(st2, name, getattr(self.st, name))
        for k, v in tw.items():
            setattr(st2, k, v.data_ptr())
        la.check(self.lib.aomarl_set_frame_pipeline(self.ctx, C.byref(self.st), C.byref(st2)))
        self._twin, self._twin_st = tw, st2

    def frame_pipeline_state(self):
        """(frame in flight, last reduced frame lives in the twin, pipelined steps, moves that ran beside a frame)"""
        a, b = C.c_int(0), C.c_int(0)
        n, o = C.c_ulonglong(0), C.c_ulonglong(0)
        la.check(self.lib.aomarl_frame_pipeline_state(self.ctx, C.byref(a), C.byref(b), C.byref(n), C.byref(o)))
        return bool(a.value), bool(b.value), int(n.value), int(o.value)

    def _frame_buf(self, name):
        if getattr(self, "_twin", None) is not None and self.frame_pipeline_state()[1]:
            return self._twin[name]
        return self.t[name]

    @property
    def strehl(self):
        """[nenv, 4]: SR SE, SR LE, phase variance, mean phase variance (targetCompass.py:139-159)."""
        s = self.t["strehl"]
        avg = torch.where(s[:, 4] > 0, s[:, 3] / s[:, 4].clamp(min=1), torch.zeros_like(s[:, 3]))
        return torch.stack([s[:, 0], s[:, 1], s[:, 2], avg], dim=1)

    @property
    def strehl_fit(self):
        """The same tuple with comp_strehl(do_fit=True), the reference's default: both Strehl ratios from the PSF
        peak fitted by two 1-D sincs (targetCompass.py:139-159; k_strehl_commit, strehl slots 6 / 7)."""
        la.check(self.lib.aomarl_strehl_fit(self.ctx, C.byref(self.st), 0, self.nenv, self._stream()))
        s = self.t["strehl"]
        avg = torch.where(s[:, 4] > 0, s[:, 3] / s[:, 4].clamp(min=1), torch.zeros_like(s[:, 3]))
        return torch.stack([s[:, 6], s[:, 7], s[:, 2], avg], dim=1)

    def dm_shape(self, k, env_begin=0, env_count=None):
        """Shape of DM k, [env_count, dim, dim] (materialised on demand: tip-tilt mirrors are
        never stored, their consumers evaluate the two planes on the fly)."""
        b, n = self._range(env_begin, env_count)
        self._ensure_shape()
        d = self.s.dms[k]
        out = torch.empty(n, d.dim, d.dim, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_get_dm_shape(self.ctx, C.byref(self.st), b, n, k,
                                              out.data_ptr(), self._stream()))
        return out

    def set_option(self, name, value):
        la.check(self.lib.aomarl_set_option(self.ctx, name.encode(), int(value)))
        if name == "prefetch_atmos":
            self.prefetch = bool(value)
        if name == "graph_step":
            self.graph_step = bool(value)
        if name == "force_unfused_frame" and value:
            self._set_defer(False)

    def dm_from_voltage_available(self):
        return bool(self.lib.aomarl_dm_from_voltage_available(self.ctx))

    def env_step(self, glue, action, gain, state_out, reward_out=None):
        """One TrainerRPC.env_step for every environment in ONE library call (aomarl_env_step):
        rl_control from Btt coordinates, apply_control, Strehl, per-agent rewards, next_part_one,
        v2m . err, state assembly.  `glue` is a libaomarl.EnvGlue the caller owns."""
        want = self.defer_shape and (self._defer_on or self.dm_from_voltage_available())
        if want != self._defer_on:
            self._set_defer(want)
        ptrs = self.__dict__.get("_accum_ptrs")
        if ptrs is None or ptrs[0] is not self.accumx or ptrs[1] is not self.accumy:    # (numpy -> ctypes costs ~1 us each)
            ptrs = self._accum_ptrs = (self.accumx, self.accumy, la.fptr(self.accumx), la.fptr(self.accumy))
        la.check(self.lib.aomarl_env_step(self.ctx, self._st_ref, glue._ref if hasattr(glue, "_ref") else C.byref(glue),
                                          action.data_ptr(), gain, ptrs[2], ptrs[3], state_out.data_ptr(),
                                          reward_out.data_ptr() if reward_out is not None else None,
                                          self._stream()))
        self._stale = self._defer_on
        if self.prefetch and not self.pending_atmos:
            self.pending_atmos, self._pending_range = True, (0, self.nenv)

    def policy_env_step(self, glue, desc, state, eps, seed, counter, gain, action, mean, state_out, reward_out=None):
        """aomarl_policy_env_step: the actors on `state` and the environment step with their action in ONE library call."""
        want = self.defer_shape and (self._defer_on or self.dm_from_voltage_available())
        if want != self._defer_on:
            self._set_defer(want)
        ptrs = self.__dict__.get("_accum_ptrs")
        if ptrs is None or ptrs[0] is not self.accumx or ptrs[1] is not self.accumy:
            ptrs = self._accum_ptrs = (self.accumx, self.accumy, la.fptr(self.accumx), la.fptr(self.accumy))
        la.check(self.lib.aomarl_policy_env_step(
                self.ctx, self._st_ref, glue._ref if hasattr(glue, "_ref") else C.byref(glue), C.byref(desc),
                state.data_ptr(), eps.data_ptr() if eps is not None else None, seed & 0xFFFFFFFF, counter & 0xFFFFFFFF,
                gain, ptrs[2], ptrs[3], action.data_ptr(), mean.data_ptr(), state_out.data_ptr(),
                reward_out.data_ptr() if reward_out is not None else None, self._stream()))
        self._stale = self._defer_on
        if self.prefetch and not self.pending_atmos:
            self.pending_atmos, self._pending_range = True, (0, self.nenv)

    def renew_frame_stream(self):
        """aomarl_set_option("renew_frame_stream"): the library's frame stream (frame pipeline) destroyed and created anew
        -- another hardware queue for it; nothing may be in flight (behind a full reset)."""
        la.check(self.lib.aomarl_set_option(self.ctx, b"renew_frame_stream", 1))

    def graph_stats(self):
        """(graphs captured, graphs replayed) by aomarl_env_step under set_option("graph_step", 1)."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        la.check(self.lib.aomarl_graph_stats(self.ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def _set_defer(self, on):
        on = bool(on)
        if on != self._defer_on:
            la.check(self.lib.aomarl_set_option(self.ctx, b"defer_dm_shape", int(on)))
            self._defer_on = on
        if not on:
            self._ensure_shape()

    def _ensure_shape(self):
        """Bring st.dm_shape's stack-array planes up to date with st.voltage."""
        if self._stale:
            la.check(self.lib.aomarl_materialize_dm_shape(self.ctx, C.byref(self.st), 0, self.nenv,
                                                          self._stream()))
            self._stale = False

    def screen(self, layer, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        d = self.s.screen_dim[layer]
        out = torch.empty(n, d, d, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_get_screen(self.ctx, C.byref(self.st), b, n, layer,
                                            out.data_ptr(), self._stream()))
        return out

    def set_screen(self, layer, screens, env_begin=0, env_count=None):
        """Overwrite one layer with logical screens [env_count, dim, dim] (ring origin reset)."""
        b, n = self._range(env_begin, env_count)
        d = self.s.screen_dim[layer]
        src = torch.as_tensor(screens, dtype=torch.float32, device=self.device).contiguous()
        if src.shape != (n, d, d):
            raise ValueError("screens must be [env_count, %d, %d]" % (d, d))
        la.check(self.lib.aomarl_set_screen(self.ctx, C.byref(self.st), b, n, layer,
                                            src.data_ptr(), self._stream()))

    # ------------------------------------------------------------------ configuration
    def set_cmat(self, cmat):
        if getattr(self, "_s2m_rows", 0):
            self.set_slopes2modes(None)          # v2m . cmat of the old matrix
        cmat = np.ascontiguousarray(cmat, dtype=np.float32)
        if cmat.shape != (self.s.nactu, self.s.nslope):
            raise ValueError("cmat must be [nactu, nslope]")
        la.check(self.lib.aomarl_set_cmat(self.ctx, la.fptr(cmat)))

    def set_gain(self, gain):
        la.check(self.lib.aomarl_set_gain(self.ctx, float(gain)))

    def set_env_gains(self, gains):
        """One integrator gain per environment ([nenv]) for do_control, or None for the scalar
        again (aomarl_set_env_gains)."""
        if gains is None:
            la.check(self.lib.aomarl_set_env_gains(self.ctx, None, 0))
            return
        g = np.ascontiguousarray(gains, dtype=np.float32).reshape(-1)
        if g.size != self.nenv:
            raise ValueError("one gain per environment: expected %d, got %d" % (self.nenv, g.size))
        la.check(self.lib.aomarl_set_env_gains(self.ctx, la.fptr(g), int(g.size)))

    def set_modal(self, v2m, m2v, freedom=None, action_modes=None):
        v2m = np.ascontiguousarray(v2m, dtype=np.float32)
        m2v = np.ascontiguousarray(m2v, dtype=np.float32)
        nm = v2m.shape[0]
        if v2m.shape != (nm, self.s.nactu) or m2v.shape != (self.s.nactu, nm):
            raise ValueError("v2m must be [nmodes, nactu] and m2v [nactu, nmodes]")
        fr = np.zeros(nm, dtype=np.float32) if freedom is None else \
            np.ascontiguousarray(freedom, dtype=np.float32)
        am = np.zeros(0, dtype=np.int32) if action_modes is None else \
            np.ascontiguousarray(np.asarray(action_modes) % nm, dtype=np.int32)
        la.check(self.lib.aomarl_set_modal(self.ctx, nm, la.fptr(v2m), la.fptr(m2v), la.fptr(fr),
                                           int(am.size), la.iptr(am) if am.size else None))
        self.nmodes, self.nact = nm, int(am.size)

    @staticmethod
    def calibration_id():
        return hip_calibration_id()

    def reload_dms(self):
        """After actuator filtering changed s.dms: rebuild the static description + state."""
        torch.cuda.synchronize(self.device)
        self._s2m_rows = 0
        self._create_ctx()
        self._alloc()

    # ------------------------------------------------------------------ per-frame API
    # ------------------------------------------------------------------ prefetched reset
    def prefetch_reset_begin(self, seeds):
        """aomarl_reset_prefetch_begin for the whole batch: the screens of the NEXT reset (with these seeds) start
        growing in a shadow state on a stream of their own; prefetch_reset_advance(k) runs k more of the
        2 x dim rounds -- call it once per step of the running episode --, and reset(seeds) with the same seeds
        adopts them (runs what is left first).  Any other reset drops the prefetch."""
        n = self.nenv
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.int64) & 0xFFFFFFFF, (n,)), dtype=np.uint32)
        if getattr(self, "_rp", None) is None:
            f32, i32 = dict(dtype=torch.float32, device=self.device), dict(dtype=torch.int32, device=self.device)
            t = {"screens": torch.zeros(n, self.screen_stride, **f32), "origin": torch.zeros_like(self.t["origin"]),
                 "seeds": torch.zeros(n, **i32), "ext_count": torch.zeros_like(self.t["ext_count"]),
                 "frame": torch.zeros(n, **i32), "work": torch.zeros(self.t["work"].numel(), **f32)}
            for k in ("com", "com1", "com2", "err", "voltage"):
                t[k] = torch.zeros(n, self.ld_actu, **f32)
            st2 = la.State()
            for name, _ in la.State._fields_:
                setattr(st2, name, getattr(self.st, name))       # (slopes, dm_shape, Strehl ...: never touched through it)
            for k, v in t.items():
                setattr(st2, k, v.data_ptr())
            # rp_own_stream: the rounds on a stream of their own instead of the library's low-priority side stream
            # (one more hardware queue: measured, it can alias with the frame stream and serialise the pipelined order)
            own = torch.cuda.Stream(device=self.device) if getattr(self, "rp_own_stream", False) else None
            self._rp = dict(t=t, st=st2, stream=own, seeds=None, left=0)
            # the shadow's buffers were zero-filled on the caller's stream just now; the rounds start on another one
            torch.cuda.current_stream(self.device).synchronize()
        rp = self._rp
        if rp["stream"] is not None:
            rp["stream"].wait_stream(torch.cuda.current_stream(self.device))
        la.check(self.lib.aomarl_reset_prefetch_begin(self.ctx, C.byref(rp["st"]), 0, n, la.uptr(seeds), self._rp_stream()))
        rp["seeds"], rp["left"] = seeds.copy(), -1

    def _rp_stream(self):
        st = self._rp["stream"]
        return C.c_void_p(st.cuda_stream) if st is not None else None

    def prefetch_reset_advance(self, nrounds):
        """nrounds more rounds of the begun prefetch (returns the rounds left; 0 when none was begun)."""
        rp = getattr(self, "_rp", None)
        if rp is None or rp["seeds"] is None or rp["left"] == 0:
            return 0
        left = C.c_int(0)
        la.check(self.lib.aomarl_reset_prefetch_advance(self.ctx, int(nrounds), self._rp_stream(), C.byref(left)))
        rp["left"] = int(left.value)
        return rp["left"]

    def prefetch_reset_pending(self, seeds=None):
        """True when a prefetch has been begun (for these seeds, if given) and not yet adopted."""
        rp = getattr(self, "_rp", None)
        if rp is None or rp["seeds"] is None:
            return False
        if seeds is None:
            return True
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.int64) & 0xFFFFFFFF, (self.nenv,)), dtype=np.uint32)
        return bool(np.array_equal(seeds, rp["seeds"]))

    def reset(self, seeds, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        seeds = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.int64) &
                                                     0xFFFFFFFF, (n,)), dtype=np.uint32)
        rp = getattr(self, "_rp", None)
        if rp is not None and rp["seeds"] is not None:
            if (b, n) == (0, self.nenv) and np.array_equal(seeds, rp["seeds"]):
                la.check(self.lib.aomarl_reset_adopt(self.ctx, C.byref(self.st), b, n, la.uptr(seeds),
                                                     la.fptr(self.accumx), la.fptr(self.accumy),
                                                     self._rp_stream(), self._stream()))
                self.prefetched_resets = getattr(self, "prefetched_resets", 0) + 1
            else:                               # other seeds / a part of the batch: the prefetch is of no use
                la.check(self.lib.aomarl_reset_prefetch_cancel(self.ctx))
                la.check(self.lib.aomarl_reset(self.ctx, C.byref(self.st), b, n, la.uptr(seeds),
                                               la.fptr(self.accumx), la.fptr(self.accumy), self._stream()))
            rp["seeds"], rp["left"] = None, 0
        else:
            la.check(self.lib.aomarl_reset(self.ctx, C.byref(self.st), b, n, la.uptr(seeds),
                                           la.fptr(self.accumx), la.fptr(self.accumy),
                                           self._stream()))
        # the library drops a prefetched frame only when the reset covers its range (a reset of a
        # range disjoint from it leaves it pending, one that cuts into it is refused above)
        if self.pending_atmos:
            pb, pn = self._pending_range
            if b <= pb and b + n >= pb + pn:
                self.pending_atmos = False
        if (b, n) == (0, self.nenv):
            self._stale = False             # commands, voltages and shapes are all zero again

    def move_atmos(self, env_begin=0, env_count=None):
        """Atmos.move_atmos; after `prefetch_atmos` of the same range: only waits for that move."""
        b, n = self._range(env_begin, env_count)
        la.check(self.lib.aomarl_move_atmos(self.ctx, C.byref(self.st), b, n,
                                            la.fptr(self.accumx), la.fptr(self.accumy),
                                            self._stream()))
        self.pending_atmos = False

    def prefetch_atmos(self, env_begin=0, env_count=None):
        """Issue the NEXT frame's move_atmos now, on the library's side stream, behind everything
        enqueued so far (aomarl_prefetch_atmos): call it once the kernels that read this frame's
        screens are enqueued.  The screens are one frame ahead until the next `move_atmos`."""
        b, n = self._range(env_begin, env_count)
        la.check(self.lib.aomarl_prefetch_atmos(self.ctx, C.byref(self.st), b, n,
                                                la.fptr(self.accumx), la.fptr(self.accumy),
                                                self._stream()))
        self.pending_atmos, self._pending_range = True, (b, n)

    # ------------------------------------------------------------------ run-time wind / r0 (atmosCompass.py:79-135)
    def layer_values(self, layer):
        """(deltax, deltay, amplitude) of a layer as the library holds them now."""
        v = [C.c_float(0.) for _ in range(3)]
        la.check(self.lib.aomarl_get_layer(self.ctx, int(layer), *[C.byref(x) for x in v]))
        return tuple(float(x.value) for x in v)

    def atmos_change_blocked(self):
        """Why a change of wind / r0 would not take effect on the NEXT frame, or None: a move that is already issued
        (prefetch_atmos, a pipelined frame in flight) keeps the atmosphere it was planned with."""
        if self.pending_atmos:
            return "the next frame's atmosphere is already moved (prefetch_atmos)"
        if self.frame_pipeline_state()[0]:
            return "a frame is in flight (frame pipeline)"
        return None

    def _cancel_reset_prefetch(self):
        rp = getattr(self, "_rp", None)
        if rp is not None and rp["seeds"] is not None:
            la.check(self.lib.aomarl_reset_prefetch_cancel(self.ctx))
            rp["seeds"], rp["left"] = None, 0
            return True
        return False

    def set_wind(self, layer, deltax, deltay, mirror_stencils=True):
        """aomarl_set_wind: layer `layer` moves by (deltax, deltay) pixels per frame from the next PLANNED move on;
        where a component changes sign its stencil is mirrored (atmosCompass.py:124-135).  A prefetched reset (grown
        along the old sign) is dropped.  Returns True when one was dropped."""
        dropped = self._cancel_reset_prefetch()
        la.check(self.lib.aomarl_set_wind(self.ctx, int(layer), float(np.float32(deltax)), float(np.float32(deltay)),
                                          int(bool(mirror_stencils))))
        return dropped

    def set_stencil(self, layer, axis, istencil):
        """aomarl_set_stencil: Tscreen.set_istencilx (axis 0) / set_istencily (axis 1), flat logical indices."""
        self._cancel_reset_prefetch()
        ist = np.ascontiguousarray(istencil, dtype=np.uint32)
        la.check(self.lib.aomarl_set_stencil(self.ctx, int(layer), int(axis), la.uptr(ist), int(ist.size)))

    def set_amplitudes(self, amplitude):
        """aomarl_set_r0: the noise amplitude of every layer's new lines (um); the screens as they stand are kept."""
        dropped = self._cancel_reset_prefetch()
        a = np.ascontiguousarray(amplitude, dtype=np.float32).reshape(-1)
        la.check(self.lib.aomarl_set_r0(self.ctx, la.fptr(a), int(a.size)))
        return dropped

    def extrude(self, layers, dirs, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        l = np.ascontiguousarray(layers, dtype=np.int32)
        d = np.ascontiguousarray(dirs, dtype=np.int32)
        la.check(self.lib.aomarl_extrude(self.ctx, C.byref(self.st), b, n, int(l.size),
                                         la.iptr(l), la.iptr(d), self._stream()))

    def _need_phase(self):
        if self.t["wfs_phase"] is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            self.t["wfs_phase"] = torch.zeros(self.nenv, self.s.n, self.s.n, **f32)
            self.t["tar_phase"] = torch.zeros(self.nenv, self.s.pupdiam, self.s.pupdiam, **f32)
            self.st.wfs_phase = self.t["wfs_phase"].data_ptr()
            self.st.tar_phase = self.t["tar_phase"].data_ptr()

    def _need_bincube(self):
        if self.t["bincube"] is None:
            self.t["bincube"] = torch.zeros(self.nenv, self.s.nvalid, self.s.npix * self.s.npix,
                                            dtype=torch.float32, device=self.device)
            self.st.bincube = self.t["bincube"].data_ptr()

    def raytrace_wfs(self, atm=True, dms=True, reset=True, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        self._need_phase()
        self._ensure_shape()
        fl = (la.TRACE_ATMOS if atm else 0) | (la.TRACE_DMS if dms else 0) | \
            (la.TRACE_RESET if reset else 0)
        la.check(self.lib.aomarl_raytrace_wfs(self.ctx, C.byref(self.st), b, n, fl,
                                              self._stream()))

    def raytrace_target(self, atm=True, dms=True, reset=True, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        self._need_phase()
        self._ensure_shape()
        fl = (la.TRACE_ATMOS if atm else 0) | (la.TRACE_DMS if dms else 0) | \
            (la.TRACE_RESET if reset else 0)
        la.check(self.lib.aomarl_raytrace_target(self.ctx, C.byref(self.st), b, n, fl,
                                                 self._stream()))

    def comp_image(self, from_phase_buffer=False, noise=True, write_bincube=False, cog=True,
                   atm=True, dms=True, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        fl = 0
        if from_phase_buffer:
            self._need_phase()
            fl |= la.IMG_FROM_PHASE_BUFFER
        if noise:
            fl |= la.IMG_NOISE
        if write_bincube:
            self._need_bincube()
            fl |= la.IMG_WRITE_BINCUBE
        if cog:
            fl |= la.IMG_COG
        if not atm:
            fl |= la.IMG_NO_ATMOS
        if not dms:
            fl |= la.IMG_NO_DMS
        self._ensure_shape()
        la.check(self.lib.aomarl_comp_image(self.ctx, C.byref(self.st), b, n, fl, self._stream()))

    def do_centroids(self, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        la.check(self.lib.aomarl_do_centroids(self.ctx, C.byref(self.st), b, n, self._stream()))

    def slopes_geom(self, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        la.check(self.lib.aomarl_slopes_geom(self.ctx, C.byref(self.st), b, n, self._stream()))

    def do_control(self, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        if getattr(self, "_twin", None) is not None and (b, n) == (0, self.nenv):
            # (frame pipeline: the last REDUCED frame's slopes live in the state or in its twin by parity -- the library
            # picks the view: a do_control the residual shortcut left to be run on demand)
            la.check(self.lib.aomarl_do_control_reduced(self.ctx, C.byref(self.st), self._stream()))
            return
        la.check(self.lib.aomarl_do_control(self.ctx, C.byref(self.st), b, n, self._stream()))

    def set_com(self, com, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        com = torch.as_tensor(com, dtype=torch.float32, device=self.device).contiguous()
        if com.shape != (n, self.s.nactu):
            raise ValueError("Dimension mismatch")   # rtcCompass.py:471-472
        la.check(self.lib.aomarl_set_com(self.ctx, C.byref(self.st), b, n, com.data_ptr(),
                                         self._stream()))

    def rl_control(self, action, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        action = torch.as_tensor(action, dtype=torch.float32, device=self.device).contiguous()
        if action.shape != (n, self.nact):
            raise ValueError("action must be [env_count, %d]" % self.nact)
        la.check(self.lib.aomarl_rl_control(self.ctx, C.byref(self.st), b, n, action.data_ptr(),
                                            self._stream()))

    def rl_control_modes(self, m0, m1, g, action=None, env_begin=0, env_count=None, out=None):
        """rl_control from known Btt coordinates: modes = m0 + g m1 (+ action); com = m2v . modes.
        Returns the modes ([env_count, nmodes]) -- they are v2m . com of the new command."""
        b, n = self._range(env_begin, env_count)
        m0, m1 = m0.contiguous(), m1.contiguous()
        if m0.shape != (n, self.nmodes) or m1.shape != (n, self.nmodes):
            raise ValueError("modal vectors must be [env_count, %d]" % self.nmodes)
        ptr = None
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device).contiguous()
            if action.shape != (n, self.nact):
                raise ValueError("action must be [env_count, %d]" % self.nact)
            ptr = action.data_ptr()
        if out is None:
            out = torch.empty(n, self.nmodes, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_rl_control_modes(self.ctx, C.byref(self.st), b, n, m0.data_ptr(),
                                                  m1.data_ptr(), float(g), ptr, out.data_ptr(),
                                                  self._stream()))
        return out

    def apply_control(self, comp_voltage=True, env_begin=0, env_count=None, defer_shape=False):
        """defer_shape: leave the stack-array shapes to the one-pass frame kernel (which then
        evaluates them from st.voltage); only honoured when the library supports it."""
        b, n = self._range(env_begin, env_count)
        fl = la.APPLY_COMP_VOLTAGE if comp_voltage else 0
        defer = bool(defer_shape) and self.dm_from_voltage_available()
        if defer:
            fl |= la.APPLY_DEFER_STACK_SHAPE
        elif (b, n) != (0, self.nenv):
            self._ensure_shape()              # a partial refresh must not hide older stale rows
        la.check(self.lib.aomarl_apply_control(self.ctx, C.byref(self.st), b, n, fl,
                                               self._stream()))
        self._stale = defer or (self._stale and (b, n) != (0, self.nenv))

    def comp_dm_shape(self, volts=None, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        ptr = None
        if volts is not None:
            volts = torch.as_tensor(volts, dtype=torch.float32, device=self.device).contiguous()
            if volts.shape != (n, self.s.nactu):
                raise ValueError("volts must be [env_count, nactu]")
            ptr = volts.data_ptr()
            # the shapes no longer follow st.voltage: the composites must read them from memory
            self._set_defer(False)
        if (b, n) != (0, self.nenv):
            self._ensure_shape()
        la.check(self.lib.aomarl_comp_dm_shape(self.ctx, C.byref(self.st), b, n, ptr,
                                               self._stream()))
        if (b, n) == (0, self.nenv):
            self._stale = False

    def target_psf(self, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        if not all(float(o).is_integer() for t in (self.s.tar_atm_off + self.s.tar_dm_off)
                   for o in t):
            self._need_phase()
        self._ensure_shape()
        la.check(self.lib.aomarl_target_psf(self.ctx, C.byref(self.st), b, n, self._stream()))

    def target_image(self, env_begin=0, env_count=None):
        """Target.get_tar_image(expo_type="se") (targetCompass.py:71-92) of every environment of the range:
        [env_count, npsf, npsf], centred, raw |FFT2|^2 (aomarl_target_image: on demand, one environment at a time)."""
        b, n = self._range(env_begin, env_count)
        out = torch.empty(n, self.s.npsf, self.s.npsf, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_target_image(self.ctx, C.byref(self.st), b, n, out.data_ptr(), self._stream()))
        return out

    def frame_fused_available(self):
        return bool(self.lib.aomarl_frame_fused_available(self.ctx))

    def frame_fused(self, noise=True, write_bincube=False, cog=True, env_begin=0, env_count=None,
                    dm_from_voltage=None):
        """target_psf + comp_image from one pass over the phase (aomarl_frame_fused).
        dm_from_voltage: evaluate the stack-array DM from st.voltage inside the kernel (default:
        exactly when the stored shapes are stale, i.e. after a deferred apply_control)."""
        b, n = self._range(env_begin, env_count)
        if dm_from_voltage is None:
            dm_from_voltage = self._stale
        fl = 0
        if dm_from_voltage:
            fl |= la.IMG_DM_FROM_VOLTAGE
        else:
            self._ensure_shape()
        if noise:
            fl |= la.IMG_NOISE
        if write_bincube:
            self._need_bincube()
            fl |= la.IMG_WRITE_BINCUBE
        if cog:
            fl |= la.IMG_COG
        la.check(self.lib.aomarl_frame_fused(self.ctx, C.byref(self.st), b, n, fl, self._stream()))

    def frame_kernel_name(self):
        """Instantiation of the one-pass frame kernel the last frame_fused launched, as rocprofv3
        prints it (aomarl_frame_kernel_name)."""
        return self.lib.aomarl_frame_kernel_name(self.ctx).decode()

    def frame_kernel_time(self):
        """(sum of launch durations in ms, launches) recorded under set_option("time_frame_kernel",
        n) since the last call (aomarl_frame_kernel_time)."""
        tot, n = C.c_double(0.0), C.c_int(0)
        la.check(self.lib.aomarl_frame_kernel_time(self.ctx, C.byref(tot), C.byref(n)))
        return tot.value, n.value

    def comp_strehl(self, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        la.check(self.lib.aomarl_comp_strehl(self.ctx, C.byref(self.st), b, n, self._stream()))

    def reset_strehl(self, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        la.check(self.lib.aomarl_reset_strehl(self.ctx, C.byref(self.st), b, n, self._stream()))

    def set_slopes2modes(self, s2m):
        """s2m = v2m . cmat ([nmodes, nslope]) for `slopes2modes`, or None to drop it."""
        if s2m is None:
            la.check(self.lib.aomarl_set_slopes2modes(self.ctx, 0, None))
            self._s2m_rows = 0
            return
        s2m = np.ascontiguousarray(s2m, dtype=np.float32)
        if s2m.ndim != 2 or s2m.shape[1] != self.s.nslope:
            raise ValueError("s2m must be [nmodes, nslope]")
        la.check(self.lib.aomarl_set_slopes2modes(self.ctx, int(s2m.shape[0]), la.fptr(s2m)))
        self._s2m_rows = int(s2m.shape[0])

    def slopes2modes(self, env_begin=0, env_count=None):
        """-(v2m . cmat) . slopes: the Btt coordinates of err without err (aomarl_slopes2modes)."""
        b, n = self._range(env_begin, env_count)
        out = torch.empty(n, self._s2m_rows, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_slopes2modes(self.ctx, C.byref(self.st), b, n, out.data_ptr(),
                                              self._stream()))
        return out

    def volts2modes(self, vec, out=None):
        """[rows, nactu] (any row stride, e.g. the padded views `com` / `err`) -> [rows, nmodes]."""
        if vec.stride(1) != 1:
            vec = vec.contiguous()
        if out is None:
            out = torch.empty(vec.shape[0], self.nmodes, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_volts2modes(self.ctx, C.byref(self.st), vec.shape[0],
                                             vec.data_ptr(), vec.stride(0), out.data_ptr(),
                                             self._stream()))
        return out

    def next_part_one(self, write_bincube=False, env_begin=0, env_count=None, overlap=None):
        """move_atmos -> {target trace + PSF  ||  WFS trace + image + COG -> do_control}.
        The science path and the WFS path only read the screens and DM shapes, so with
        `overlap` (default: self.overlap_target) the memory-bound PSF kernels run on a side HIP
        stream next to the matrix-pipe-bound spot kernel and are joined before returning."""
        b, n = self._range(env_begin, env_count)
        if overlap is None:
            overlap = self.overlap_target
        # stack-array shapes left to the frame kernel by a deferred apply_control: let the
        # composite evaluate them from the voltages; otherwise it reads them from memory
        if self._stale and self.defer_shape and self.dm_from_voltage_available():
            if not self._defer_on:
                la.check(self.lib.aomarl_set_option(self.ctx, b"defer_dm_shape", 1))
                self._defer_on = True
        else:
            self._set_defer(False)
        if not overlap:
            fl = 0
            if write_bincube:
                self._need_bincube()
                fl |= la.IMG_WRITE_BINCUBE
            la.check(self.lib.aomarl_next_part_one(self.ctx, C.byref(self.st), b, n,
                                                   la.fptr(self.accumx), la.fptr(self.accumy), fl,
                                                   self._stream()))
            # the library prefetches for one range at a time (a second range steps in plain order)
            if self.prefetch and not self.pending_atmos:
                self.pending_atmos, self._pending_range = True, (b, n)
            return
        self.move_atmos(b, n)
        self.target_and_wfs(write_bincube=write_bincube, env_begin=b, env_count=n)
        self.do_control(b, n)

    def target_and_wfs(self, write_bincube=False, noise=True, cog=True, env_begin=0,
                       env_count=None):
        """target_psf on the side stream, comp_image on the current one, joined at the end."""
        b, n = self._range(env_begin, env_count)
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self.target_psf(b, n)
        self.comp_image(noise=noise, write_bincube=write_bincube, cog=cog, env_begin=b,
                        env_count=n)
        main.wait_stream(self._side)

    def next_part_two(self, action=None, env_begin=0, env_count=None):
        b, n = self._range(env_begin, env_count)
        ptr = None
        if action is not None:
            action = torch.as_tensor(action, dtype=torch.float32, device=self.device).contiguous()
            if action.shape != (n, self.nact):
                raise ValueError("action must be [env_count, %d]" % self.nact)
            ptr = action.data_ptr()
        self._set_defer(self.defer_shape and self.dm_from_voltage_available())
        if not self._defer_on and (b, n) != (0, self.nenv):
            self._ensure_shape()
        la.check(self.lib.aomarl_next_part_two(self.ctx, C.byref(self.st), b, n, ptr,
                                               self._stream()))
        self._stale = self._defer_on or (self._stale and (b, n) != (0, self.nenv))

    def gemm_nt(self, A, B, alpha=1.0, beta=0.0, Cout=None):
        """C = alpha * A @ B.T + beta * C on the library's fp32 MFMA GEMM (tests)."""
        A, B = A.contiguous(), B.contiguous()
        M, K = A.shape
        N = B.shape[0]
        if Cout is None:
            Cout = torch.zeros(M, N, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_gemm_nt(M, N, K, alpha, A.data_ptr(), A.stride(0), B.data_ptr(),
                                         B.stride(0), beta, Cout.data_ptr(), Cout.stride(0),
                                         self._stream()))
        return Cout

    # ------------------------------------------------------------------ geometric controller
    def geo_twin(self, IF):
        """A GEO twin of this simulator (see HipGeoTwin)."""
        return HipGeoTwin(self, IF)

    # ------------------------------------------------------------------ calibration backend
    def dm_response(self, commands, geometric):
        """slopes [K, nslope] for commands [K, nactu], atmosphere off (modal.calibrate backend)."""
        commands = np.ascontiguousarray(commands, dtype=np.float32)
        K = commands.shape[0]
        out = np.zeros((K, self.s.nslope), dtype=np.float32)
        for k0 in range(0, K, self.nenv):
            n = min(self.nenv, K - k0)
            v = torch.from_numpy(commands[k0:k0 + n]).to(self.device)
            self.comp_dm_shape(v, 0, n)
            if geometric:
                self.raytrace_wfs(atm=False, dms=True, reset=True, env_begin=0, env_count=n)
                self.slopes_geom(0, n)
            else:
                self.comp_image(noise=False, cog=True, atm=False, dms=True, env_begin=0,
                                env_count=n)
            out[k0:k0 + n] = self.slopes[:n].cpu().numpy()
        return out


class HipGeoTwin(object):
    """The geometric ("GEO") reference controller next to a HipSim: controller 1 / DMs 1, 3 /
    target 1 of the reference's non-noise parameter files (rlSupervisor.py:989-1013,
    rtc_init.py:418-448).  Same geometry as the main loop's mirrors and target (the parameter
    files duplicate them), same atmosphere: a second aomarl_state that SHARES the main state's
    screens and owns its commands, DM shapes, science phase and Strehl accumulators."""

    def __init__(self, sim, IF):
        from . import modal
        self.sim, self.lib, self.ctx = sim, sim.lib, sim.ctx
        s, n, dev = sim.s, sim.nenv, sim.device
        self.s, self.nenv, self.device = s, n, dev
        W = np.ascontiguousarray(modal.geo_projector(IF), dtype=np.float32)
        if W.shape != (s.nactu, s.nactu + 1):
            raise ValueError("influence matrix does not match the system's actuators")
        la.check(self.lib.aomarl_set_geo(self.ctx, la.fptr(W)))
        f32 = dict(dtype=torch.float32, device=dev)
        Wn = 2 * s.strehl_halfwin
        t = {}
        for k in ("com", "com1", "com2", "err", "voltage"):
            t[k] = torch.zeros(n, sim.ld_actu, **f32)
        t["slopes"] = torch.zeros(n, s.nslope, **f32)
        t["dm_shape"] = torch.zeros(n, sim.shape_stride, **f32)
        t["tar_phase"] = torch.zeros(n, s.pupdiam, s.pupdiam, **f32)
        t["strehl"] = torch.zeros(n, 8, **f32)
        t["le_img"] = torch.zeros(n, Wn * Wn, **f32)
        t["frame"] = torch.zeros(n, dtype=torch.int32, device=dev)
        t["work"] = torch.zeros(int(self.lib.aomarl_workspace_floats(self.ctx, n)), **f32)
        self.t = t
        self.gwork = torch.zeros(int(self.lib.aomarl_geo_workspace_floats(self.ctx, n)), **f32)
        st = la.State()
        st.nenv, st.ld_actu = n, sim.ld_actu
        for k in ("screens", "origin", "seeds", "ext_count"):          # shared atmosphere
            setattr(st, k, sim.t[k].data_ptr())
        for k, v in t.items():
            setattr(st, k, v.data_ptr())
        self.st = st

    def _stream(self):
        return la.raw_stream(self.device)

    @property
    def com(self):
        return self.t["com"][:, :self.s.nactu]

    @property
    def strehl(self):
        s = self.t["strehl"]
        avg = torch.where(s[:, 4] > 0, s[:, 3] / s[:, 4].clamp(min=1), torch.zeros_like(s[:, 3]))
        return torch.stack([s[:, 0], s[:, 1], s[:, 2], avg], dim=1)

    @property
    def strehl_fit(self):
        la.check(self.lib.aomarl_strehl_fit(self.ctx, C.byref(self.st), 0, self.sim.nenv, self._stream()))
        s = self.t["strehl"]
        avg = torch.where(s[:, 4] > 0, s[:, 3] / s[:, 4].clamp(min=1), torch.zeros_like(s[:, 3]))
        return torch.stack([s[:, 6], s[:, 7], s[:, 2], avg], dim=1)

    def reset(self):
        for k in ("com", "com1", "com2", "err", "voltage", "dm_shape", "strehl", "le_img"):
            self.t[k].zero_()

    def next_part_one_geo(self, env_begin=0, env_count=None):
        """target.raytrace(atmosphere) -> do_control(sources=target) -> apply_control (delay 0)
        -> target.raytrace(dms, reset=False) -> pending PSF  (rlSupervisor.py:989-1013)."""
        b, n = self.sim._range(env_begin, env_count)
        L, st, sm = self.lib, C.byref(self.st), self._stream()
        la.check(L.aomarl_raytrace_target(self.ctx, st, b, n,
                                          la.TRACE_ATMOS | la.TRACE_RESET | la.TRACE_MASK, sm))
        la.check(L.aomarl_geo_control(self.ctx, st, b, n, self.gwork.data_ptr(), sm))
        la.check(L.aomarl_apply_control(self.ctx, st, b, n, 0, sm))       # voltage = com, shapes
        la.check(L.aomarl_raytrace_target(self.ctx, st, b, n, la.TRACE_DMS, sm))
        la.check(L.aomarl_target_psf_buffer(self.ctx, st, b, n, sm))

    def comp_strehl(self, env_begin=0, env_count=None):
        b, n = self.sim._range(env_begin, env_count)
        la.check(self.lib.aomarl_comp_strehl(self.ctx, C.byref(self.st), b, n, self._stream()))

    def target_image(self, env_begin=0, env_count=None):
        """Target.get_tar_image(1, "se") (targetCompass.py:71-92): the full npsf x npsf image of the geometric
        controller's target -- the frame's atmosphere (the shared screens) + this twin's mirrors, as next_part_one_geo
        left them: [env_count, npsf, npsf], centred, raw |FFT2|^2 (aomarl_target_image on the twin's state)."""
        b, n = self.sim._range(env_begin, env_count)
        out = torch.empty(n, self.s.npsf, self.s.npsf, dtype=torch.float32, device=self.device)
        la.check(self.lib.aomarl_target_image(self.ctx, C.byref(self.st), b, n, out.data_ptr(), self._stream()))
        return out
