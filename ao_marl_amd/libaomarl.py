"""ctypes binding of the C ABI in include/aomarl.h (libaomarl_hip.so, built in-tree by
ao_marl_amd/csrc/Makefile or __graft_entry__.build()).

There is NO fallback: if the shared library is missing or a symbol is absent, importing the
product path raises.  The CPU oracle under oracle/ is never used from here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# AOMARL_LIB: another build of the same library (A/B measurements of kernel variants, csrc/Makefile `variant`)
LIB_PATH = os.environ.get("AOMARL_LIB") or os.path.join(HERE, "libaomarl_hip.so")
MAX_LAYERS, MAX_DMS, ABI_VERSION = 8, 4, 2
PRECISION_F32, PRECISION_SPLIT_F16 = 0, 1

DM_PZT, DM_TT = 0, 1
TRACE_ATMOS, TRACE_DMS, TRACE_RESET, TRACE_MASK = 1, 2, 4, 8
IMG_FROM_PHASE_BUFFER, IMG_NOISE, IMG_WRITE_BINCUBE, IMG_COG, IMG_NO_ATMOS, IMG_NO_DMS = \
    1, 2, 4, 8, 16, 32
IMG_DM_FROM_VOLTAGE = 64
APPLY_COMP_VOLTAGE, APPLY_DEFER_STACK_SHAPE = 1, 2

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)


class DmDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("dim", C.c_int32), ("nact", C.c_int32),
                ("influsize", C.c_int32), ("ninflupos", C.c_int64), ("influ", _fp),
                ("influpos", _ip), ("ninflu", _ip), ("influstart", _ip),
                ("wfs_xoff", C.c_float), ("wfs_yoff", C.c_float), ("tar_xoff", C.c_float),
                ("tar_yoff", C.c_float)]


class LayerDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nstencil", C.c_int32), ("A", _fp), ("B", _fp),
                ("istx", _up), ("isty", _up), ("deltax", C.c_float), ("deltay", C.c_float),
                ("amplitude", C.c_float), ("wfs_xoff", C.c_float), ("wfs_yoff", C.c_float),
                ("tar_xoff", C.c_float), ("tar_yoff", C.c_float)]


class Desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n", C.c_int32), ("pupdiam", C.c_int32),
                ("mpupil", _fp), ("spupil", _fp),
                ("nvalid", C.c_int32), ("pdiam", C.c_int32), ("nfft", C.c_int32),
                ("npix", C.c_int32), ("nrebin", C.c_int32), ("nxsub", C.c_int32),
                ("phasemap", _ip), ("halfxy", _fp), ("binmap", _ip), ("flux", _fp),
                ("validsubsx", _ip), ("validsubsy", _ip),
                ("nphot", C.c_float), ("wfs_lambda", C.c_float), ("noise", C.c_float),
                ("cog_offset", C.c_float), ("cog_scale", C.c_float), ("subapd", C.c_float),
                ("nlayers", C.c_int32), ("layers", LayerDesc * MAX_LAYERS),
                ("ndm", C.c_int32), ("dms", DmDesc * MAX_DMS),
                ("tar_lambda", C.c_float), ("npsf", C.c_int32), ("strehl_halfwin", C.c_int32),
                ("nactu", C.c_int32), ("nslope", C.c_int32), ("gain", C.c_float),
                ("delay", C.c_float)]


class State(C.Structure):
    _fields_ = [("nenv", C.c_int32), ("ld_actu", C.c_int32), ("screens", C.c_void_p),
                ("origin", C.c_void_p), ("seeds", C.c_void_p), ("ext_count", C.c_void_p),
                ("com", C.c_void_p), ("com1", C.c_void_p), ("com2", C.c_void_p),
                ("err", C.c_void_p), ("voltage", C.c_void_p), ("slopes", C.c_void_p),
                ("dm_shape", C.c_void_p), ("bincube", C.c_void_p), ("wfs_phase", C.c_void_p),
                ("tar_phase", C.c_void_p), ("strehl", C.c_void_p), ("le_img", C.c_void_p),
                ("frame", C.c_void_p), ("work", C.c_void_p)]


# every symbol include/aomarl.h declares: (name, restype, argtypes)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_range = [_vp, C.POINTER(State), _i, _i]
SYMBOLS = [
    ("aomarl_last_error", C.c_char_p, []),
    ("aomarl_abi_version", _i, []),
    ("aomarl_set_precision", _i, [_i]),
    ("aomarl_get_precision", _i, []),
    ("aomarl_gemm_saturated", _i, [C.POINTER(C.c_uint), _vp]),
    ("aomarl_arith_families", _i, []),
    ("aomarl_arith_family_name", C.c_char_p, [_i]),
    ("aomarl_arith_launches", C.c_ulonglong, [_i]),
    ("aomarl_arith_reset", None, []),
    ("aomarl_create", _i, [C.POINTER(Desc), C.POINTER(_vp)]),
    ("aomarl_destroy", _i, [_vp]),
    ("aomarl_set_cmat", _i, [_vp, _fp]),
    ("aomarl_set_gain", _i, [_vp, _f]),
    ("aomarl_set_env_gains", _i, [_vp, _fp, _i]),
    ("aomarl_set_modal", _i, [_vp, _i, _fp, _fp, _fp, _i, _ip]),
    ("aomarl_workspace_floats", C.c_size_t, [_vp, _i]),
    ("aomarl_screen_stride", C.c_size_t, [_vp]),
    ("aomarl_dmshape_stride", C.c_size_t, [_vp]),
    ("aomarl_reset", _i, _range + [_up, _fp, _fp, _vp]),
    ("aomarl_target_image", _i, _range + [_vp, _vp]),
    ("aomarl_strehl_fit", _i, _range + [_vp]),
    ("aomarl_reset_prefetch_begin", _i, _range + [_up, _vp]),
    ("aomarl_reset_prefetch_advance", _i, [_vp, _i, _vp, C.POINTER(_i)]),
    ("aomarl_reset_prefetch_cancel", _i, [_vp]),
    ("aomarl_reset_adopt", _i, _range + [_up, _fp, _fp, _vp, _vp]),
    ("aomarl_move_atmos", _i, _range + [_fp, _fp, _vp]),
    ("aomarl_prefetch_atmos", _i, _range + [_fp, _fp, _vp]),
    ("aomarl_set_wind", _i, [_vp, _i, C.c_float, C.c_float, _i]),
    ("aomarl_set_stencil", _i, [_vp, _i, _i, _up, _i]),
    ("aomarl_set_r0", _i, [_vp, _fp, _i]),
    ("aomarl_get_layer", _i, [_vp, _i, _fp, _fp, _fp]),
    ("aomarl_extrude", _i, _range + [_i, _ip, _ip, _vp]),
    ("aomarl_get_screen", _i, _range + [_i, _vp, _vp]),
    ("aomarl_set_screen", _i, _range + [_i, _vp, _vp]),
    ("aomarl_raytrace_wfs", _i, _range + [_i, _vp]),
    ("aomarl_raytrace_target", _i, _range + [_i, _vp]),
    ("aomarl_comp_image", _i, _range + [_i, _vp]),
    ("aomarl_do_centroids", _i, _range + [_vp]),
    ("aomarl_slopes_geom", _i, _range + [_vp]),
    ("aomarl_do_control", _i, _range + [_vp]),
    ("aomarl_set_com", _i, _range + [_vp, _vp]),
    ("aomarl_rl_control", _i, _range + [_vp, _vp]),
    ("aomarl_rl_control_modes", _i, _range + [_vp, _vp, C.c_float, _vp, _vp, _vp]),
    ("aomarl_apply_control", _i, _range + [_i, _vp]),
    ("aomarl_comp_dm_shape", _i, _range + [_vp, _vp]),
    ("aomarl_get_dm_shape", _i, _range + [_i, _vp, _vp]),
    ("aomarl_set_option", _i, [_vp, C.c_char_p, _i]),
    ("aomarl_target_psf", _i, _range + [_vp]),
    ("aomarl_gemm_batched", _i, [_i, _i, _i, _i, _i, _i, _vp, _i, C.c_longlong, _vp, _i, C.c_longlong, _vp,
                                 C.c_longlong, _vp, _i, C.c_longlong, _i, _i, _vp]),
    ("aomarl_sac_layout", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("aomarl_sac_create", _i, [_vp, C.POINTER(C.c_void_p)]),
    ("aomarl_sac_destroy", _i, [_vp]),
    ("aomarl_sac_update", _i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp, _vp, _vp, C.c_uint32,
                               C.c_uint32, _i, _i, _vp, _vp]),
    ("aomarl_split_states", _i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    ("aomarl_policy_sample", _i, [_i, _i, _i, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp,
                                  _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    ("aomarl_assemble_state", _i, [_i, _i, C.POINTER(C.c_void_p), _ip, _ip, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_void_p), _vp, _vp]),
    ("aomarl_assemble_state_cols", _i, [_i, _i, C.POINTER(C.c_void_p), _ip, _ip, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), _vp, _vp, _vp]),
    ("aomarl_agent_rewards", _i, [_i, _i, _i, _vp, _i, _vp, C.c_float, _vp, _vp]),
    ("aomarl_actor_tiled_floats", C.c_longlong, [_i, _i, _i]),
    ("aomarl_actor_tile_weights", _i, [_i, _i, _i, _vp, _vp, _vp]),
    ("aomarl_actor_forward", _i, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    ("aomarl_env_step", _i, [_vp, C.POINTER(State), _vp, _vp, _f, _fp, _fp, _vp, _vp, _vp]),
    ("aomarl_policy_env_step", _i, [_vp, C.POINTER(State), _vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _f, _fp, _fp,
                                    _vp, _vp, _vp, _vp, _vp]),
    ("aomarl_env_step_shortcut", _i, [_vp, _vp]),
    ("aomarl_do_control_reduced", _i, [_vp, C.POINTER(State), _vp]),
    ("aomarl_graph_stats", _i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    ("aomarl_set_frame_pipeline", _i, [_vp, C.POINTER(State), C.POINTER(State)]),
    ("aomarl_frame_pipeline_state", _i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_ulonglong),
                                         C.POINTER(C.c_ulonglong)]),
    ("aomarl_frame_kernel_name", C.c_char_p, [_vp]),
    ("aomarl_frame_kernel_time", _i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    ("aomarl_denoiser_create", _i, [C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_void_p)]),
    ("aomarl_denoiser_apply", _i, [_vp, _vp, C.c_longlong, _vp]),
    ("aomarl_denoiser_apply_f32", _i, [_vp, _vp, C.c_longlong, _vp]),
    ("aomarl_denoiser_apply_split_f16", _i, [_vp, _vp, C.c_longlong, _vp]),
    ("aomarl_denoiser_overflow", _i, [_vp, C.POINTER(C.c_uint), _vp]),
    ("aomarl_denoiser_destroy", _i, [_vp]),
    ("aomarl_target_psf_buffer", _i, _range + [_vp]),
    ("aomarl_set_geo", _i, [_vp, _fp]),
    ("aomarl_geo_workspace_floats", C.c_size_t, [_vp, _i]),
    ("aomarl_geo_control", _i, _range + [_vp, _vp]),
    ("aomarl_frame_fused_available", _i, [_vp]),
    ("aomarl_dm_from_voltage_available", _i, [_vp]),
    ("aomarl_materialize_dm_shape", _i, _range + [_vp]),
    ("aomarl_frame_fused", _i, _range + [_i, _vp]),
    ("aomarl_comp_strehl", _i, _range + [_vp]),
    ("aomarl_reset_strehl", _i, _range + [_vp]),
    ("aomarl_volts2modes", _i, [_vp, C.POINTER(State), _i, _vp, _i, _vp, _vp]),
    ("aomarl_set_slopes2modes", _i, [_vp, _i, _fp]),
    ("aomarl_slopes2modes", _i, _range + [_vp, _vp]),
    ("aomarl_next_part_one", _i, _range + [_fp, _fp, _i, _vp]),
    ("aomarl_next_part_two", _i, _range + [_vp, _vp]),
    ("aomarl_gemm_nt", _i, [_i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, _vp]),
    ("aomarl_gemm_nt_split", _i, [_i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, _f, _f, _vp, C.c_longlong, _vp]),
    ("aomarl_gemm_nt_batched", _i, [_i, _i, _i, _i, _vp, _i, C.c_longlong, _vp, _i, C.c_longlong,
                                    _vp, C.c_longlong, _vp, _i, C.c_longlong, _i, _vp]),
]

_lib = None


class SacDesc(C.Structure):
    """aomarl_sac_desc (include/aomarl.h)"""
    _fields_ = [(n, C.c_int32) for n in ("n_agents", "batch", "in_max", "act_max", "hidden", "hidden_critic",
                                         "n_hidden",
                                         "state_dim", "action_dim")] + \
               [("state_gather", C.c_void_p), ("action_gather", C.c_void_p), ("n_act", C.c_void_p),
                ("target_entropy", C.c_void_p)] + \
               [(n, C.c_float) for n in ("gamma", "tau", "lr", "beta1", "beta2", "adam_eps", "log_sig_min",
                                         "log_sig_max", "action_scale", "action_bias")] + \
               [(n, C.c_void_p) for n in ("policy", "policy_m", "policy_v", "policy_grad", "critic",
                                          "critic_m", "critic_v", "critic_grad", "critic_target",
                                          "log_alpha", "log_alpha_m", "log_alpha_v", "log_alpha_grad",
                                          "alpha")]


class ActorDesc(C.Structure):
    """aomarl_actor_desc (include/aomarl.h)"""
    _fields_ = [(n, C.c_int32) for n in ("n_agents", "nenv", "state_dim", "in_max", "act_max", "hidden",
                                         "n_hidden", "action_dim")] + \
               [("gather", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p),
                ("Wh", C.POINTER(C.c_void_p)), ("bh", C.POINTER(C.c_void_p)),
                ("Whead", C.c_void_p), ("bhead", C.c_void_p), ("sc_agent", C.c_void_p),
                ("sc_local", C.c_void_p)] + \
               [(n, C.c_float) for n in ("log_sig_min", "log_sig_max", "scale", "bias")] + \
               [(n, C.c_void_p) for n in ("x", "h0", "h1", "head")] + [("flags", C.c_int32)] + \
               [("W1_tiled", C.c_void_p), ("Wh_tiled", C.POINTER(C.c_void_p)), ("Whead_tiled", C.c_void_p)]


ACTOR_LAYER_BY_LAYER = 1


class EnvGlue(C.Structure):
    """aomarl_env_glue (include/aomarl.h)"""
    _fields_ = [(n, C.c_int32) for n in ("nmodes", "dm_dim", "n_agents", "nhist", "ring_pos")] + \
               [(n, C.c_void_p) for n in ("sel", "mean_dm", "std_dm", "mean_res", "std_res", "lohi")] + \
               [("reward_factor", C.c_float), ("modes_ring", C.c_void_p), ("res_modes", C.c_void_p),
                ("denoiser", C.c_void_p), ("denoiser_f32", C.c_int32), ("flags", C.c_int32)]


ENV_STEP_UNFUSED = 1


class AomarlError(RuntimeError):
    pass


def build(force=False):
    """Compile libaomarl_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(HERE, "csrc")
    cmd = ["make", "-s", "-C", src_dir]
    if force:
        cmd.insert(1, "-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    """Load the HIP library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AomarlError(
                "libaomarl_hip.so is missing (%s). Build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` or `make -C ao_marl_amd/csrc`. There is no CPU "
                "fallback for the product path." % LIB_PATH)
    # PyTorch first, when it is there: it ships a HIP runtime of its own, and the library must resolve its HIP calls
    # to the runtime the tensors live in -- loaded the other way round (build() then smoke() in one process) the library
    # binds the system's runtime and its first hipMalloc finds "no ROCm-capable device" beside torch's.
    try:
        import torch  # noqa: F401
    except ImportError:                                     # a pure C-ABI user: the system's runtime is the only one
        pass
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise AomarlError("libaomarl_hip.so does not export %s" % name)
        fn.restype, fn.argtypes = res, args
    if L.aomarl_abi_version() != ABI_VERSION:
        raise AomarlError("libaomarl_hip.so ABI %d != binding %d" %
                          (L.aomarl_abi_version(), ABI_VERSION))
    _lib = L
    # AOMARL_PRECISION=split_f16 selects the fast mode for the whole process (e.g. to run the test suite
    # in it); the default is the reference's arithmetic, fp32
    mode = os.environ.get("AOMARL_PRECISION", "").strip().lower()
    if mode:
        set_precision(mode)
    return L


_PRECISIONS = {"f32": PRECISION_F32, "fp32": PRECISION_F32, "split_f16": PRECISION_SPLIT_F16,
               "split-f16": PRECISION_SPLIT_F16, "fast": PRECISION_SPLIT_F16}


def set_precision(mode):
    """Arithmetic of the whole library (aomarl_set_precision): "f32" (default, the reference's) or
    "split_f16" (fast mode: fp16 operand pairs with 22-bit mantissa, fp32 accumulation, in the frame
    kernel's DFTs, the internal GEMMs and the denoiser)."""
    if isinstance(mode, str):
        if mode.lower() not in _PRECISIONS:
            raise ValueError("unknown precision %r (f32 | split_f16)" % mode)
        mode = _PRECISIONS[mode.lower()]
    check(load().aomarl_set_precision(int(mode)))


def get_precision():
    return "split_f16" if load().aomarl_get_precision() == PRECISION_SPLIT_F16 else "f32"


def gemm_saturated(stream=None):
    """Threads of split-fp16 GEMM launches since the last call that clipped an operand at the fp16 range
    (aomarl_gemm_saturated); 0 in the default precision."""
    n = C.c_uint(0)
    check(load().aomarl_gemm_saturated(C.byref(n), stream))
    return int(n.value)


def arith_launches(reset=False):
    """{family: launches since the last reset} for the kernel families that exist in more than one
    arithmetic (aomarl_arith_*), e.g. {"gemm:f32_mfma": 12, "gemm:split_f16_mfma": 0, ...}."""
    L = load()
    out = {L.aomarl_arith_family_name(i).decode(): int(L.aomarl_arith_launches(i))
           for i in range(L.aomarl_arith_families())}
    if reset:
        L.aomarl_arith_reset()
    return out


def dtype_string(launches):
    """The `dtype` of a bench line from what was launched: "f32" when no split-fp16 family ran, otherwise
    every split-fp16 family by name."""
    split = sorted(k.split(":")[0] for k, v in launches.items() if v and k.endswith(":split_f16_mfma"))
    f32 = sorted(k.split(":")[0] for k, v in launches.items() if v and k.endswith(":f32_mfma"))
    if not split:
        return "f32"
    s = "f16x2-split operands (hi+lo, 22-bit mantissa), f32 accumulate in: " + ", ".join(split)
    if f32:
        s += "; f32 in: " + ", ".join(f32)
    return s + "; f32 vector arithmetic elsewhere"


def raw_stream(device):
    """The caller's current HIP stream on `device` as a ctypes pointer.  torch.cuda.current_stream() builds a Stream
    object (5 us per call, twice per environment step on a host-bound step); the raw handle is one C call."""
    import torch
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def check(rc):
    if rc != 0:
        raise AomarlError(load().aomarl_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(_fp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def uptr(a):
    return a.ctypes.data_as(_up)


def make_desc(s):
    """SimArrays -> (Desc, keepalive list of the NumPy arrays the Desc points to)."""
    keep = []

    def f32(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        keep.append(a)
        return fptr(a)

    def i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return iptr(a)

    def u32(a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        keep.append(a)
        return uptr(a)

    d = Desc()
    d.abi_version = ABI_VERSION
    d.n, d.pupdiam = s.n, s.pupdiam
    d.mpupil, d.spupil = f32(s.mpupil), f32(s.spupil)
    d.nvalid, d.pdiam, d.nfft, d.npix, d.nrebin, d.nxsub = (s.nvalid, s.pdiam, s.nfft, s.npix,
                                                           s.nrebin, s.nxsub)
    d.phasemap, d.halfxy, d.binmap, d.flux = i32(s.phasemap), f32(s.halfxy), i32(s.binmap), f32(s.flux)
    d.validsubsx, d.validsubsy = i32(s.validsubsx), i32(s.validsubsy)
    d.nphot, d.wfs_lambda, d.noise = float(s.nphot), s.wfs_lambda, s.noise
    d.cog_offset, d.cog_scale, d.subapd = s.cog_offset, s.cog_scale, s.subapd
    if s.nscreens > MAX_LAYERS or len(s.dms) > MAX_DMS:
        raise AomarlError("too many layers / DMs for the C ABI")
    d.nlayers = s.nscreens
    shared = {}
    for l in range(s.nscreens):
        L = d.layers[l]
        L.dim, L.nstencil = s.screen_dim[l], int(s.istx[l].size)
        # layers that share the same A/B arrays must hand over the same pointers
        key = (id(s.A[l]), id(s.B[l]))
        if key not in shared:
            shared[key] = (f32(s.A[l]), f32(s.B[l]))
        L.A, L.B = shared[key]
        L.istx, L.isty = u32(s.istx[l]), u32(s.isty[l])
        L.deltax, L.deltay = float(s.deltax[l]), float(s.deltay[l])
        L.amplitude = float(s.amplitude[l])
        L.wfs_xoff, L.wfs_yoff = s.wfs_atm_off[l]
        L.tar_xoff, L.tar_yoff = s.tar_atm_off[l]
    d.ndm = len(s.dms)
    for k, m in enumerate(s.dms):
        D = d.dms[k]
        D.dim, D.nact = m.dim, m.ntotact
        if m.type == "pzt":
            D.type, D.influsize = DM_PZT, m.influsize
            D.influ = f32(m.influ.flatten("F"))
            D.influpos, D.ninflu, D.influstart = i32(m.influpos), i32(m.ninflu), i32(m.influstart)
            D.ninflupos = int(m.influpos.size)
        else:
            D.type, D.influsize = DM_TT, m.dim
            D.influ = f32(m.influ)
        D.wfs_xoff, D.wfs_yoff = s.wfs_dm_off[k]
        D.tar_xoff, D.tar_yoff = s.tar_dm_off[k]
    d.tar_lambda, d.npsf, d.strehl_halfwin = s.tar_lambda, s.npsf, s.strehl_halfwin
    d.nactu, d.nslope, d.gain, d.delay = s.nactu, s.nslope, s.gain, s.delay
    return d, keep


def linear_batched(x, weight, bias=None, relu=False, out=None):
    """act(x @ weight^T + bias) for stacked layers on the library's batched fp32 MFMA GEMM.
    x [B, M, K], weight [B, N, K] (nn.Linear layout), bias [B, N] or None -> [B, M, N]."""
    import torch
    L = load()
    B, M, K = x.shape
    N = weight.shape[1]
    assert weight.shape == (B, N, K) and x.stride(2) == 1 and weight.stride(2) == 1
    if out is None:
        out = torch.empty(B, M, N, dtype=torch.float32, device=x.device)
    check(L.aomarl_gemm_nt_batched(
            B, M, N, K, x.data_ptr(), x.stride(1), x.stride(0), weight.data_ptr(), weight.stride(1),
            weight.stride(0), bias.data_ptr() if bias is not None else None,
            bias.stride(0) if bias is not None else 0, out.data_ptr(), out.stride(1), out.stride(0),
            1 if relu else 0, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out


def tile_weights(w):
    """[A, N, K] (nn.Linear layout, stacked) -> the tile order of the one-kernel actor
    (aomarl_actor_tile_weights)."""
    import torch
    A, N, K = w.shape
    w = w.contiguous()
    out = torch.empty(load().aomarl_actor_tiled_floats(A, N, K), dtype=torch.float32, device=w.device)
    check(load().aomarl_actor_tile_weights(A, N, K, w.data_ptr(), out.data_ptr(), _stream_of(w)))
    return out


def _stream_of(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def split_states(state, gather_i32, out=None):
    """[nenv, state_dim] -> [A, nenv, in_max] (aomarl_split_states)."""
    import torch
    nenv, sd = state.shape
    A, in_max = gather_i32.shape
    state = state.contiguous()
    if out is None:
        out = torch.empty(A, nenv, in_max, dtype=torch.float32, device=state.device)
    check(load().aomarl_split_states(nenv, sd, A, in_max, gather_i32.data_ptr(), state.data_ptr(),
                                     out.data_ptr(), _stream_of(state)))
    return out


def policy_sample(head, act_max, sc_agent_i32, sc_local_i32, log_sig_min, log_sig_max, scale, bias,
                  seed, counter, eps=None):
    """head [A, nenv, 2*act_max] -> (action, mean) [nenv, action_dim] (aomarl_policy_sample)."""
    import torch
    A, nenv, _ = head.shape
    ad = sc_agent_i32.numel()
    head = head.contiguous()
    action = torch.empty(nenv, ad, dtype=torch.float32, device=head.device)
    mean = torch.empty_like(action)
    if eps is not None:
        eps = eps.contiguous()
    check(load().aomarl_policy_sample(nenv, act_max, ad, head.data_ptr(), log_sig_min, log_sig_max,
                                      scale, bias, sc_agent_i32.data_ptr(), sc_local_i32.data_ptr(),
                                      eps.data_ptr() if eps is not None else None,
                                      int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF,
                                      action.data_ptr(), mean.data_ptr(), _stream_of(head)))
    return action, mean


def assemble_state(blocks, norms=None, out=None, sel_i32=None):
    """blocks: list of [nenv, d_k] tensors (row stride free, unit column stride); norms: list of
    (mean, std) tensors or None per block -> [nenv, sum d_k] (aomarl_assemble_state).  sel_i32: every
    block contributes its columns sel (aomarl_assemble_state_cols)."""
    import torch
    n = len(blocks)
    nenv = blocks[0].shape[0]
    for b in blocks:
        assert b.stride(1) == 1 and b.shape[0] == nenv and b.dtype == torch.float32
    src = (C.c_void_p * n)(*[b.data_ptr() for b in blocks])
    ld = (C.c_int32 * n)(*[b.stride(0) for b in blocks])
    widths = [b.shape[1] if sel_i32 is None else int(sel_i32.numel()) for b in blocks]
    dim = (C.c_int32 * n)(*widths)
    mean = std = None
    if norms is not None:
        mean = (C.c_void_p * n)(*[(m[0].data_ptr() if m is not None else None) for m in norms])
        std = (C.c_void_p * n)(*[(m[1].data_ptr() if m is not None else None) for m in norms])
    total = sum(widths)
    if out is None:
        out = torch.empty(nenv, total, dtype=torch.float32, device=blocks[0].device)
    check(load().aomarl_assemble_state_cols(nenv, n, src, ld, dim, mean, std,
                                            sel_i32.data_ptr() if sel_i32 is not None else None,
                                            out.data_ptr(), _stream_of(blocks[0])))
    return out


def agent_rewards(res_modes, lohi_i32, factor):
    """[nenv, nmodes] -> [nenv, A]: -factor * mean(res[:, lo:hi]**2) (aomarl_agent_rewards)."""
    import torch
    assert res_modes.stride(1) == 1
    nenv, nm = res_modes.shape
    A = lohi_i32.shape[0]
    out = torch.empty(nenv, A, dtype=torch.float32, device=res_modes.device)
    check(load().aomarl_agent_rewards(nenv, nm, A, res_modes.data_ptr(), res_modes.stride(0),
                                      lohi_i32.data_ptr(), float(factor), out.data_ptr(),
                                      _stream_of(res_modes)))
    return out


def gemm_batched(A, B, transA=False, transB=False, bias=None, relu=False, out=None, accumulate=False):
    """act(opA(A) @ opB(B) + bias) for stacked matrices (aomarl_gemm_batched).  A: [b, M, K] or
    [b, K, M] (transA); B: [b, N, K] or [b, K, N] (transB); bias [b, N]; returns [b, M, N]."""
    import torch
    assert A.stride(2) == 1 and B.stride(2) == 1
    b = A.shape[0]
    M, K = (A.shape[2], A.shape[1]) if transA else (A.shape[1], A.shape[2])
    N = B.shape[2] if transB else B.shape[1]
    assert (B.shape[1] if transB else B.shape[2]) == K and B.shape[0] == b
    if out is None:
        out = torch.empty(b, M, N, dtype=torch.float32, device=A.device)
    check(load().aomarl_gemm_batched(
            b, int(transA), int(transB), M, N, K, A.data_ptr(), A.stride(1), A.stride(0), B.data_ptr(),
            B.stride(1), B.stride(0), bias.data_ptr() if bias is not None else None,
            bias.stride(0) if bias is not None else 0, out.data_ptr(), out.stride(1), out.stride(0),
            1 if relu else 0, 1 if accumulate else 0, _stream_of(A)))
    return out
