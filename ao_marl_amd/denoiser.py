"""Per-sub-aperture denoising autoencoder in the WFS path.

Restates the forward of the reference's `DenoisingAutoencoderCNN2DSingleSubapeture`
(src/autoencoder/autoencoder_models.py:130-197) functionally on a state dict in the reference's
checkpoint layout (keys encoder1..3 / decoder1..3 .weight/.bias), and the data flow of
`RlSupervisor.autoencoder_denoising` (rlSupervisor.py:876-891) without its host round trip:
the bincube stays on the device, the denoised spots are written back in place and the centroider
runs on them.

Orientation: the reference feeds the network `np.moveaxis(np.array(d_bincube), -1, 0)`, i.e.
[subap][x][y] (COMPASS arrays are first-index-fastest), the transpose of this repo's [y][x] tiles;
the trained weights expect that, so tiles are transposed on the way in and out.

On the GPU the whole network is one fused MFMA kernel of the library (aomarl_denoiser_apply,
ao_marl_amd/csrc/aomarl_denoise.hip); the tensor-library path (`forward`, MIOpen convolutions) is
kept as the definition it is tested against and for CPU runs.  1 712 128 MAC per 16x16 image.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

KEYS = ("encoder1", "encoder2", "encoder3", "decoder1", "decoder2", "decoder3")

# The default kernel carries activations as fp16 pairs: |v| must stay below 65504.  Activations of
# this architecture's trained weights reach at most ~2x the brightest input pixel (measured on the
# shipped network in fp64: faint, bright, uniform, single-pixel and random images); inputs whose
# bound is above FP16_INPUT_LIMIT go to the all-fp32 kernel.  What slips through is counted by the
# kernel itself and raised by `check_range`.
FP16_INPUT_LIMIT = 65504.0 / 4.0


def shipped_weights_path():
    """The reference's trained single-sub-aperture autoencoder (its state_dict, re-saved as plain
    tensors by tools/import_denoiser_weights.py)."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "denoiser_subap_16x16.pt")


class SubapDenoiser(object):
    def __init__(self, state_dict, device="cuda:0", dtype=torch.float32, chunk=65536):
        self.device, self.dtype, self.chunk = torch.device(device), dtype, int(chunk)
        self.w = {}
        for k in KEYS:
            for p in ("weight", "bias"):
                t = state_dict["%s.%s" % (k, p)].detach().to(self.device, dtype)
                if p == "weight":
                    t = t.contiguous(memory_format=torch.channels_last)
                self.w["%s.%s" % (k, p)] = t
        shapes = {k: tuple(self.w[k + ".weight"].shape) for k in KEYS}
        want = {"encoder1": (16, 1, 3, 3), "encoder2": (32, 16, 3, 3), "encoder3": (64, 32, 3, 3),
                "decoder1": (64, 32, 4, 4), "decoder2": (32, 16, 4, 4), "decoder3": (16, 1, 3, 3)}
        if shapes != want:
            raise ValueError("unexpected autoencoder layout %r" % (shapes,))
        self._handle = None
        self.input_bound = None          # largest pixel value the caller can produce (set_input_bound)
        self._used_fp16 = False
        self.use_native = self.device.type == "cuda" and dtype == torch.float32
        self._host = {k: state_dict[k].detach().to("cpu", torch.float32).contiguous().numpy()
                      for k in ["%s.%s" % (a, b) for a in KEYS for b in ("weight", "bias")]}

    def _native(self):
        if self._handle is None:
            from . import libaomarl as la
            fp = C.POINTER(C.c_float)
            wt = (fp * 6)(*[self._host[k + ".weight"].ctypes.data_as(fp) for k in KEYS])
            bs = (fp * 6)(*[self._host[k + ".bias"].ctypes.data_as(fp) for k in KEYS])
            h = C.c_void_p()
            la.check(la.load().aomarl_denoiser_create(wt, bs, C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                from . import libaomarl as la
                la.load().aomarl_denoiser_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    @classmethod
    def load(cls, path=None, **kw):
        sd = torch.load(path or shipped_weights_path(), map_location="cpu", weights_only=True)
        return cls(sd.get("state_dict", sd), **kw)

    def set_input_bound(self, bound):
        """Largest pixel value the images can hold (e.g. photons of the brightest sub-aperture + noise
        margin).  Decides between the split-fp16 kernel and the all-fp32 one."""
        self.input_bound = float(bound)

    def wants_f32(self, bincube=None):
        """True in the library's default precision (f32); in the fast mode (libaomarl.set_precision
        ("split_f16")) only when the images may leave the range the split-fp16 kernel is exact in.
        Without a declared bound the cube itself is measured (one device reduction + sync)."""
        from . import libaomarl as la
        if la.get_precision() == "f32":         # the library's default: the reference's arithmetic
            return True
        if self.input_bound is None:
            if bincube is None:
                return False
            return float(bincube.abs().max()) >= FP16_INPUT_LIMIT
        return self.input_bound >= FP16_INPUT_LIMIT

    def check_range(self):
        """Raise if any split-fp16 launch since the last check saturated (aomarl_denoiser_overflow).
        Synchronises; the supervisor calls it at episode boundaries."""
        if not (self._handle and self._used_fp16):
            return
        from . import libaomarl as la
        n = C.c_uint(0)
        la.check(la.load().aomarl_denoiser_overflow(
                self._handle, C.byref(n), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self._used_fp16 = False
        if n.value:
            raise FloatingPointError(
                    "WFS-image denoiser: activations left the fp16 range in %d kernel threads since the "
                    "last check; those frames are wrong.  Declare the input range (set_input_bound) or "
                    "call denoise_bincube_(..., f32=True)" % n.value)

    @torch.no_grad()
    def forward(self, x):
        """x: [N, 1, 16, 16] (reference orientation) -> same shape."""
        w = self.w
        x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
        x = F.relu(F.conv2d(x, w["encoder1.weight"], w["encoder1.bias"], padding=1))
        x = F.max_pool2d(x, 2)
        x = F.relu(F.conv2d(x, w["encoder2.weight"], w["encoder2.bias"], padding=1))
        x = F.max_pool2d(x, 2)
        x = F.relu(F.conv2d(x, w["encoder3.weight"], w["encoder3.bias"], padding=1))
        x = F.relu(F.conv_transpose2d(x, w["decoder1.weight"], w["decoder1.bias"], stride=2,
                                      padding=1))
        x = F.relu(F.conv_transpose2d(x, w["decoder2.weight"], w["decoder2.bias"], stride=2,
                                      padding=1))
        x = F.conv_transpose2d(x, w["decoder3.weight"], w["decoder3.bias"], stride=1, padding=1)
        return x.float()

    @torch.no_grad()
    def denoise_bincube_(self, bincube, f32=None):
        """In place on a [nenv, nvalid, 256] bincube of [y][x] tiles.  f32 = True: every product on
        fp32 matrix instructions (aomarl_denoiser_apply_f32); False: fp16 pairs; None (default):
        what the library's precision mode says (wants_f32: fp32 unless the fast mode is on and the
        declared / measured input range fits fp16 pairs)."""
        n, nv, np2 = bincube.shape
        if self.use_native and bincube.is_contiguous() and bincube.dtype == torch.float32:
            from . import libaomarl as la
            if f32 is None:
                f32 = self.wants_f32(bincube)
            self._used_fp16 = self._used_fp16 or not f32
            fn = la.load().aomarl_denoiser_apply_f32 if f32 else la.load().aomarl_denoiser_apply_split_f16
            la.check(fn(
                    self._native(), bincube.data_ptr(), n * nv,
                    C.c_void_p(torch.cuda.current_stream(bincube.device).cuda_stream)))
            return bincube
        flat = bincube.view(n * nv, 16, 16)
        for i0 in range(0, n * nv, self.chunk):
            t = flat[i0:i0 + self.chunk]
            y = self.forward(t.transpose(1, 2).unsqueeze(1))
            t.copy_(y.squeeze(1).transpose(1, 2))
        return bincube
