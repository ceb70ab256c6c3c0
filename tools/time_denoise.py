"""k_denoise on one step's worth of spot images (256 envs x 1200 sub-apertures): time and parity
against the tensor-library definition (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_marl_amd.denoiser import SubapDenoiser
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "host_denoiser.pt"), map_location="cpu", weights_only=True)
dn = SubapDenoiser(sd.get("state_dict", sd), device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
cube = torch.rand(256, 1200, 256, generator=g, device="cuda") * 30.0
small = cube[:2].clone()
ref = dn.forward(small.view(-1, 16, 16).transpose(1, 2).unsqueeze(1)).squeeze(1).transpose(1, 2).reshape(2, 1200, 256)
ref64 = None
for f32 in (False, True):
    out = dn.denoise_bincube_(small.clone(), f32=f32)
    print("%s: max |native - torch| = %.3e (max |out| %.3e)" % ("fp32 " if f32 else "f16x2", (out - ref).abs().max().item(), ref.abs().max().item()))
    work = cube.clone()
    for _ in range(2):
        dn.denoise_bincube_(work, f32=f32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dn.denoise_bincube_(work, f32=f32)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("   k_denoise: %.3f ms per 307200 images = %.1f TFLOP/s (2 x 1 712 128 MAC per image)" % (ms, 307200 * 2 * 1712128 / ms * 1e-9))
