"""Re-save the trained single-sub-aperture autoencoder weights the tests hold
(tests/golden/host_denoiser.pt, made by tools/gen_golden_host.py from the reference's shipped
checkpoint) as package data: ao_marl_amd/data/denoiser_subap_16x16.pt = the bare state_dict
(reference DATA -- trained weights -- not source).  bench.py and VecAoEnv users load it through
ao_marl_amd.denoiser.SubapDenoiser.load()."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tests", "golden", "host_denoiser.pt")
dst = os.path.join(ROOT, "ao_marl_amd", "data", "denoiser_subap_16x16.pt")
sd = torch.load(src, map_location="cpu", weights_only=True)["state_dict"]
torch.save({k: v.clone().contiguous() for k, v in sd.items()}, dst)
print("wrote", dst, {k: tuple(v.shape) for k, v in sd.items()})
