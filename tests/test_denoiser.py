"""A17: the sub-aperture denoiser == the reference's module on the shipped weights (golden
vectors from tools/gen_golden_host.py), orientation handling, and the fused device-side flow."""
import os

import numpy as np
import pytest
import torch

from ao_marl_amd.denoiser import SubapDenoiser


@pytest.fixture(scope="module")
def blob(golden_dir):
    return torch.load(os.path.join(golden_dir, "host_denoiser.pt"), weights_only=True)


def test_forward_matches_reference_module(blob):
    d = SubapDenoiser(blob["state_dict"], device="cpu")
    assert sum(v.numel() for v in blob["state_dict"].values()) == 64449
    y = d.forward(blob["x"])
    assert torch.allclose(y, blob["y"], atol=2e-5, rtol=1e-5)


def test_bincube_round_trip_uses_the_reference_orientation(blob):
    """rlSupervisor.py:884-889: bincube (16,16,N) [x][y][i] -> moveaxis -> network -> back."""
    d = SubapDenoiser(blob["state_dict"], device="cpu")
    g = torch.Generator().manual_seed(1)
    cube = torch.rand(2, 5, 256, generator=g) * 20          # [env][subap][y*16+x]
    want = cube.clone()
    for e in range(2):
        compass = cube[e].view(5, 16, 16).permute(2, 1, 0).numpy()       # [x][y][i]
        moved = np.moveaxis(compass, -1, 0)                               # [i][x][y]
        out = d.forward(torch.from_numpy(np.ascontiguousarray(moved)).unsqueeze(1)).squeeze(1)
        want[e] = out.permute(0, 2, 1).reshape(5, 256)                    # back to [y][x]
    got = d.denoise_bincube_(cube.clone())
    assert torch.allclose(got, want, atol=1e-5)
    assert not torch.allclose(got, d.forward(cube.view(10, 1, 16, 16)).view(2, 5, 256), atol=1e-3)


@pytest.mark.gpu
def test_denoised_wfs_path_matches_oracle(blob):
    """config 5: noisy 40x40 sensor -> denoiser -> COG, device-resident, vs the oracle's noisy
    images pushed through the same network on the CPU."""
    from ao_marl_amd import geometry as G, params, system
    from ao_marl_amd.sim import HipSim
    from oracle import aoref
    from tests.test_gpu_large import QuickOracle, _push
    sysm = G.build_system(params.builtin("production_sh_40x40_8m_3layers_d0_noise"))
    s = system.from_system(sysm)
    s.cmat = np.zeros((s.nactu, s.nslope), dtype=np.float32)
    sim = HipSim(s, nenv=2, keep_bincube=True)
    oracles = [QuickOracle(s, seed=sd) for sd in (11, 12)]
    sim.reset([11, 12])
    sim.t["seeds"].copy_(torch.tensor([11, 12], dtype=torch.int32))
    _push(sim, oracles)
    dn_gpu = SubapDenoiser(blob["state_dict"], device="cuda:0")
    dn_cpu = SubapDenoiser(blob["state_dict"], device="cpu")
    sim.comp_image(noise=True, write_bincube=True, cog=False)
    dn_gpu.denoise_bincube_(sim.t["bincube"])
    sim.do_centroids()
    sl = sim.slopes.cpu().numpy()
    for e, o in enumerate(oracles):
        o.raytrace_wfs(atm=True, dms=False, reset=True)
        o.comp_image(noise=True)
        cube = torch.from_numpy(o.bincube.copy()).unsqueeze(0)
        o.bincube[:] = dn_cpu.denoise_bincube_(cube)[0].numpy()
        o.do_centroids()
        good = np.abs(sl[e] - o.slopes) < 2e-3
        assert good.mean() > 0.995, good.mean()


@pytest.mark.gpu
def test_fused_mfma_denoiser_matches_the_tensor_library_path(golden_dir):
    """aomarl_denoiser_apply (one fused MFMA kernel) vs the functional forward that the CPU tests
    pin to the reference module, on the shipped weights: fixture images + random spot-like images."""
    from ao_marl_amd.denoiser import SubapDenoiser
    g = torch.load(os.path.join(golden_dir, "host_denoiser.pt"), weights_only=True)
    dn = SubapDenoiser(g["state_dict"], device="cuda:0")
    assert dn.use_native
    gen = torch.Generator().manual_seed(3)
    cube = (torch.rand(3, 50, 256, generator=gen) * 40.0).cuda()
    cube[0, :8] = g["x"].reshape(8, 16, 16).transpose(1, 2).reshape(8, 256).cuda()   # [y][x] tiles
    want = cube.clone()
    dn.use_native = False
    dn.denoise_bincube_(want)
    dn.use_native = True
    got = cube.clone()
    dn.denoise_bincube_(got)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 2e-5 * scale
    ref = g["y"].reshape(8, 16, 16).transpose(1, 2).reshape(8, 256).cuda()
    assert (got[0, :8] - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("peak", [1.0, 40.0, 3000.0])
def test_split_fp16_and_fp32_denoiser_kernels_agree_with_fp64(golden_dir, peak):
    """Both fused kernels (aomarl_denoiser_apply: fp16 hi + lo pairs, fp32 accumulation;
    aomarl_denoiser_apply_f32: fp32 matrix instructions) against the network evaluated in fp64, from
    faint to bright spots: the split kernel stays within 3x of the fp32 kernel's own rounding."""
    from ao_marl_amd.denoiser import SubapDenoiser
    g = torch.load(os.path.join(golden_dir, "host_denoiser.pt"), weights_only=True)
    dn = SubapDenoiser(g["state_dict"], device="cuda:0")
    dn64 = SubapDenoiser(g["state_dict"], device="cpu", dtype=torch.float64)
    gen = torch.Generator().manual_seed(int(peak))
    cube = torch.rand(2, 300, 256, generator=gen) * peak
    cube[1] *= torch.rand(300, 1, generator=gen)               # a mix of faint and bright tiles
    want = dn64.denoise_bincube_(cube.double().clone())
    scale = want.abs().max().item()
    err = {}
    for f32 in (False, True):
        got = dn.denoise_bincube_(cube.cuda().clone(), f32=f32).cpu().double()
        assert torch.isfinite(got).all()
        err[f32] = (got - want).abs().max().item() / scale
    assert err[True] < 5e-6
    assert err[False] < max(3 * err[True], 5e-6), err


@pytest.mark.gpu
@pytest.mark.parametrize("f32", [True, False])
def test_fused_denoiser_walking_several_images_per_workgroup(golden_dir, f32):
    """More images than the launch has workgroups (4096): a workgroup walks 2-3 images and re-uses its two LDS
    regions with another shape in every layer, so what it must redraw between them -- the one-pixel zero borders of
    the six grids, from the host-built table of LDS offsets -- decides the SECOND image's values.  Every image against
    the same image denoised alone in a small launch (one image per workgroup) and against the functional path."""
    from ao_marl_amd.denoiser import SubapDenoiser
    g = torch.load(os.path.join(golden_dir, "host_denoiser.pt"), weights_only=True)
    dn = SubapDenoiser(g["state_dict"], device="cuda:0")
    n = 2 * 4096 + 37
    gen = torch.Generator().manual_seed(11)
    cube = torch.rand(1, n, 256, generator=gen) * 60.0
    cube[0, ::3] *= 0.05                                        # bright images next to faint ones
    cube = cube.cuda()
    got = dn.denoise_bincube_(cube.clone(), f32=f32)
    alone = torch.empty_like(cube)
    for b in range(0, n, 2048):                                # launches of at most 2048 workgroups: one image each
        alone[:, b:b + 2048] = dn.denoise_bincube_(cube[:, b:b + 2048].clone(), f32=f32)
    assert torch.equal(got, alone)
    dn.use_native = False
    want = dn.denoise_bincube_(cube.clone())
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < (2e-5 if f32 else 6e-5) * scale


@pytest.mark.gpu
def test_supervisor_with_denoiser_is_independent_of_the_atmosphere_prefetch(golden_dir):
    """rlSupervisor's autoencoder branch (image -> denoiser -> centroids) with the next frame's
    move_atmos issued on the side stream right behind the image kernel, against the plain order."""
    from ao_marl_amd.denoiser import SubapDenoiser
    from ao_marl_amd.env import VecRlSupervisor
    g = torch.load(os.path.join(golden_dir, "host_denoiser.pt"), weights_only=True)
    outs = []
    for pre in (True, False):
        dn = SubapDenoiser(g["state_dict"], device="cuda:0")
        sup = VecRlSupervisor("production_sh_40x40_8m_3layers_d0_noise", {}, 3, initial_seed=21,
                              autoencoder=dn, prefetch_atmos=pre)
        assert sup.prefetch_atmos == pre
        sup.reset()
        for _ in range(4):
            sup.next_part_one()
            sup.next_part_two(None, linear_control=True)
        outs.append((sup.get_slopes().clone(), sup.get_command().clone(), sup.get_strehl().clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_pure_delay_0_order_with_the_denoiser_in_the_sensor_path(golden_dir):
    """`modification_online` with the autoencoder (rlSupervisor.py:954-987: the denoiser sits between image formation
    and centroiding whatever the order; :964-965 only moves the target's trace behind apply_control).  The sensor's
    side -- noisy image, denoiser, centroids, integrator -- is the plain order's (another image kernel forms the
    spots: the stand-alone sensor path instead of the one-pass frame kernel, same Philox draws: a photon count may
    differ where the expected flux straddles a rounding threshold); the Strehl is the pure-delay-0 one: it already
    sees the command this step applied."""
    from ao_marl_amd.denoiser import SubapDenoiser
    from ao_marl_amd.env import VecRlSupervisor
    g = torch.load(os.path.join(golden_dir, "host_denoiser.pt"), weights_only=True)
    sups = []
    for online in (True, False):
        dn = SubapDenoiser(g["state_dict"], device="cuda:0")
        sup = VecRlSupervisor("production_sh_40x40_8m_3layers_d0_noise", dict(modification_online=online), 3,
                              initial_seed=21, autoencoder=dn, prefetch_atmos=False)
        assert sup.pure_delay_0 == online and not sup.prefetch_atmos
        sup.reset()
        sups.append(sup)
    on, off = sups
    for it in range(5):
        for sup in sups:
            sup.next_part_one()
            sup.next_part_two(None, linear_control=True)
        sl_a, sl_b = on.get_slopes(), off.get_slopes()
        ca, cb = on.get_command(), off.get_command()
        if it == 0:
            # the first frame starts from identical loop state: the two image kernels agree on the photon counts of
            # (nearly) every pixel, so the slopes and the integrator's first command do
            assert ((sl_a - sl_b).abs() < 1e-3).float().mean().item() >= 0.999
            assert (ca - cb).abs().max().item() < 5e-3 * cb.abs().max().item() + 5e-3
        else:
            # ... afterwards each loop runs on its own commands: a photon count that differs in one frame (the expected
            # flux straddling a rounding threshold) moves a centroid by 0.03 pixel and the tip-tilt rows of the command
            # matrix carry it into the next frame's mirror shape -- same loop, statistically, not frame by frame
            assert ((sl_a - sl_b).abs() < 2e-2).float().mean().item() >= 0.99, it
            assert (ca - cb).abs().max().item() < 0.1 * cb.abs().max().item(), it
    # delay 0: the command of this next_part_two is on the mirror when the target is traced behind it, so the
    # short-exposure Strehl of the pure-delay-0 order is one frame ahead of the plain order's
    sr_on, sr_off = on.get_strehl()[:, 0], off.get_strehl()[:, 0]
    assert torch.isfinite(sr_on).all() and (sr_on > 0.05).all()
    assert (sr_on - sr_off).abs().max().item() > 1e-4
    on.autoencoder.check_range()
