"""GPU parity tests for the FFT path -- through the C ABI (include/sdsp_hip.h) on a real MI355X.

Checker: the CPU oracle (pinned bit-for-bit to the reference) and the committed golden vectors.
Tolerances: f64 kernels are held to the reference's own bound 4*N*eps_double
(testFFT.cpp:37), relative to max|X|; f32 kernels to max_k|X_gpu - X_ref| / max_k|X_ref| <= 1e-6
per transform, X_ref = the reference algorithm in double on the fp32-rounded input.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_max_err

pytestmark = pytest.mark.gpu

EPS64 = np.finfo(np.float64).eps
TOL32 = 1e-6


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def sd():
    import simpledsp_amd
    simpledsp_amd.load(build_if_missing=True)
    return simpledsp_amd


def _run(sd, torch, x, radix, T, precision):
    dt = torch.complex128 if precision == sd.F64 else torch.complex64
    d = torch.from_numpy(np.ascontiguousarray(x)).to(dt).cuda()
    plan = sd.FftPlan(x.shape[-1], radix, T, precision, max_batch=max(1, d.numel() // x.shape[-1]))
    plan.exec(d)
    torch.cuda.synchronize()
    return d.cpu().numpy()


def _tol64(n):
    return 4 * n * EPS64


def test_golden_vectors_f64_and_f32(sd, torch_cuda, fft_golden):
    checked = 0
    for key in fft_golden.files:
        if "__r" not in key:
            continue
        tag, op = key.split("__")
        radix, rev = int(op[1]), op.endswith("rev")
        T = sd.reverse_fft if rev else sd.forward_fft
        x, want = fft_golden[f"{tag}__in"], fft_golden[key]
        n = x.shape[-1]
        got64 = _run(sd, torch_cuda, x, radix, T, sd.F64)
        assert rel_max_err(got64, want) < _tol64(n), (key, rel_max_err(got64, want))
        got32 = _run(sd, torch_cuda, x.astype(np.complex64), radix, T, sd.F32)
        # inputs of the rand*/bench/cos fixtures: rand* are fp32-representable; for the others the
        # reference result on the fp32-rounded input is recomputed by the oracle in the next test
        if tag.startswith("rand"):
            assert rel_max_err(got32, want) < TOL32, (key, rel_max_err(got32, want))
        checked += 1
    assert checked == 56


@pytest.mark.parametrize("radix", [2, 4])
def test_reference_known_answer_tests(sd, torch_cuda, radix):
    # testFFT.cpp:17-67 / :127-177 through the f64 kernels at the reference's own tolerance
    N, n = 64, 7
    i = np.arange(N)
    s = np.cos(n * 2 * np.pi * i / N).astype(np.complex128)
    S = np.zeros(N, np.complex128)
    S[n] = S[N - n] = N / 2
    tol = 4 * N * EPS64
    assert np.abs(_run(sd, torch_cuda, s, radix, sd.forward_fft, sd.F64) - S).max() < tol
    assert np.abs(_run(sd, torch_cuda, S, radix, sd.reverse_fft, sd.F64) - s).max() < tol
    s2 = np.cos(n * 2 * np.pi * i / N + np.pi / 2).astype(np.complex128)
    S2 = np.zeros(N, np.complex128)
    S2[n], S2[N - n] = 1j * N / 2, -1j * N / 2
    assert np.abs(_run(sd, torch_cuda, s2, radix, sd.forward_fft, sd.F64) - S2).max() < tol
    # f32 kernels: the same bound re-expressed for fp32 (4*64*eps32 = 3.05e-5 against a peak of 32)
    tol32 = 4 * N * np.finfo(np.float32).eps
    assert np.abs(_run(sd, torch_cuda, s.astype(np.complex64), radix, sd.forward_fft, sd.F32) - S).max() < tol32
    assert np.abs(_run(sd, torch_cuda, S.astype(np.complex64), radix, sd.reverse_fft, sd.F32) - s).max() < tol32


@pytest.mark.parametrize("radix", [2, 4])
def test_linearity(sd, torch_cuda, radix):
    # testFFT.cpp:70-125 / :180-235
    N = 256
    i = np.arange(N)
    x1 = np.sin(2 * np.pi * 1000.0 / 8000.0 * i).astype(np.complex128)
    x2 = np.sin(2 * np.pi * 500.0 / 8000.0 * i).astype(np.complex128)
    f = lambda v: _run(sd, torch_cuda, v, radix, sd.forward_fft, sd.F64)
    assert np.abs(f(1.5 * x1 + 2.5 * x2) - (1.5 * f(x1) + 2.5 * f(x2))).max() < 4 * N * EPS64


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("radix", [2, 4])
def test_every_size_against_oracle(sd, torch_cuda, oracle, radix, precision):
    prec = sd.F32 if precision == "f32" else sd.F64
    top = 14 if prec == sd.F32 else 13
    rng = np.random.default_rng(100 + radix)
    for k in range(1, top + 1):
        n = 1 << k
        if radix == 4 and k % 2:
            continue
        batch = 5 if n <= 1024 else 2
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
            want = oracle.fft(x.astype(np.complex128), radix, rev)
            got = _run(sd, torch_cuda, x.astype(np.complex128) if prec == sd.F64 else x, radix, T, prec)
            err = rel_max_err(got, want)
            assert err < (_tol64(n) if prec == sd.F64 else TOL32), (n, radix, rev, err)


@pytest.mark.parametrize("n,radix,batch", [(16, 2, 1), (16, 4, 300), (32, 2, 129), (64, 4, 1000), (128, 2, 33), (256, 4, 17),
                                           (512, 2, 9), (512, 2, 1031), (256, 2, 4098), (256, 2, 3), (256, 4, 4099), (256, 4, 2), (2048, 2, 130), (1024, 2, 7), (1024, 4, 5), (1024, 4, 1001), (1024, 2, 64), (2048, 2, 3), (4096, 2, 5),
                                           (8192, 2, 3), (16384, 2, 2), (16384, 4, 3)])
def test_register_pass_family_ragged_and_variants(sd, torch_cuda, oracle, n, radix, batch):
    # f32 sizes 16..2048 run through csrc/fft_reg.hip (2048/n transforms per workgroup: ragged tails); the one-wave kernels of
    # csrc/fft_wave.hip (N = 256 / 1024 / 2048) are their sizes' variant 2 since the family's tiles shrank to 2048 points (round 3)
    torch = torch_cuda
    rng = np.random.default_rng(n * 7 + batch)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        want = oracle.fft(x.astype(np.complex128), radix, rev)
        plan = sd.FftPlan(n, radix, T, sd.F32, max_batch=batch)
        big = n >= 8192  # registers-resident single-pass kernel (csrc/fft_big.hip) is variant 0 there: radix-2 stages, or the
        # seven radix-4 stages of (16384, 4) -- whose variant 1 is csrc/fft_mix.hip (leading radix-4 stage + the N = 4096 machinery)
        mix = (n, radix) == (16384, 4)
        wave2 = (radix == 2 and n in (256, 2048)) or (radix == 4 and n == 256)  # csrc/fft_wave.hip: 1024 points (or one transform of 2048) per wave
        assert plan.info.kernel.decode() == ("sdsp_fft4096_r2_f32" if (n, radix) == (4096, 2) else
                                             "sdsp_fft_big_kernel" if big else "sdsp_fft_reg_kernel")
        outs = []
        # register-pass family streaming / default policy (mix sizes: fft_big), coverage kernel, and the size's other kernel: the tuned one
        # (4096 radix 2, N >= 8192) or the one-wave kernel (N = 256 / 1024 / 2048: variant 2; N = 1024's runs the family's arithmetic: same bits)
        for variant in ((2, 1, 99, 0) if (n, radix) == (4096, 2) or big else (0, 1, 99, 2)):
            plan.set_variant(variant)
            if variant == 2 and not big and (n, radix) != (4096, 2):
                assert plan.info.kernel.decode() == ("sdsp_fft1024_wave" if n == 1024 else "sdsp_fft_wave_f32" if wave2 else "sdsp_fft_reg_kernel")
            d = torch.from_numpy(x).cuda()
            guard = torch.full((64,), 7.0 + 3.0j, dtype=torch.complex64, device="cuda")  # overrun detector
            plan.exec(d)
            torch.cuda.synchronize()
            outs.append(d.cpu().numpy())
            assert rel_max_err(outs[-1], want) < TOL32, (n, radix, rev, variant, rel_max_err(outs[-1], want))
            assert bool((guard == 7.0 + 3.0j).all())
        if not mix:
            assert np.array_equal(outs[0], outs[1])  # the same kernel with the other cache policy
        if wave2:  # passes of log2(N / 64) stages instead of four: another split of the same twiddles, not the same roundings
            assert rel_max_err(outs[0], outs[3]) < 1e-6
        elif n == 1024:
            assert np.array_equal(outs[0], outs[3])


@pytest.mark.parametrize("n,radix,ref_radix", [(8192, 0, 2), (16384, 4, 4)])
@pytest.mark.parametrize("batch", [1, 5, 300])
def test_mixed_radix_kernels(sd, torch_cuda, oracle, n, radix, ref_radix, batch):
    """csrc/fft_mix.hip: N = 2 * 4^6 through the radix-4 kernel with one radix-2 stage (plans of radix AUTO);
    N = 4^7 with genuine radix-4 stages (sdsp::fft_radix4<T,16384>, fft.h:301-360) in csrc/fft_big.hip's radix-4 form
    (default) and in csrc/fft_mix.hip (variant 1).  Against the oracle's algorithm of the
    stage type the plan reports, against the radix-2-stage kernel (variant 1) on every transform, both directions, and
    the round trip."""
    torch = torch_cuda
    rng = np.random.default_rng(n + batch + radix)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    pick = sorted({0, batch - 1, batch // 2})
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        plan = sd.FftPlan(n, radix, T, sd.F32, max_batch=batch)
        # the default is csrc/fft_big.hip (radix-4 form at (16384, 4), radix-2 stages at (8192, AUTO)) and csrc/fft_mix.hip variant 1
        first, second = "sdsp_fft_big_kernel", "sdsp_fft_mix_f32"
        assert plan.info.kernel.decode() == first and plan.info.hbm_passes == 1 and plan.info.radix == ref_radix
        d = torch.from_numpy(x).cuda()
        guard = torch.full((64,), 7.0 + 3.0j, dtype=torch.complex64, device="cuda")
        plan.exec(d)
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        assert bool((guard == 7.0 + 3.0j).all())
        want = oracle.fft(x[pick].astype(np.complex128), ref_radix, rev)
        assert rel_max_err(got[pick], want) < TOL32, rel_max_err(got[pick], want)
        plan.set_variant(1)
        assert plan.info.kernel.decode() == second
        d2 = torch.from_numpy(x).cuda()
        plan.exec(d2)
        torch.cuda.synchronize()
        assert rel_max_err(got, d2.cpu().numpy().astype(np.complex128)) < TOL32
    fwd = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=batch)
    inv = sd.FftPlan(n, radix, sd.reverse_fft, sd.F32, max_batch=batch)
    d = torch.from_numpy(x).cuda()
    fwd.exec(d)
    inv.exec(d)
    torch.cuda.synchronize()
    assert rel_max_err(d.cpu().numpy(), x) < TOL32
    # an explicit radix 2 is honoured: radix-2 butterflies only; AUTO at 16384 picks that form too (2 points faster than the radix-4 one)
    assert sd.FftPlan(n, 2, sd.forward_fft, sd.F32).info.kernel.decode() == "sdsp_fft_big_kernel"
    auto = sd.FftPlan(16384, 0, sd.forward_fft, sd.F32).info
    assert auto.kernel.decode() == "sdsp_fft_big_kernel" and auto.radix == 2


@pytest.mark.parametrize("batch", [1, 3, 255, 2049])
def test_fft4096_radix2_tuned_kernel(sd, torch_cuda, oracle, batch):
    torch = torch_cuda
    rng = np.random.default_rng(40 + batch)
    x = (rng.standard_normal((batch, 4096)) + 1j * rng.standard_normal((batch, 4096))).astype(np.complex64)
    pick = rng.choice(batch, size=min(batch, 6), replace=False)
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        plan = sd.FftPlan(4096, 2, T, sd.F32, max_batch=batch)
        assert plan.info.kernel.decode() == "sdsp_fft4096_r2_f32"
        d = torch.from_numpy(x).cuda()
        plan.exec(d)
        torch.cuda.synchronize()
        want = oracle.fft(x[pick].astype(np.complex128), 2, rev)
        assert rel_max_err(d.cpu().numpy()[pick], want) < TOL32


def test_radix_auto_picks_the_stage_type(sd, torch_cuda, oracle):
    # SDSP_HIP_RADIX_AUTO (0): radix 4 where n is a power of 4, radix 2 otherwise (SURVEY 8f-4 "mixed" entry)
    rng = np.random.default_rng(5)
    for n, want_radix in ((1024, 4), (2048, 2), (4096, 4), (8192, 2)):
        plan = sd.FftPlan(n, 0, sd.forward_fft, sd.F32, max_batch=3)
        assert plan.info.radix == want_radix
        x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex64)
        d = torch_cuda.from_numpy(x).cuda()
        plan.exec(d)
        torch_cuda.cuda.synchronize()
        assert rel_max_err(d.cpu().numpy(), oracle.fft(x.astype(np.complex128), want_radix)) < TOL32


def test_n1_is_identity(sd, torch_cuda):
    x = np.array([[1 + 2j], [3 - 1j]], np.complex64)
    assert np.array_equal(_run(sd, torch_cuda, x, 2, sd.forward_fft, sd.F32), x)


@pytest.mark.parametrize("batch", [1, 3, 255, 1000, 2049])
def test_fft4096_ragged_batches_all_variants(sd, torch_cuda, oracle, batch):
    torch = torch_cuda
    rng = np.random.default_rng(batch)
    x = (rng.standard_normal((batch, 4096)) + 1j * rng.standard_normal((batch, 4096))).astype(np.complex64)
    pick = rng.choice(batch, size=min(batch, 6), replace=False)
    want = oracle.fft(x[pick].astype(np.complex128), 4)
    want_rev = oracle.fft(x[pick].astype(np.complex128), 4, True)
    first = None
    for T, ref in ((sd.forward_fft, want), (sd.reverse_fft, want_rev)):
        plan = sd.FftPlan(4096, 4, T, sd.F32, max_batch=batch)
        assert plan.info.kernel.decode().startswith("sdsp_fft4096_r4_f32")
        for variant in range(3):  # the default and its two documented alternates (scheduling only)
            plan.set_variant(variant)
            d = torch.from_numpy(x).cuda()
            plan.exec(d)
            torch.cuda.synchronize()
            got = d.cpu().numpy()
            assert rel_max_err(got[pick], ref) < TOL32, (variant, rel_max_err(got[pick], ref))
            if T is sd.forward_fft:
                if first is None:
                    first = got
                else:  # variants only differ in scheduling: same arithmetic, same bits
                    assert np.array_equal(got, first), variant
        # the coverage kernel computes the same transform
        plan.set_variant(99)
        d = torch.from_numpy(x).cuda()
        plan.exec(d)
        torch.cuda.synchronize()
        assert rel_max_err(d.cpu().numpy()[pick], ref) < TOL32


@pytest.mark.parametrize("n,radix,batch", [(1 << 16, 2, 5), (1 << 16, 4, 3), (1 << 17, 2, 3), (1 << 18, 4, 2), (1 << 19, 2, 2),
                                           (1 << 21, 2, 3), (1 << 22, 4, 1), (1 << 22, 2, 3), (1 << 23, 2, 1)])
def test_three_pass_mid_sizes(sd, torch_cuda, oracle, n, radix, batch):
    """N = 2^16 .. 2^19, f32: two passes over HBM (csrc/fft_2pass.hip, N = N1 x N2 with N1, N2 in {256, 512, 1024}); round 3:
    N = 2^21 = 1024 x 2048 and N = 2^22 = 2048 x 2048 too (a factor of 2048 = 32 threads x 64 points; were four nested passes);
    variant 1: the three streaming passes of csrc/fft_mid.hip (16-point column step, 16 x batch rows on the tuned
    single-pass kernels, untwist) -- which N = 2^23 still runs, nested; last variant: the general four-step
    through the coverage kernel.  All against the oracle, with a plan whose workspace is smaller than the batch (slices)."""
    torch = torch_cuda
    rng = np.random.default_rng(n + radix + batch)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    two_pass = n != (1 << 23)
    mid_passes = 3 if n < (1 << 20) else 4  # column step + the rows' passes (one; two for rows of 2^17 .. 2^19) + untwist
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        want = oracle.fft(x.astype(np.complex128), radix, rev) if n <= (1 << 20) else \
            (np.fft.ifft(x.astype(np.complex128), axis=-1) if rev else np.fft.fft(x.astype(np.complex128), axis=-1))
        plan = sd.FftPlan(n, radix, T, sd.F32, max_batch=2)  # batch > max_batch: the exec runs in slices
        expect = [("sdsp_fft2p_cols", 2), ("sdsp_fft_col16_kernel", mid_passes), ("sdsp_fft_tile_kernel", 2)] if two_pass else \
                 [("sdsp_fft_col16_kernel", 4), ("sdsp_fft_tile_kernel", 2)]  # nested: column step + two-pass rows + untwist
        for variant, (kernel, passes) in enumerate(expect):
            plan.set_variant(variant)
            assert plan.info.kernel.decode().startswith(kernel), (variant, plan.info.kernel)
            assert plan.info.hbm_passes == passes
            d = torch.from_numpy(x).cuda()
            guard = torch.full((64,), 7.0 + 3.0j, dtype=torch.complex64, device="cuda")
            plan.exec(d)
            torch.cuda.synchronize()
            got = d.cpu().numpy()
            assert bool((guard == 7.0 + 3.0j).all())
            assert rel_max_err(got, want) < TOL32, (n, radix, rev, variant, rel_max_err(got, want))


@pytest.mark.parametrize("n,radix,batch", [(1 << 14, 2, 5), (1 << 14, 4, 3), (1 << 15, 2, 3), (1 << 16, 4, 2), (1 << 16, 2, 3), (1 << 17, 2, 2),
                                           (1 << 18, 2, 2), (1 << 19, 2, 1), (1 << 20, 2, 2)])
def test_three_pass_f64(sd, torch_cuda, oracle, n, radix, batch):
    """The three-pass schedule in double (N = 2^14 .. 2^21): rows on the f64 register-pass family, nested from 2^18."""
    torch = torch_cuda
    rng = np.random.default_rng(n * 3 + radix)
    x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        want = oracle.fft(x, radix, rev)
        plan = sd.FftPlan(n, radix, T, sd.F64, max_batch=2)
        fast = None  # round 3: sizes whose default is no longer the three-pass schedule (it became their variant 1)
        if n == 1 << 14:
            fast = ("sdsp_fft_big_f64_kernel", 1)  # the registers-resident kernel in double (radix-2 or radix-4 stages)
        elif n >= 1 << 15:
            fast = ("sdsp_fft2p_cols+sdsp_fft2p_rows", 2)  # the two-pass kernels in double (2^15 = 128 x 256 since the second part of round 3)
        if fast:
            assert plan.info.kernel.decode() == fast[0] and plan.info.hbm_passes == fast[1]
            d0 = torch.from_numpy(x).cuda()
            plan.exec(d0)
            torch.cuda.synchronize()
            assert rel_max_err(d0.cpu().numpy(), want) < _tol64(n), (n, radix, rev)
            plan.set_variant(1)
        assert plan.info.kernel.decode().startswith("sdsp_fft_col16_kernel")
        # column step + the rows' own passes + untwist.  Rows of 1024 .. 16384 (radix 2) are one pass (N = 16384: the
        # registers-resident kernel in double, were three), rows of 2^15 three, rows of 2^16 two (the two-pass kernels in double)
        rows_passes = {1 << 14: 1, 1 << 15: 1, 1 << 16: 1, 1 << 17: 1, 1 << 18: 1, 1 << 19: 2, 1 << 20: 2}[n]
        assert plan.info.hbm_passes == 2 + rows_passes
        d = torch.from_numpy(x).cuda()
        plan.exec(d)
        torch.cuda.synchronize()
        assert rel_max_err(d.cpu().numpy(), want) < _tol64(n), (n, radix, rev)
        plan.set_variant(2 if fast else 1)  # the general four-step through the coverage kernel
        d2 = torch.from_numpy(x).cuda()
        plan.exec(d2)
        torch.cuda.synchronize()
        assert rel_max_err(d2.cpu().numpy(), want) < _tol64(n)


@pytest.mark.parametrize("n,radix,batch", [(1 << 15, 2, 3), (1 << 16, 4, 2), (1 << 18, 2, 2), (1 << 20, 2, 2), (1 << 20, 4, 1)])
def test_four_step_large_transforms(sd, torch_cuda, oracle, n, radix, batch):
    # sizes beyond LDS: two HBM passes.  The reference itself cannot be compiled for these
    # (SURVEY 8c); the checker is the oracle, itself cross-checked against numpy at 2^14..2^17.
    rng = np.random.default_rng(n + radix)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    want = oracle.fft(x.astype(np.complex128), radix)
    got = _run(sd, torch_cuda, x, radix, sd.forward_fft, sd.F32)
    assert rel_max_err(got, want) < TOL32, rel_max_err(got, want)
    assert rel_max_err(want, np.fft.fft(x.astype(np.complex128))) < 1e-12
    back = _run(sd, torch_cuda, got, radix, sd.reverse_fft, sd.F32)
    assert rel_max_err(back, x) < TOL32
    if n <= (1 << 16):
        got64 = _run(sd, torch_cuda, x.astype(np.complex128), radix, sd.forward_fft, sd.F64)
        assert rel_max_err(got64, want) < _tol64(n)


@pytest.mark.parametrize("n", [8192, 16384, 32768])
@pytest.mark.parametrize("batch", [1, 5, 300])
def test_large_single_pass_kernel(sd, torch_cuda, oracle, n, batch):
    """csrc/fft_big.hip (N = 8192 / 16384 / 32768 radix 2, f32): against the oracle on a few transforms,
    against the previous kernels (register family / four-step, variant 1) on all of them, forward and
    reverse, plus the round trip."""
    torch = torch_cuda
    rng = np.random.default_rng(n + batch)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        plan = sd.FftPlan(n, 2, T, sd.F32, max_batch=batch)
        assert plan.info.kernel.decode() == "sdsp_fft_big_kernel" and plan.info.hbm_passes == 1
        d = torch.from_numpy(x).cuda()
        guard = torch.full((64,), 7.0 + 3.0j, dtype=torch.complex64, device="cuda")
        plan.exec(d)
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        assert bool((guard == 7.0 + 3.0j).all())
        pick = sorted({0, batch - 1, batch // 2})
        want = oracle.fft(x[pick].astype(np.complex128), 2, rev)
        assert rel_max_err(got[pick], want) < TOL32, rel_max_err(got[pick], want)
        plan.set_variant(1)
        d2 = torch.from_numpy(x).cuda()
        plan.exec(d2)
        torch.cuda.synchronize()
        assert rel_max_err(got, d2.cpu().numpy().astype(np.complex128)) < TOL32
    fwd = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch)
    inv = sd.FftPlan(n, 2, sd.reverse_fft, sd.F32, max_batch=batch)
    d = torch.from_numpy(x).cuda()
    fwd.exec(d)
    inv.exec(d)
    torch.cuda.synchronize()
    assert rel_max_err(d.cpu().numpy(), x) < TOL32


def test_fft1m_with_a_one_transform_workspace(sd, torch_cuda, oracle):
    """A plan created with max_batch = 1 owns a single intermediate: fewer than the 8 x 3 the persistent kernel needs, so
    both variants of such a plan run the two-launch schedule, in chunks of one transform.  (Round 1's regression, found by
    tests/fuzz_crosscheck.py seed 31: a variant wrote past such a workspace.)"""
    rng = np.random.default_rng(31)
    n = 1 << 20
    x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
    want = np.fft.fft(x.astype(np.complex128), axis=-1)
    for radix in (2, 4):
        for variant in (0, 1):
            plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=1)
            plan.set_variant(variant)
            assert plan.info.kernel.decode() == "sdsp_fft1m_cols+sdsp_fft1m_rows"
            assert plan.launches(2) == 4  # two passes per chunk of one
            d = torch_cuda.from_numpy(x).cuda()
            guard = torch_cuda.full((1 << 16,), 7.0 + 3.0j, dtype=torch_cuda.complex64, device="cuda")
            plan.exec(d)
            torch_cuda.cuda.synchronize()
            assert rel_max_err(d.cpu().numpy(), want) < TOL32, (radix, variant)
            assert bool((guard == 7.0 + 3.0j).all())


def test_fft1m_schedules_agree(sd, torch_cuda, oracle):
    # N = 2^20 radix-2: the persistent launch (variant 0 of a plan whose workspace holds its 8 x 3 intermediates) and the
    # two launches per chunk (variant 1; also what a plan with a small workspace runs) do the same arithmetic -> the same
    # bits; batch 37 is ragged against the chunk size, leaves the eight ticket queues uneven and is longer than the ring
    torch = torch_cuda
    n, batch = 1 << 20, 37
    g = torch.Generator(device="cuda").manual_seed(20)
    x = torch.view_as_complex(torch.randn((batch, n, 2), generator=g, device="cuda"))
    want = oracle.fft(x[[0, 17, 36]].cpu().numpy().astype(np.complex128), 2)
    plan = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch)
    first = None
    assert plan.info.kernel.decode() == "sdsp_fft1m_fused"
    small = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=3)
    assert small.info.kernel.decode() == "sdsp_fft1m_cols+sdsp_fft1m_rows"
    for p, variant in ((plan, 0), (plan, 1), (small, 0), (plan, 0)):
        p.set_variant(variant)
        for rep in range(3):  # the intermediate ring is re-used across calls: warm caches must not leak stale lines
            y = x.clone()
            p.exec(y)
            p.exec(y.clone())  # a second call right behind it must not disturb the first one's result
            p.status()  # synchronises; raises if a bounded wait of the in-kernel hand-off gave up
            assert rel_max_err(y[[0, 17, 36]].cpu().numpy(), want) < TOL32, variant
            if first is None:
                first = y
            else:
                assert torch.equal(first, y), (variant, rep)
    # variant 2: the same schedule through fft_2pass.hip's generic persistent kernel (other tile functions: same tolerance, not the same bits)
    plan.set_variant(2)
    assert plan.info.kernel.decode() == "sdsp_fft2p_fused" and plan.launches(batch) == 1
    y = x.clone()
    plan.exec(y)
    plan.status()
    assert rel_max_err(y[[0, 17, 36]].cpu().numpy(), want) < TOL32
    assert rel_max_err(y.cpu().numpy(), first.cpu().numpy().astype(np.complex128)) < TOL32


@pytest.mark.parametrize("n,radix,precision,batch", [(4096, 4, "f32", 1), (4096, 4, "f32", 67), (4096, 2, "f32", 5),
                                                     (256, 4, "f32", 33), (1024, 2, "f64", 4), (1 << 15, 2, "f32", 2),
                                                     (16, 2, "f32", 300), (64, 4, "f32", 70), (1024, 2, "f32", 9), (1024, 4, "f32", 1030),
                                                     (256, 2, "f32", 1027), (256, 4, "f32", 1026), (512, 2, "f32", 77), (2048, 2, "f32", 35),
                                                     (2048, 2, "f32", 3), (16384, 4, "f32", 2), (8192, 2, "f32", 2), (8192, 2, "f32", 37),
                                                     (16384, 2, "f32", 5), (16384, 4, "f32", 7), (1 << 15, 2, "f32", 9),
                                                     (64, 4, "f64", 70), (4096, 4, "f64", 3), (8192, 2, "f64", 2), (16384, 2, "f64", 2),
                                                     (4096, 2, "f64", 5), (8192, 2, "f64", 37), (16384, 2, "f64", 19),
                                                     # the two-pass sizes: the forward transform's pass 2 multiplies by h on its way out (two launches per
                                                     # chunk for these small batches; the persistent launch for the last four)
                                                     (1 << 16, 2, "f32", 3), (1 << 16, 4, "f32", 2), (1 << 17, 2, "f64", 2), (1 << 21, 2, "f32", 2),
                                                     (1 << 20, 2, "f32", 2), (1 << 16, 2, "f32", 600), (1 << 20, 2, "f32", 33), (1 << 18, 2, "f64", 70),
                                                     (1 << 22, 2, "f32", 9)])
def test_fast_convolution_matches_reference_composition(sd, torch_cuda, oracle, n, radix, precision, batch):
    """x <- IFFT(FFT(x) .* H).  Checker: the reference's own composition
    fft_radix<forward>(x); x *= H; fft_radix<reverse_fft>(x) through the oracle, in double."""
    torch = torch_cuda
    prec = sd.F32 if precision == "f32" else sd.F64
    cdt = np.complex64 if prec == sd.F32 else np.complex128
    rng = np.random.default_rng(n + batch)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(cdt)
    h = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    if n <= (1 << 16) and batch <= 300:
        spec = oracle.fft(x.astype(np.complex128), radix) * h.astype(np.complex128)
        want = oracle.fft(spec, radix, True)
    else:  # large cases: numpy (the oracle is cross-checked against it at these sizes in test_four_step_large_transforms)
        want = np.fft.ifft(np.fft.fft(x.astype(np.complex128), axis=-1) * h.astype(np.complex128), axis=-1)
    plan = sd.FftPlan(n, radix, sd.forward_fft, prec, max_batch=batch)
    tol = 2e-6 if prec == sd.F32 else 8 * n * EPS64  # two transforms and a product
    outs = []
    fused = n <= (16384 if prec == sd.F32 else 8192) or (radix == 2 and n == (1 << 15 if prec == sd.F32 else 16384))
    # two fused forms: the one-wave kernel, or (radix 2, N = 8192 / 16384) the registers-resident kernel of csrc/fft_big.hip, + the
    # register-pass MODE 3 as variant 2
    # (double, radix 2, N = 4096 / 8192 / 16384: csrc/fft_big64.hip's convolution form; MODE 3 of the f64 register-pass family = variant 2 up to 8192)
    two_fused = (prec == sd.F32 and (n in (256, 1024, 16384) or (radix == 2 and n in (512, 2048, 4096, 8192)))) or \
                (prec == sd.F64 and radix == 2 and n in (4096, 8192))
    for variant in ((0, 1, 2) if two_fused else (0, 1) if fused else (0,)):
        plan.set_variant(variant)  # f32 n <= 16384 (radix 2: 32768), f64 n <= 8192: 0 = fused single kernel, 1 = three launches
        d, hd = torch.from_numpy(x).cuda(), torch.from_numpy(h).cuda()
        plan.convolve(d, hd)
        torch.cuda.synchronize()
        outs.append(d.cpu().numpy())
        assert rel_max_err(outs[-1], want) < tol, (variant, rel_max_err(outs[-1], want))
    for other in outs[1:]:
        assert rel_max_err(outs[0], other) < (1e-6 if prec == sd.F32 else tol)
    with pytest.raises(sd.SdspHipError):
        sd.FftPlan(n, radix, sd.reverse_fft, prec).convolve(torch.from_numpy(x).cuda(), torch.from_numpy(h).cuda())


@pytest.mark.parametrize("n_real,radix,batch,precision", [(32, 2, 5, "f32"), (32, 4, 130, "f32"), (128, 4, 33, "f32"), (1024, 2, 7, "f32"),
                                                           (512, 2, 1027, "f32"), (512, 2, 2, "f32"), (512, 4, 1029, "f32"), (1024, 2, 130, "f32"), (2048, 2, 1030, "f32"), (4096, 2, 9, "f32"),
                                                           (2048, 4, 5, "f32"), (8192, 2, 3, "f32"), (8192, 4, 2, "f32"), (32768, 2, 2, "f32"), (32768, 2, 67, "f32"),
                                                           (16384, 2, 1, "f32"), (16384, 2, 131, "f32"), (32768, 4, 2, "f32"), (32768, 4, 41, "f32"), (8192, 4, 9, "f32"), (8192, 2, 300, "f32"), (65536, 2, 1, "f32"), (65536, 2, 19, "f32"),
                                                           (32, 2, 70, "f64"), (128, 4, 33, "f64"), (2048, 4, 5, "f64"), (16384, 2, 2, "f64"),
                                                           (8192, 2, 33, "f64"), (8192, 4, 3, "f64"), (16384, 2, 19, "f64"), (32768, 2, 1, "f64"), (32768, 2, 9, "f64")])
def test_real_input_packing(sd, torch_cuda, oracle, n_real, radix, batch, precision):
    """SURVEY 8(f)-3.  Checker: the reference algorithm on the real signal as a complex one (what the
    reference's own tests do, testFFT.cpp:23-25): bins 0..n_real/2 of oracle.fft(x + 0j).  f64: the reference's precision,
    held to its own bound 4 N eps."""
    torch = torch_cuda
    rng = np.random.default_rng(n_real + batch)
    f64 = precision == "f64"
    rdt, cdt = (np.float64, np.complex128) if f64 else (np.float32, np.complex64)
    TOL = 4 * n_real * EPS64 if f64 else TOL32
    prec = sd.F64 if f64 else sd.F32
    x = rng.standard_normal((batch, n_real)).astype(rdt)
    full = oracle.fft(x.astype(np.complex128), 2)  # any valid radix gives the same DFT
    half = n_real // 2
    want = full[:, :half].copy()
    want[:, 0] = full[:, 0].real + 1j * full[:, half].real  # packed: (X[0], X[N/2])
    fwd = sd.RfftPlan(n_real, radix, sd.forward_fft, max_batch=batch, precision=prec)
    d = torch.from_numpy(x).cuda()
    spec = fwd.exec(d)
    torch.cuda.synchronize()
    got = spec.cpu().numpy()
    assert got.shape == (batch, half)
    assert rel_max_err(got, want) < TOL, rel_max_err(got, want)
    # inverse: the oracle's packed spectrum back to the real samples, and the GPU round trip
    inv = sd.RfftPlan(n_real, radix, sd.reverse_fft, max_batch=batch, precision=prec)
    packed = torch.view_as_real(torch.from_numpy(want.astype(cdt)).cuda()).reshape(batch, n_real).contiguous()
    back = inv.exec(packed)
    torch.cuda.synchronize()
    assert rel_max_err(back.cpu().numpy(), x) < TOL
    again = inv.exec(torch.view_as_real(spec).reshape(batch, n_real))
    torch.cuda.synchronize()
    assert rel_max_err(again.cpu().numpy(), x) < 2 * TOL
    # f32, n_real = 1024: variant 0 is a one-wave kernel whose split / merge trades partners by ds_bpermute (csrc/fft_wave.hip:
    # real_pack_stage); variant 1 the register-pass family's in-LDS split: the same numbers to rounding.  (n_real = 512 / 2048: the family
    # is the default since its tiles are 2048 points, the one-wave kernels their variant 2)
    wave = not f64 and radix == 2 and half == 512  # where the wave kernels measured faster (capi.hip; round 3: only n_real = 1024 is left)
    # f32, radix 2, n_real / 2 = 8192 / 16384: split / merge inside the registers-resident kernel (csrc/fft_big.hip, REAL: the pairs
    # meet in LDS); variant 1 the register-pass family's, as above
    big = not f64 and ((radix == 2 and half in (2048, 4096, 8192, 16384, 32768)) or (radix == 4 and half in (4096, 16384)))
    # double, radix 2, n_real / 2 = 4096 / 8192 / 16384: the same inside csrc/fft_big64.hip (n_real = 32768 exists there only)
    big64 = f64 and radix == 2 and half in (4096, 8192, 16384)
    assert fwd.info.kernel.decode() == ("sdsp_fft1024_wave" if wave and half == 1024 else "sdsp_fft_wave_f32" if wave else
                                        "sdsp_fft_big_kernel" if big else "sdsp_fft_big_f64_real_kernel" if big64 else
                                        "sdsp_fft_reg_f64_kernel" if f64 else "sdsp_fft_reg_kernel")
    if wave or (big and half <= 16384) or (big64 and half <= 8192):  # n_real = 65536 (f64: 32768) exists in the registers-resident kernel only
        fwd.set_variant(1)
        inv.set_variant(1)
        assert fwd.info.kernel.decode() == ("sdsp_fft_reg_f64_kernel" if f64 else "sdsp_fft_reg_kernel")
        spec1 = fwd.exec(torch.from_numpy(x).cuda())
        back1 = inv.exec(torch.view_as_real(torch.from_numpy(want.astype(cdt)).cuda()).reshape(batch, n_real).contiguous())
        torch.cuda.synchronize()
        assert rel_max_err(spec1.cpu().numpy(), got) < (TOL if f64 else 1e-6)
        assert rel_max_err(back1.cpu().numpy(), back.cpu().numpy()) < (TOL if f64 else 1e-6)
    with pytest.raises(sd.SdspHipError):
        sd.RfftPlan(4096, 4)  # n_real/2 = 2048 is not a power of 4


def test_plan_twiddles_are_the_rounded_reference_row(sd, torch_cuda, fft_golden):
    # a4: the HBM-resident table equals the reference's last table row rounded once to fp32
    plan = sd.FftPlan(4096, 4, sd.forward_fft, sd.F32)
    assert np.array_equal(plan.twiddles(), fft_golden["wcoeffs4096_lastrow_fwd"].astype(np.complex64))
    plan64 = sd.FftPlan(4096, 2, sd.reverse_fft, sd.F64)
    assert np.array_equal(plan64.twiddles(), np.conj(fft_golden["wcoeffs4096_lastrow_fwd"]))
    info = plan.info
    assert info.algorithmic_bytes == 65536 and info.hbm_passes == 1


def test_host_pointer_and_sharded_entry_points(sd, torch_cuda, oracle):
    from simpledsp_amd import _lib as L
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((37, 1024)) + 1j * rng.standard_normal((37, 1024))).astype(np.complex128)
    want = oracle.fft(x, 2)
    a = x.copy()
    sd.fft_radix2(a)  # numpy array -> sdsp_hip_fft_exec_host
    assert rel_max_err(a, want) < _tol64(1024)
    plan = sd.FftPlan(1024, 2, sd.forward_fft, sd.F64)
    b = x.copy()
    arr = (C.c_void_p * 1)(plan._h)
    L.check(sd.load().sdsp_hip_fft_exec_sharded(arr, 1, b.ctypes.data, b.shape[0]))
    assert np.array_equal(a, b)
    assert sd.load().sdsp_hip_fft_exec(plan._h, None, 0, None) == 0  # empty batch is a no-op


def test_sharded_entry_with_several_plans(sd, torch_cuda, oracle):
    # contiguous batch ranges, one host thread + plan per shard, no collective.  With one
    # GPU the shards all live on device 0; the partitioning and threading are what is exercised.
    from simpledsp_amd import _lib as L
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((11, 4096)) + 1j * rng.standard_normal((11, 4096))).astype(np.complex64)
    want = oracle.fft(x.astype(np.complex128), 4)
    plans = [sd.FftPlan(4096, 4, sd.forward_fft, sd.F32, max_batch=11) for _ in range(3)]
    arr = (C.c_void_p * 3)(*[p._h for p in plans])
    y = x.copy()
    L.check(sd.load().sdsp_hip_fft_exec_sharded(arr, 3, y.ctypes.data, 11))
    assert rel_max_err(y, want) < TOL32
    bad = sd.FftPlan(1024, 4, sd.forward_fft, sd.F32)
    arr2 = (C.c_void_p * 2)(plans[0]._h, bad._h)
    assert sd.load().sdsp_hip_fft_exec_sharded(arr2, 2, y.ctypes.data, 11) == L.ERR_INVALID_ARG


@pytest.mark.parametrize("batch", [65536, 262144])
def test_baseline_config2_full_size_properties(sd, torch_cuda, oracle, batch):
    """BASELINE config 2 at full size (65536 x 4096 f32, 2 GiB) and one config-5 shard (2M transforms / 8 GPUs =
    262144 x 4096 f32, 8 GiB, on this one GPU): size-independent properties + spot checks against the oracle."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(0x5D5B)
    x = torch.randn((batch, 4096, 2), generator=g, device="cuda", dtype=torch.float32)
    x = torch.view_as_complex(x)
    y = x.clone()
    fwd = sd.FftPlan(4096, 4, sd.forward_fft, sd.F32, max_batch=batch)
    rev = sd.FftPlan(4096, 4, sd.reverse_fft, sd.F32, max_batch=batch)
    fwd.exec(y)
    # Parseval: sum|X|^2 = N * sum|x|^2 per transform
    ex = (x.abs() ** 2).sum(dim=1, dtype=torch.float64)
    ey = (y.abs() ** 2).sum(dim=1, dtype=torch.float64)
    assert float(((ey / (4096 * ex)) - 1).abs().max()) < 1e-5
    # spot checks against the oracle, first / last / random transforms
    idx = [0, 1, batch // 2, batch - 2, batch - 1] + list(np.random.default_rng(1).choice(batch, 11))
    want = oracle.fft(x[idx].cpu().numpy().astype(np.complex128), 4)
    assert rel_max_err(y[idx].cpu().numpy(), want) < TOL32
    # round trip: reverse(forward(x)) == x
    rev.exec(y)
    num = (y - x).abs().amax(dim=1)
    den = x.abs().amax(dim=1)
    assert float((num / den).max()) < TOL32


def test_convolve_rejects_real_input_plans(sd, torch_cuda):
    # a real-input plan's transform is not the complex DFT the product is defined on: UNSUPPORTED, not a wrong answer
    from simpledsp_amd import _lib as L
    plan = sd.RfftPlan(1024, 2, sd.forward_fft, max_batch=2)
    x = torch_cuda.zeros((2, 512), dtype=torch_cuda.complex64, device="cuda")
    h = torch_cuda.ones((512,), dtype=torch_cuda.complex64, device="cuda")
    rc = sd.load().sdsp_hip_fft_convolve(plan._h, x.data_ptr(), h.data_ptr(), 2, None)
    assert rc == L.ERR_UNSUPPORTED


def test_plans_on_a_second_device(sd, torch_cuda, oracle):
    """Kernels that need more than 64 KiB of dynamic LDS raise that limit per DEVICE (the attribute belongs to the
    function object of the current device): plans on device 1 must launch too.  Needs two GPUs; the driver's
    one-GPU box skips it."""
    torch = torch_cuda
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    rng = np.random.default_rng(77)
    for n, radix in ((8192, 2), (16384, 4), (1 << 20, 2)):
        x = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(np.complex64)
        want = np.fft.fft(x.astype(np.complex128), axis=-1)
        for dev in (0, 1):
            plan = sd.FftPlan(n, radix, sd.forward_fft, sd.F32, max_batch=2, device=dev)
            d = torch.from_numpy(x).to(f"cuda:{dev}")
            plan.exec(d)
            torch.cuda.synchronize(dev)
            assert rel_max_err(d.cpu().numpy(), want) < TOL32, (n, dev)
        with pytest.raises(ValueError):
            plan.exec(torch.from_numpy(x).to("cuda:0"))  # plan on device 1, tensor on device 0


@pytest.mark.parametrize("n,radix,precision,batch", [(4096, 4, "f32", 1000), (4096, 2, "f32", 777), (64, 2, "f32", 70001),
                                                     (8, 2, "f32", 20000), (8192, 0, "f32", 300), (32768, 2, "f32", 67),
                                                     (1024, 4, "f64", 1500), (100, 2, "f32", 0)])
def test_launch_pieces_are_bit_identical(sd, torch_cuda, n, radix, precision, batch):
    """sdsp_hip_set_launch_piece_bytes (include/sdsp_hip.h): a batch issued as several launches over consecutive pieces of
    the buffer gives the bits of the single launch -- ragged last pieces, pieces that are rounded to a workgroup's tile
    (N = 8: 4096 transforms), every single-launch kernel family."""
    torch = torch_cuda
    if n == 100:  # not a power of two: no plan; the knob itself round-trips
        old = sd.get_launch_piece_bytes()
        sd.set_launch_piece_bytes(12345)
        assert sd.get_launch_piece_bytes() == 12345
        sd.set_launch_piece_bytes(old)
        return
    prec = sd.F64 if precision == "f64" else sd.F32
    dt = torch.complex128 if precision == "f64" else torch.complex64
    g = torch.Generator(device="cuda").manual_seed(n + batch)
    x = torch.view_as_complex(torch.randn((batch, n, 2), generator=g, device="cuda", dtype=torch.float64 if precision == "f64" else torch.float32))
    old = sd.get_launch_piece_bytes()
    try:
        outs = []
        for piece in (0, 1 << 20, 3 << 20):
            sd.set_launch_piece_bytes(piece)
            for T in (sd.forward_fft, sd.reverse_fft):
                p = sd.FftPlan(n, radix, T, prec, max_batch=batch)
                d = x.clone().to(dt)
                p.exec(d)
                torch.cuda.synchronize()
                outs.append((piece, T.__name__, d.cpu().numpy()))
                p.close()
        for piece, tname, o in outs[2:]:
            ref = outs[0][2] if tname == "forward_fft" else outs[1][2]
            assert np.array_equal(o, ref), (n, radix, piece, tname)
    finally:
        sd.set_launch_piece_bytes(old)


@pytest.mark.parametrize("radix,batch", [(2, 1), (2, 257), (4, 3), (4, 1000)])
def test_one_wave_kernel_f64(sd, torch_cuda, oracle, radix, batch):
    """N = 1024 in double -- the reference's own test point (testFFT.cpp:239, BASELINE config 1) -- has the one-transform-per-wave
    kernel (csrc/fft_wave.hip) as its variant 1; the f64 register-pass kernel (variant 0, the default: the wave kernel gains
    nothing in double) computes the same arithmetic: same bits.
    Both are held to the reference's bound 4 N eps against the oracle; ragged batches (two waves per workgroup)."""
    torch = torch_cuda
    rng = np.random.default_rng(1024 + batch)
    x = rng.standard_normal((batch, 1024)) + 1j * rng.standard_normal((batch, 1024))
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        want = oracle.fft(x, radix, rev)
        plan = sd.FftPlan(1024, radix, T, sd.F64, max_batch=batch)
        assert plan.info.kernel.decode() == "sdsp_fft_reg_f64_kernel"
        outs = []
        for variant in (0, 1):
            plan.set_variant(variant)
            d = torch.from_numpy(x).cuda()
            guard = torch.full((64,), 7.0 + 3.0j, dtype=torch.complex128, device="cuda")
            plan.exec(d)
            torch.cuda.synchronize()
            outs.append(d.cpu().numpy())
            assert np.abs(outs[-1] - want).max() < _tol64(1024) * max(1.0, np.abs(want).max()), (radix, rev, variant)
            assert bool((guard == 7.0 + 3.0j).all())
        assert plan.info.kernel.decode() == "sdsp_fft1024_wave"
        assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("n,radix,precision,batch", [(4096, 4, "f32", 700), (256, 4, "f32", 9000), (1024, 2, "f64", 600), (32768, 2, "f32", 40)])
def test_convolve_launch_pieces_are_bit_identical(sd, torch_cuda, n, radix, precision, batch):
    """sdsp_hip_fft_convolve goes out in the same launch pieces as sdsp_hip_fft_exec (fused kernels, and at N = 32768 the
    three-launch composition, which is not split): the bits of the single launch."""
    torch = torch_cuda
    prec = sd.F64 if precision == "f64" else sd.F32
    rdt = torch.float64 if precision == "f64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(n * 3 + batch)
    x = torch.view_as_complex(torch.randn((batch, n, 2), generator=g, device="cuda", dtype=rdt))
    h = torch.view_as_complex(torch.randn((n, 2), generator=g, device="cuda", dtype=rdt))
    old = sd.get_launch_piece_bytes()
    try:
        outs = []
        for piece in (0, 1 << 20, 5 << 19):
            sd.set_launch_piece_bytes(piece)
            p = sd.FftPlan(n, radix, sd.forward_fft, prec, max_batch=batch)
            d = x.clone()
            p.convolve(d, h)
            torch.cuda.synchronize()
            outs.append(d)
            p.close()
        assert torch.equal(torch.view_as_real(outs[0]), torch.view_as_real(outs[1]))
        assert torch.equal(torch.view_as_real(outs[0]), torch.view_as_real(outs[2]))
    finally:
        sd.set_launch_piece_bytes(old)


# (n, radix, precision) -> (kernel, hbm_passes, stage_radix) of the DEFAULT variant: the size table of DESIGN.md section 1.
# plan_get_info and exec read the same select_kernel() (csrc/capi.hip), so this table is what runs.
STAGES_2_THEN_4 = 24
SIZE_TABLE = [
    (16, 2, "f32", "sdsp_fft_reg_kernel", 1, 2), (64, 4, "f32", "sdsp_fft_reg_kernel", 1, 4),
    (256, 2, "f32", "sdsp_fft_reg_kernel", 1, 2), (256, 4, "f32", "sdsp_fft_reg_kernel", 1, 4),
    (512, 2, "f32", "sdsp_fft_reg_kernel", 1, 2),
    (1024, 2, "f32", "sdsp_fft_reg_kernel", 1, 2), (1024, 4, "f32", "sdsp_fft_reg_kernel", 1, 4),
    (2048, 2, "f32", "sdsp_fft_reg_kernel", 1, 2),
    (4096, 2, "f32", "sdsp_fft4096_r2_f32", 1, 2), (4096, 4, "f32", "sdsp_fft4096_r4_f32", 1, 4),
    (8192, 2, "f32", "sdsp_fft_big_kernel", 1, 2), (8192, 0, "f32", "sdsp_fft_big_kernel", 1, 2),
    (16384, 2, "f32", "sdsp_fft_big_kernel", 1, 2), (16384, 4, "f32", "sdsp_fft_big_kernel", 1, 4),
    (32768, 2, "f32", "sdsp_fft_big_kernel", 1, 2),
    (1 << 16, 2, "f32", "sdsp_fft2p_cols+sdsp_fft2p_rows", 2, 2), (1 << 16, 4, "f32", "sdsp_fft2p_cols+sdsp_fft2p_rows", 2, 2),
    (1 << 18, 4, "f32", "sdsp_fft2p_cols+sdsp_fft2p_rows", 2, 2), (1 << 19, 2, "f32", "sdsp_fft2p_cols+sdsp_fft2p_rows", 2, 2),
    (1 << 20, 2, "f32", "sdsp_fft1m_fused", 2, 2), (1 << 20, 4, "f32", "sdsp_fft1m_fused", 2, 2),
    (64, 4, "f64", "sdsp_fft_reg_f64_kernel", 1, 4), (1024, 2, "f64", "sdsp_fft_reg_f64_kernel", 1, 2),
    (4096, 4, "f64", "sdsp_fft_big_f64_kernel", 1, 4), (4096, 2, "f64", "sdsp_fft_big_f64_kernel", 1, 2),
    (8192, 2, "f64", "sdsp_fft_big_f64_kernel", 1, 2), (16384, 2, "f64", "sdsp_fft_big_f64_kernel", 1, 2),
    (16384, 4, "f64", "sdsp_fft_big_f64_kernel", 1, 4), (1 << 15, 2, "f64", "sdsp_fft2p_cols+sdsp_fft2p_rows", 2, 2),
]


@pytest.mark.parametrize("n,radix,batch", [(4096, 2, 5), (8192, 2, 1), (8192, 2, 7), (16384, 2, 3), (4096, 4, 6), (16384, 4, 1), (16384, 4, 5)])
def test_registers_resident_kernel_f64(sd, torch_cuda, oracle, n, radix, batch):
    """fft_big64.hip (round 3): N = 4096 / 8192 / 16384 in double, transform in registers, one HBM pass -- radix-2 stages, and
    genuine radix-4 stages for sdsp::fft_radix4 plans of N = 4096 / 16384 (the reference's own precision and stage type);
    held to the reference's own bound 4 N eps against the oracle of the plan's radix in both directions, and to the kernel
    it replaces as the default."""
    torch = torch_cuda
    rng = np.random.default_rng(n + batch)
    x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
    v_big = 0
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        want = oracle.fft(x, radix, rev)
        plan = sd.FftPlan(n, radix, T, sd.F64, max_batch=batch)
        plan.set_variant(v_big)
        assert plan.info.kernel.decode() == "sdsp_fft_big_f64_kernel" and plan.info.hbm_passes == 1 and plan.info.stage_radix == radix
        d = torch.from_numpy(x).cuda()
        guard = torch.full((1 << 14,), 7.0 + 3.0j, dtype=torch.complex128, device="cuda")
        plan.exec(d)
        torch.cuda.synchronize()
        got = d.cpu().numpy()
        assert rel_max_err(got, want) < _tol64(n), (n, rev, rel_max_err(got, want))
        assert bool((guard == 7.0 + 3.0j).all())
        plan.set_variant(1 - v_big)  # what served the size before
        assert plan.info.kernel.decode() != "sdsp_fft_big_f64_kernel"
        d2 = torch.from_numpy(x).cuda()
        plan.exec(d2)
        torch.cuda.synchronize()
        assert rel_max_err(d2.cpu().numpy(), got) < _tol64(n)


@pytest.mark.parametrize("n,radix,precision,kernel,passes,stages", SIZE_TABLE)
def test_plan_info_reports_the_kernel_and_stage_type_that_run(sd, torch_cuda, oracle, n, radix, precision, kernel, passes, stages):
    """round 2 verdict, next #6: `info.stage_radix` is the butterfly type of the kernel that exec dispatches to (one table for
    both), an explicit radix-4 plan above N = 16384 says that its default kernel runs radix-2 butterflies, and variant 8 of
    such a plan (the coverage kernel) runs -- and reports -- genuine radix-4 stages; all of them are held to the oracle of the
    plan's radix."""
    torch = torch_cuda
    prec = sd.F64 if precision == "f64" else sd.F32
    plan = sd.FftPlan(n, radix, sd.forward_fft, prec, max_batch=32)
    info = plan.info
    assert info.kernel.decode() == kernel
    assert info.hbm_passes == passes
    assert info.stage_radix == stages
    assert info.radix == (radix or (4 if sd.isPowerOf4(n) and n != 16384 else 2))  # the plan's own (validated) radix
    rng = np.random.default_rng(n + radix)
    x = (rng.standard_normal((3, n)) + 1j * rng.standard_normal((3, n))).astype(np.complex128 if precision == "f64" else np.complex64)
    want = oracle.fft(x.astype(np.complex128), info.radix) if n <= 1 << 16 else np.fft.fft(x.astype(np.complex128), axis=-1)
    d = torch.from_numpy(x).cuda()
    plan.exec(d)
    torch.cuda.synchronize()
    assert rel_max_err(d.cpu().numpy(), want) < (TOL32 if precision == "f32" else _tol64(n))
    if radix == 4 and n >= 1 << 16:
        plan.set_variant(8)
        info = plan.info
        assert info.kernel.decode() == "sdsp_fft_tile_kernel" and info.stage_radix == 4
        d = torch.from_numpy(x).cuda()
        plan.exec(d)
        torch.cuda.synchronize()
        assert rel_max_err(d.cpu().numpy(), want) < TOL32


def test_mixed_radix_plan_reports_two_then_four(sd, torch_cuda):
    plan = sd.FftPlan(8192, 0, sd.forward_fft, sd.F32, max_batch=4)
    plan.set_variant(1)
    info = plan.info
    assert info.kernel.decode() == "sdsp_fft_mix_f32" and info.stage_radix == STAGES_2_THEN_4


def test_launch_count_query(sd, torch_cuda):
    """sdsp_hip_fft_plan_launches: what bench.py divides a step by (it used to re-derive the piece count by hand)"""
    piece = sd.get_launch_piece_bytes()
    try:
        sd.set_launch_piece_bytes(1 << 30)
        p = sd.FftPlan(4096, 4, sd.forward_fft, sd.F32, max_batch=16)
        assert p.launches(65536) == 2 and p.launches(262144) == 8 and p.launches(40000) == 1 and p.launches(0) == 0
        assert sd.FftPlan(16384, 2, sd.forward_fft, sd.F32, max_batch=16).launches(1 << 14) == 1  # never in pieces
        assert sd.FftPlan(1 << 20, 2, sd.forward_fft, sd.F32, max_batch=256).launches(256) == 1
        p2 = sd.FftPlan(1 << 16, 2, sd.forward_fft, sd.F32, max_batch=2048)
        assert p2.launches(2048) == 1  # the persistent two-pass launch (one per 1024 units of 8 MiB)
        assert p2.launches(2049) == 2 and p2.launches(4097) == 3  # max_batch = 2048 sized the counters: 128 units of 16 per launch
        p2.set_variant(3)
        assert p2.launches(2048) == 2 * 4  # chunks of 2^25 / n = 512 transforms, two passes each
        sd.set_launch_piece_bytes(0)
        assert p.launches(262144) == 1
    finally:
        sd.set_launch_piece_bytes(piece)


def test_fft1m_lost_handoff_is_reported(sd, torch_cuda):
    """a persistent N = 2^20 launch whose bounded hand-off wait gives up used to return OK from
    exec_host with invalid output.  With the wait bound forced to zero ticks every wait that has to poll gives up: the
    synchronous host entry must return an error, the asynchronous one must leave it for status(), and a later healthy call
    must clear it."""
    torch = torch_cuda
    n, batch = 1 << 20, 32
    plan = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch)
    assert plan.info.kernel.decode() == "sdsp_fft1m_fused"
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    plan.set_wait_limit(0)
    with pytest.raises(sd.SdspHipError) as e:
        plan.exec_host(x.copy())
    assert "gave up" in str(e.value)
    d = torch.from_numpy(x).cuda()
    plan.exec(d)  # asynchronous: returns OK ...
    with pytest.raises(sd.SdspHipError):
        plan.status()  # ... and the caller learns here
    plan.set_wait_limit(200_000_000)
    y = x.copy()
    plan.exec_host(y)  # a healthy call clears the sticky word and computes the right thing
    plan.status()
    want = np.fft.fft(x[[0, batch - 1]].astype(np.complex128), axis=-1)
    assert rel_max_err(y[[0, batch - 1]], want) < TOL32


# which variant of a two-pass plan is the persistent launch, which the two launches per chunk (capi.hip: select_kernel)
def _two_pass_variants(sd, n, prec):
    probe = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=max(16, (1 << 28) // (n * (16 if prec == sd.F64 else 8))))
    fused = 0 if probe.info.kernel.decode() == "sdsp_fft2p_fused" else 3
    return fused, 3 - fused


@pytest.mark.parametrize("n,precision,batch", [(1 << 16, "f32", 531), (1 << 17, "f32", 300), (1 << 18, "f32", 131), (1 << 19, "f32", 67),
                                               (1 << 21, "f32", 19), (1 << 22, "f32", 9),
                                               (1 << 15, "f64", 515), (1 << 16, "f64", 259), (1 << 17, "f64", 131), (1 << 18, "f64", 67), (1 << 19, "f64", 35), (1 << 20, "f64", 17)])
def test_two_pass_persistent_schedule(sd, torch_cuda, oracle, n, precision, batch):
    """fft_2pass.hip: the two passes in ONE persistent, ticketed launch (sdsp_fft2p_fused) do the same arithmetic as the two
    launches per chunk -> the same bits.  The batches are ragged against the ticket unit (a short last unit), leave the ticket
    queues uneven and are longer than the ring of intermediates, so slots are re-used within a call; the ring is also re-used
    across calls.  A few transforms are held to the oracle (numpy above 2^20, where the oracle takes seconds per transform)."""
    torch = torch_cuda
    f64 = precision == "f64"
    prec = sd.F64 if f64 else sd.F32
    fused, chunked = _two_pass_variants(sd, n, prec)
    g = torch.Generator(device="cuda").manual_seed(n % 1009 + batch)
    x = torch.view_as_complex(torch.randn((batch, n, 2), generator=g, device="cuda", dtype=torch.float64 if f64 else torch.float32))
    pick = [0, batch // 2, batch - 1]
    xs = x[pick].cpu().numpy().astype(np.complex128)
    tol = _tol64(n) if f64 else TOL32
    for T, rev in ((sd.forward_fft, False), (sd.reverse_fft, True)):
        want = oracle.fft(xs, 2, rev) if n <= (1 << 20) else (np.fft.ifft(xs, axis=-1) if rev else np.fft.fft(xs, axis=-1))
        plan = sd.FftPlan(n, 2, T, prec, max_batch=batch)
        plan.set_variant(fused)
        assert plan.info.kernel.decode() == "sdsp_fft2p_fused" and plan.info.hbm_passes == 2
        assert plan.launches(batch) == 1
        first = None
        for rep in range(2):
            y = x.clone()
            plan.exec(y)
            plan.exec(y.clone())  # a second call right behind it must not disturb the first one's result
            plan.status()  # synchronises; raises if a bounded wait of the in-kernel hand-off gave up
            assert rel_max_err(y[pick].cpu().numpy(), want) < tol, (rev, rep)
            if first is None:
                first = y
            else:
                assert torch.equal(torch.view_as_real(first), torch.view_as_real(y))
        plan.set_variant(chunked)
        assert plan.info.kernel.decode() == "sdsp_fft2p_cols+sdsp_fft2p_rows"
        z = x.clone()
        plan.exec(z)
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(first), torch.view_as_real(z)), rev
    # a plan made for a batch smaller than the ring of intermediates runs the two launches under either variant number
    small = sd.FftPlan(n, 2, sd.forward_fft, prec, max_batch=2)
    for variant in (0, 3):
        small.set_variant(variant)
        assert small.info.kernel.decode() == "sdsp_fft2p_cols+sdsp_fft2p_rows"


def test_two_pass_lost_handoff_is_reported(sd, torch_cuda):
    """The persistent two-pass launch shares sdsp_fft1m_fused's bounded waits and sticky abort word (handoff.h): with the wait
    bound forced to zero every wait gives up; exec_host must return the error, exec leaves it for status(), a healthy call clears it."""
    torch = torch_cuda
    n, batch = 1 << 18, 160
    fused, _ = _two_pass_variants(sd, n, sd.F32)
    plan = sd.FftPlan(n, 2, sd.forward_fft, sd.F32, max_batch=batch)
    plan.set_variant(fused)
    assert plan.info.kernel.decode() == "sdsp_fft2p_fused"
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    plan.set_wait_limit(0)
    with pytest.raises(sd.SdspHipError) as e:
        plan.exec_host(x.copy())
    assert "gave up" in str(e.value)
    d = torch.from_numpy(x).cuda()
    plan.exec(d)
    with pytest.raises(sd.SdspHipError):
        plan.status()
    plan.set_wait_limit(200_000_000)
    y = x.copy()
    plan.exec_host(y)
    plan.status()
    want = np.fft.fft(x[[0, batch - 1]].astype(np.complex128), axis=-1)
    assert rel_max_err(y[[0, batch - 1]], want) < TOL32
    # the convolution's two halves run persistent launches too (the reverse one on the plan's partner): a lost hand-off in either is this
    # plan's to report, and the next healthy call clears it
    h = torch.from_numpy((rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)).cuda()
    d = torch.from_numpy(x).cuda()
    plan.convolve(d, h)
    plan.status()
    ref = np.fft.ifft(np.fft.fft(x[[0]].astype(np.complex128), axis=-1) * h.cpu().numpy().astype(np.complex128), axis=-1)
    assert rel_max_err(d[[0]].cpu().numpy(), ref) < 2e-6
    plan.set_wait_limit(0)
    plan.convolve(d, h)
    with pytest.raises(sd.SdspHipError):
        plan.status()
    plan.set_wait_limit(200_000_000)
    d = torch.from_numpy(x).cuda()
    plan.convolve(d, h)
    plan.status()
    assert rel_max_err(d[[0]].cpu().numpy(), ref) < 2e-6


@pytest.mark.parametrize("n,radix,precision,batch", [(4096, 4, "f32", 40000), (1 << 17, 2, "f32", 300), (1 << 20, 2, "f32", 40), (16384, 2, "f64", 64)])
def test_exec_is_stream_capturable(sd, torch_cuda, n, radix, precision, batch):
    """include/sdsp_hip.h: a plan owns its workspace and counters from creation on, so sdsp_hip_fft_exec only enqueues (memsets of the
    persistent kernels' counters + launches) and can be captured into a graph.  Captured once, replayed: the same bits as eager calls --
    N = 4096 in two launch pieces, the persistent two-pass launch, the persistent N = 2^20 launch, the f64 registers-resident kernel."""
    torch = torch_cuda
    prec = sd.F64 if precision == "f64" else sd.F32
    piece = sd.get_launch_piece_bytes()
    try:
        sd.set_launch_piece_bytes(1 << 30)
        g0 = torch.Generator(device="cuda").manual_seed(n % 997 + batch)
        x = torch.view_as_complex(torch.randn((batch, n, 2), generator=g0, device="cuda", dtype=torch.float64 if prec == sd.F64 else torch.float32))
        fwd = sd.FftPlan(n, radix, sd.forward_fft, prec, max_batch=batch)
        rev = sd.FftPlan(n, radix, sd.reverse_fft, prec, max_batch=batch)
        want = x.clone()
        for _ in range(3):  # eager: also the first calls' one-time attribute / occupancy queries
            fwd.exec(want)
            rev.exec(want)
        fwd.status()
        y = x.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                fwd.exec(y)
                rev.exec(y)
        torch.cuda.current_stream().wait_stream(s)
        assert torch.equal(torch.view_as_real(y), torch.view_as_real(x))  # capturing ran nothing
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        fwd.status()
        rev.status()
        assert torch.equal(torch.view_as_real(y), torch.view_as_real(want))
    finally:
        sd.set_launch_piece_bytes(piece)


def test_bench_under_torch_distributed_run_on_one_gpu(sd, torch_cuda):
    """round 2 verdict, next #8: the N-rank path of bench.py (RCCL process group, throw-away barrier, barrier + max over
    ranks around the timed steps, ONE JSON line from rank 0) runs here as a FRESH child process tree under
    torch.distributed.run with one rank -- never an exec of this process."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    import socket
    with socket.socket() as sock:  # a port nobody holds right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, SDSP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-other-configs",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["batch_per_gpu"] == 65536 and out["roofline"]["launches_per_step"] == 2
    assert 0.5 < out["roofline"]["frac"] < 0.9, out["roofline"]
