"""GPU parity tests for the cascaded-biquad path -- through the C ABI on a real MI355X.

The f64 kernels keep the reference's operation order and are held to BIT-EXACT agreement with
the reference (fixtures produced by the real reference, tests/golden/iir_golden.npz) and to the
reference's own 1e-12 against the Octave CSVs (testIIR.cpp:59).  The f32 kernels are held to the
normwise 1e-6 of SURVEY 8(d) on the BASELINE config-4 filter.  Block-by-block streaming must be
bit-identical to one long call (testIIR.cpp:61-75).
"""
import numpy as np
import pytest

from conftest import BAND_STOP_CASES, design, impulse_csvs, read_impulse_csv, rel_max_err, scipy_band_stop_sos

pytestmark = pytest.mark.gpu

KINDS = {"lp": 1, "hp": 2, "bp": 3}


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


def _bank(sd, m, channels, precision, kind, ftype, f0, fs, q, gain_in=1.0, variant=0):
    f = sd.casc_2o_iir(m, channels, precision, kind)
    design(f, ftype, f0, fs, q, gain_in)
    f.set_variant(variant)
    return f


def _process(torch, bank, x, **kw):
    dt = torch.float64 if bank.precision == 1 else torch.float32  # F32 and F32_F64STATE store floats
    d = torch.from_numpy(np.ascontiguousarray(x)).to(dt).cuda()
    bank.process(d, **kw)
    torch.cuda.synchronize()
    return d.cpu().numpy()


@pytest.mark.parametrize("csv", impulse_csvs(), ids=lambda p: p.stem)
def test_octave_impulse_responses_f64(sd, torch_cuda, iir_golden, csv):
    # testIIR.cpp:32-75 (+ :223-251, :304-332, :385-413)
    ftype, fs, f0, q, want = read_impulse_csv(csv)
    x = np.zeros((3, want.size))
    x[:, 0] = 1.0
    for kind, key in ((sd.IIR_GENERIC, "generic"), (ftype, "spec")):
        for variant in (0, 1, 2):
            bank = _bank(sd, 4, 3, sd.F64, kind, ftype, f0, fs, q, variant=variant)
            out = _process(torch_cuda, bank, x)
            assert np.abs(out - want).max() < 1e-12
            # the reference's own doubles, bit for bit
            assert np.array_equal(out[1], iir_golden[f"{csv.stem}__{key}"]), (key, variant)
        # streaming in 32-sample blocks + tail from a copy of the configured filter: identical
        bank2 = _bank(sd, 4, 3, sd.F64, kind, ftype, f0, fs, q)
        d = torch_cuda.from_numpy(x.copy()).cuda()
        n = want.size
        for off in range(0, n, 32):
            bank2.process(d, samples=min(32, n - off), offset=off)
        torch_cuda.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), out)


def test_final_state_matches_reference_ring(sd, torch_cuda, iir_golden):
    # m_mem / m_pos after 1000 samples (casc_2o_iir.h:78-79) vs the rotated device layout
    for tag in iir_golden["csv_names"]:
        ftype, fs, f0, q = iir_golden[f"{tag}__params"]
        bank = _bank(sd, 4, 2, sd.F64, sd.IIR_GENERIC, int(ftype), f0, fs, q)
        x = np.zeros((2, 1000))
        x[:, 0] = 1.0
        _process(torch_cuda, bank, x)
        st = bank.state.cpu().numpy()[:, 0].reshape(5, 3)  # [level][age]
        mem, pos = iir_golden[f"{tag}__generic_mem"], int(iir_golden[f"{tag}__generic_pos"])
        for age in range(3):
            assert np.array_equal(st[:, age], mem[:, (pos - 1 - age) % 3]), (tag, age)


@pytest.mark.parametrize("nm", ["lp", "hp", "bp"])
def test_gain_linearity_and_preload_f64(sd, torch_cuda, iir_golden, nm):
    ftype = KINDS[nm]
    fs, f0, q = 100e3, 10e3, 1.1
    imp = np.zeros((1, 1024))
    imp[0, 0] = 1.0
    outs = {}
    for gain_in in (1.0, 2.0):
        for kind, key in ((sd.IIR_GENERIC, "generic"), (ftype, "spec")):
            bank = _bank(sd, 4, 1, sd.F64, kind, ftype, f0, fs, q, gain_in)
            outs[gain_in, key] = _process(torch_cuda, bank, imp)[0]
            assert np.array_equal(outs[gain_in, key], iir_golden[f"gain_{nm}_{gain_in:g}__{key}"])
    assert np.abs(2.0 * outs[1.0, "generic"] - outs[2.0, "generic"]).max() < 1e-12  # testIIR.cpp:79-171
    # testIIR.cpp:173-218
    bank = _bank(sd, 4, 5, sd.F64, sd.IIR_GENERIC, ftype, f0, fs, q)
    bank.preload_filter(10.0)
    out = _process(torch_cuda, bank, np.full((5, 1024), 10.0))
    assert np.array_equal(out[3], iir_golden[f"preload_{nm}__out"])
    assert np.abs(out - (10.0 if nm == "lp" else 0.0)).max() < 1e-12


@pytest.mark.parametrize("m", [2, 4, 6, 8])
def test_section_counts_and_kinds_f64_bit_exact(sd, torch_cuda, iir_golden, m):
    if m == 4:
        src, tag, args = iir_golden["rand4096__in"], "rand4096", (10e3, 100e3, 1.1)
    else:
        src, tag, args = iir_golden["rand512__in"], f"rand512_m{m}", (3e3, 48e3, 0.9)
    x = np.tile(src, (70, 1))
    for nm, ftype in KINDS.items():
        for kind, key in ((sd.IIR_GENERIC, "generic"), (ftype, "spec")):
            bank = _bank(sd, m, 70, sd.F64, kind, ftype, args[0], args[1], args[2])
            out = _process(torch_cuda, bank, x)
            assert np.array_equal(out[0], iir_golden[f"{tag}_{nm}__{key}"]), (m, nm, key)
            assert np.array_equal(out[69], out[0])


@pytest.mark.parametrize("m", [10, 12, 16])
def test_more_than_eight_sections(sd, torch_cuda, oracle, m):
    """The reference accepts any even M (casc_2o_iir.h:25); 10 .. 16 sections run the direct kernel (correct, not
    tuned).  f64 bit-exact against the oracle's recurrence, every kind, both layouts; 18 is refused."""
    rng = np.random.default_rng(m)
    x = rng.standard_normal((70, 300))
    for nm, ftype in KINDS.items():
        for kind in (sd.IIR_GENERIC, ftype):
            fo = oracle.iir(m)
            design(fo, ftype, 3e3, 48e3, 0.9)
            want = fo.process(x[5], kind)
            bank = _bank(sd, m, 70, sd.F64, kind, ftype, 3e3, 48e3, 0.9)
            out = _process(torch_cuda, bank, x)
            assert np.array_equal(out[5], want), (m, nm, kind)
            bank_il = _bank(sd, m, 70, sd.F64, kind, ftype, 3e3, 48e3, 0.9)
            d = torch_cuda.from_numpy(np.ascontiguousarray(x.T)).cuda()
            bank_il.process_interleaved(d)
            torch_cuda.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy().T, out)
    b32 = _bank(sd, m, 70, sd.F32, sd.IIR_GENERIC, 1, 10e3, 100e3, 0.0)
    fo = oracle.iir(m)
    fo.set_lp_coeff(10e3, 100e3)
    # f32: every cascaded section adds its rounding noise -- the 1e-6 of the four-section BASELINE filter scaled by m / 4, x2
    assert rel_max_err(_process(torch_cuda, b32, x.astype(np.float32))[7], fo.process(x[7].astype(np.float32).astype(np.float64))) < 2e-6 * m / 4
    with pytest.raises(sd.SdspHipError):
        _process(torch_cuda, _bank(sd, 18, 2, sd.F64, sd.IIR_GENERIC, 1, 3e3, 48e3, 0.0), x[:2])


@pytest.mark.parametrize("channels,samples", [(1, 4096), (63, 128), (65, 96), (300, 4096), (1024, 1000), (257, 36)])
def test_f32_bank_against_oracle(sd, torch_cuda, oracle, channels, samples):
    # BASELINE config-4 filter: casc_2o_iir<4> LP, fs=100k, f0=10k (testIIR.cpp:469-474)
    rng = np.random.default_rng(channels * 7 + samples)
    x = rng.standard_normal((channels, samples)).astype(np.float32)
    pick = sorted(set([0, channels - 1] + list(rng.choice(channels, min(channels, 8)))))
    results = {}
    for variant in (0, 1, 2):
        bank = _bank(sd, 4, channels, sd.F32, sd.IIR_GENERIC, 1, 10e3, 100e3, 0.0, variant=variant)
        results[variant] = _process(torch_cuda, bank, x)
    for c in pick:
        fo = oracle.iir(4)
        fo.set_lp_coeff(10e3, 100e3)
        want = fo.process(x[c].astype(np.float64))
        assert rel_max_err(results[0][c], want) < 1e-6, (c, rel_max_err(results[0][c], want))
    # all kernel variants run the same arithmetic in the same order
    assert all(np.array_equal(results[0], results[v]) for v in (1, 2))


@pytest.mark.parametrize("csv", impulse_csvs(), ids=lambda p: p.stem)
def test_mixed_precision_octave_impulse_responses(sd, torch_cuda, csv):
    """SDSP_HIP_F32_F64STATE: float samples (the f32 kernels' 8 bytes per sample), double state and recurrence.  The 9
    Octave fixtures (testIIR.cpp:32-75) -- which the pure f32 kernels miss by up to 1.3e-4 at f0/fs = 200/39000 -- to float
    rounding; block streaming bit-identical; the sample-major layout bit-identical."""
    ftype, fs, f0, q, want = read_impulse_csv(csv)
    x = np.zeros((70, want.size), np.float32)
    x[:, 0] = 1.0
    for kind in (sd.IIR_GENERIC, ftype):
        bank = _bank(sd, 4, 70, sd.F32_F64STATE, kind, ftype, f0, fs, q)
        out = _process(torch_cuda, bank, x)
        assert out.dtype == np.float32
        assert rel_max_err(out[33], want) < 1.2e-7, rel_max_err(out[33], want)  # one rounding to float: 2^-24 = 6e-8
        bank2 = _bank(sd, 4, 70, sd.F32_F64STATE, kind, ftype, f0, fs, q)
        d = torch_cuda.from_numpy(x.copy()).cuda()
        for off in range(0, want.size, 32):
            bank2.process(d, samples=min(32, want.size - off), offset=off)
        torch_cuda.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), out)
        assert bank2.state.dtype == torch_cuda.float64
        bank3 = _bank(sd, 4, 70, sd.F32_F64STATE, kind, ftype, f0, fs, q)
        dw = torch_cuda.from_numpy(np.ascontiguousarray(x.T)).cuda()
        bank3.process_interleaved(dw)
        torch_cuda.cuda.synchronize()
        assert np.array_equal(dw.cpu().numpy().T, out)


def test_mixed_precision_random_input_and_variants(sd, torch_cuda, oracle):
    rng = np.random.default_rng(21)
    x = rng.standard_normal((300, 4096)).astype(np.float32)
    for f0 in (10e3, 500.0):  # the BASELINE filter, and a low cutoff where an f32 recurrence is off by 1e-4
        fo = oracle.iir(4)
        fo.set_lp_coeff(f0, 100e3)
        want = fo.process(x[7].astype(np.float64))
        outs = []
        for variant in (0, 1, 2):
            bank = _bank(sd, 4, 300, sd.F32_F64STATE, sd.IIR_GENERIC, 1, f0, 100e3, 0.0, variant=variant)
            outs.append(_process(torch_cuda, bank, x))
            assert rel_max_err(outs[-1][7], want) < 1.2e-7, (f0, variant, rel_max_err(outs[-1][7], want))
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
        f32 = _process(torch_cuda, _bank(sd, 4, 300, sd.F32, sd.IIR_GENERIC, 1, f0, 100e3, 0.0), x)
        if f0 == 500.0:  # what the mode is for
            assert rel_max_err(f32[7], want) > 20 * rel_max_err(outs[0][7], want)


@pytest.mark.parametrize("precision", ["f32", "f64", "mixed"])
def test_wide_supertile_kernel_is_bit_identical(sd, torch_cuda, precision):
    """variant 1 (csrc/iir.hip: sdsp_iir_wide_kernel -- 512 contiguous bytes of two channels per load / store instruction)
    takes whole super-tiles only (channels a multiple of 64, 512-byte multiples of samples); same arithmetic as variant 3,
    so the same bits, also across calls (per-channel state) and for every kind."""
    torch = torch_cuda
    prec = {"f32": sd.F32, "f64": sd.F64, "mixed": sd.F32_F64STATE}[precision]
    rng = np.random.default_rng(17)
    x = rng.standard_normal((192, 1024)).astype(np.float64 if precision == "f64" else np.float32)
    for nm, ftype in KINDS.items():
        for kind in (sd.IIR_GENERIC, ftype):
            outs = []
            for variant in (3, 1):
                bank = _bank(sd, 4, 192, prec, kind, ftype, 10e3, 100e3, 1.1, variant=variant)
                d = torch.from_numpy(x.copy()).cuda()
                assert bank.kernel_name(d, samples=512, offset=0) == ("sdsp_iir_wide_kernel" if variant == 1 else "sdsp_iir_supertile_kernel")
                bank.process(d, samples=512, offset=0)    # whole super-tiles: the wide kernel runs
                bank.process(d, samples=384, offset=512)  # f32: 1536 bytes, f64: 3072 bytes -- still whole super-tiles
                bank.process(d, samples=128, offset=896)
                torch.cuda.synchronize()
                outs.append((d.cpu().numpy(), bank.state.cpu().numpy()))
            assert np.array_equal(outs[0][0], outs[1][0]), (precision, nm, kind)
            assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("prec_name", ["f32", "f64"])
def test_block_streaming_is_stream_capturable(sd, torch_cuda, prec_name):
    """sdsp_hip_iir_process only launches: a block-by-block streaming loop (the reference's testIIR.cpp:61-75 pattern: state carried in the
    bank) captured into ONE graph and replayed gives the bits of the eager loop -- the launch-bound shape a caller would capture."""
    torch = torch_cuda
    prec = sd.F64 if prec_name == "f64" else sd.F32
    rng = np.random.default_rng(41)
    x = rng.standard_normal((256, 4096)).astype(np.float64 if prec == sd.F64 else np.float32)
    outs = []
    for captured in (False, True):
        bank = _bank(sd, 4, 256, prec, sd.IIR_GENERIC, 1, 10e3, 100e3, 1.1)
        d = torch.from_numpy(x.copy()).cuda()
        bank.process(d, samples=128, offset=0)  # first block eagerly (also: one-time queries)
        torch.cuda.synchronize()

        def blocks():
            for off in range(128, 4096, 128):
                bank.process(d, samples=128, offset=off)
        if captured:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                with torch.cuda.graph(graph, stream=s):
                    blocks()
            torch.cuda.current_stream().wait_stream(s)
            graph.replay()
        else:
            blocks()
        torch.cuda.synchronize()
        outs.append((d.cpu().numpy(), bank.state.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("m", [2, 4])
def test_landing_slot_kernel_is_bit_identical(sd, torch_cuda, m):
    """round 3: the default kernel of f32 banks with m_t <= 4 on whole [64 channels x 512 bytes] tiles is the landing-slot
    kernel (csrc/iir.hip: sdsp_iir_landing_kernel -- tile t + 1 arrives by LDS-DMA while tile t is filtered); same
    cascade_step, so the same bits as the super-tile kernel (variant 3), for every kind, across calls (per-channel state),
    and other shapes fall back to the super-tile / direct kernels inside the same call sequence."""
    torch = torch_cuda
    rng = np.random.default_rng(23)
    x = rng.standard_normal((192, 2048)).astype(np.float32)
    for nm, ftype in KINDS.items():
        for kind in (sd.IIR_GENERIC, ftype):
            outs = []
            for variant in (0, 3):
                bank = _bank(sd, m, 192, sd.F32, kind, ftype, 10e3, 100e3, 1.1, variant=variant)
                d = torch.from_numpy(x.copy()).cuda()
                want_k = "sdsp_iir_landing_kernel" if variant == 0 else "sdsp_iir_supertile_kernel"
                assert bank.kernel_name(d, samples=1024, offset=0) == want_k
                assert bank.kernel_name(d, samples=98, offset=1536) == "sdsp_iir_direct_kernel"  # 392 bytes: not 16-byte blocks
                assert bank.kernel_name(d, samples=96, offset=1664) == "sdsp_iir_supertile_kernel"  # aligned, not whole tiles
                bank.process(d, samples=1024, offset=0)   # eight whole tiles
                bank.process(d, samples=128, offset=1024)  # one whole tile
                bank.process(d, samples=384, offset=1152)  # three
                bank.process(d, samples=98, offset=1536)  # direct kernel
                bank.process(d, samples=30, offset=1634)  # direct kernel (the block starts off a 16-byte boundary)
                bank.process(d, samples=96, offset=1664)   # super-tile kernel, ragged tile
                bank.process(d, samples=288, offset=1760)
                torch.cuda.synchronize()
                outs.append((d.cpu().numpy(), bank.state.cpu().numpy()))
            assert np.array_equal(outs[0][0], outs[1][0]), (m, nm, kind)
            assert np.array_equal(outs[0][1], outs[1][1])
    # other precisions and deeper cascades keep the super-tile kernel as their default
    for prec, mm in ((sd.F64, 4), (sd.F32_F64STATE, 4), (sd.F32, 6)):
        bank = _bank(sd, mm, 192, prec, sd.IIR_GENERIC, 1, 10e3, 100e3, 1.1)
        d = torch.zeros((192, 1024), dtype=torch.float64 if prec == sd.F64 else torch.float32, device="cuda")
        assert bank.kernel_name(d) == "sdsp_iir_supertile_kernel"


def test_f32_specialised_kinds_and_streaming(sd, torch_cuda, oracle):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((130, 2048)).astype(np.float32)
    for nm, ftype in KINDS.items():
        bank = _bank(sd, 4, 130, sd.F32, ftype, ftype, 10e3, 100e3, 1.1)
        whole = _process(torch_cuda, bank, x)
        fo = oracle.iir(4)
        design(fo, ftype, 10e3, 100e3, 1.1)
        want = fo.process(x[77].astype(np.float64), ftype)
        assert rel_max_err(whole[77], want) < 1e-6
        # block streaming (aligned 256-sample blocks -> super-tile kernel; 100-sample -> direct kernel)
        for blk in (256, 100):
            bank2 = _bank(sd, 4, 130, sd.F32, ftype, ftype, 10e3, 100e3, 1.1)
            d = torch_cuda.from_numpy(x.copy()).cuda()
            for off in range(0, 2048, blk):
                bank2.process(d, samples=min(blk, 2048 - off), offset=off)
            torch_cuda.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy(), whole), (nm, blk)


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_interleaved_layout_is_bit_identical_to_channel_major(sd, torch_cuda, precision):
    # the sample-major "wire" layout -- same arithmetic, no transpose
    torch = torch_cuda
    prec = sd.F32 if precision == "f32" else sd.F64
    dt = torch.float32 if prec == sd.F32 else torch.float64
    rng = np.random.default_rng(3)
    for channels, samples in ((4, 50), (260, 333), (1024, 1000), (7, 40)):
        x = torch.from_numpy(rng.standard_normal((channels, samples))).to(dt).cuda()
        for nm, ftype in KINDS.items():
            for kind in (sd.IIR_GENERIC, ftype):
                ref = _bank(sd, 4, channels, prec, kind, ftype, 10e3, 100e3, 1.1)
                want = x.clone()
                ref.process(want)
                if channels % 2 and prec == sd.F64:
                    continue  # f64 rows must be 8-byte... every f64 shape is; odd counts are an f32 case
                for variant in (0, 1, 2):  # 16- / 8- / 4-byte lanes
                    bank = _bank(sd, 4, channels, prec, kind, ftype, 10e3, 100e3, 1.1, variant=variant)
                    y = x.t().contiguous()  # (samples, channels)
                    # stream it in ragged row blocks: state must carry across calls bit-exactly
                    off = 0
                    for blk in (7, 64, samples):
                        n = min(blk, samples - off)
                        if n > 0:
                            bank.process_interleaved(y, samples=n, offset=off)
                            off += n
                    torch.cuda.synchronize()
                    assert torch.equal(y.t(), want), (channels, samples, nm, kind, variant)
                    assert torch.equal(bank.state, ref.state)


@pytest.mark.parametrize("m", [2, 4, 8])
def test_band_stop_f64_and_f32(sd, torch_cuda, oracle, m):
    """SURVEY 8(f)-4 / README.md:15 TODO.  No reference code exists: the design is pinned to scipy's
    Butterworth band-stop at the reference's 1e-12 (testIIR.cpp:59); the recurrence itself is the reference's
    generic one, so the f64 kernel is bit-exact against the oracle run with the same coefficients."""
    import scipy.signal
    rng = np.random.default_rng(m)
    for f0, fs, q in BAND_STOP_CASES:
        bank = _bank(sd, m, 3, sd.F64, sd.IIR_GENERIC, 4, f0, fs, q)
        x = np.zeros((3, 1000))
        x[0, 0] = 1.0
        x[1:] = rng.standard_normal((2, 1000))
        out = _process(torch_cuda, bank, x)
        sos = scipy_band_stop_sos(m, f0, fs, q)
        assert np.abs(out[0] - scipy.signal.sosfilt(sos, x[0])).max() < 1e-12
        assert rel_max_err(out[1], scipy.signal.sosfilt(sos, x[1])) < 1e-10
        fo = oracle.iir(m)
        fo.set_design(bank.m_a_coeff, bank.m_b_coeff, bank.m_gain, 4)
        assert np.array_equal(out[2], fo.process(x[2], 0))
    # streaming, preload (DC passes a band-stop), f32 at the normwise 1e-6
    f0, fs, q = 10e3, 100e3, 1.1
    x = rng.standard_normal((70, 1024))
    whole = _process(torch_cuda, _bank(sd, m, 70, sd.F64, sd.IIR_GENERIC, 4, f0, fs, q), x)
    bank = _bank(sd, m, 70, sd.F64, sd.IIR_GENERIC, 4, f0, fs, q)
    parts = np.concatenate([_process(torch_cuda, bank, x[:, i:i + 64]) for i in range(0, 1024, 64)], axis=1)
    assert np.array_equal(whole, parts)
    bank = _bank(sd, m, 5, sd.F64, sd.IIR_GENERIC, 4, f0, fs, q)
    bank.preload_filter(10.0)
    assert np.abs(_process(torch_cuda, bank, np.full((5, 256), 10.0)) - 10.0).max() < 1e-9
    bank32 = _bank(sd, m, 70, sd.F32, sd.IIR_GENERIC, 4, f0, fs, q)
    x32 = x.astype(np.float32)
    got = _process(torch_cuda, bank32, x32)
    fo = oracle.iir(m)
    fo.set_design(bank32.m_a_coeff, bank32.m_b_coeff, bank32.m_gain, 4)
    assert rel_max_err(got[7], fo.process(x32[7].astype(np.float64), 0)) < 1e-6


def test_host_pointer_entry_point(sd, torch_cuda, iir_golden):
    import ctypes as C
    from simpledsp_amd import _lib as L
    lib = sd.load()
    f = sd.casc_2o_iir(4)
    f.set_lp_coeff(10e3, 100e3)
    h = C.c_void_p()
    L.check(lib.sdsp_hip_iir_plan_create(C.byref(h), 4, 0, f.m_a_coeff.ctypes.data, f.m_b_coeff.ctypes.data,
                                         f.m_gain, sd.F64, 0))
    x = np.tile(iir_golden["rand4096__in"], (3, 1)).copy()
    L.check(lib.sdsp_hip_iir_process_host(h, x.ctypes.data, 3, 4096, 4096, None))
    assert np.array_equal(x[2], iir_golden["rand4096_lp__generic"])
    y = np.tile(iir_golden["rand4096__in"], (4, 1)).copy()
    arr = (C.c_void_p * 1)(h)
    L.check(lib.sdsp_hip_iir_process_sharded(arr, 1, y.ctypes.data, 4, 4096))
    assert np.array_equal(y[3], x[0])
    lib.sdsp_hip_iir_plan_destroy(h)


def test_baseline_config4_full_size_properties(sd, torch_cuda, oracle):
    """BASELINE config 4 at full size: 1 048 576 channels x 4096 samples f32 (16 GiB), in place."""
    torch = torch_cuda
    channels, samples = 1 << 20, 4096
    g = torch.Generator(device="cuda").manual_seed(0x5D5B + 2)
    x = torch.randn((channels, samples), generator=g, device="cuda", dtype=torch.float32)
    idx = [0, 1, 63, 64, channels // 2 + 5, channels - 65, channels - 1]
    keep = x[idx].cpu().numpy()
    bank = _bank(sd, 4, channels, sd.F32, sd.IIR_GENERIC, 1, 10e3, 100e3, 0.0)
    bank.process(x)
    torch.cuda.synchronize()
    got = x[idx].cpu().numpy()
    for r, c in enumerate(idx):
        fo = oracle.iir(4)
        fo.set_lp_coeff(10e3, 100e3)
        assert rel_max_err(got[r], fo.process(keep[r].astype(np.float64))) < 1e-6, c
    assert bool(torch.isfinite(x).all())
    # unity DC gain of the low-pass: per-channel means survive the filter (size-independent check)
    del x


def test_randomised_cross_check(torch_cuda):
    """tests/fuzz_crosscheck.py: 150 random shapes / strides / offsets / variants over every C-ABI path (FFT, convolution,
    real-input, biquad banks, FIR banks) against numpy and the oracle."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "fuzz_crosscheck.py"), "7", "150"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

