"""The fp8 emulation of the oracle (oracle/fp8_emulation.py, written from the OCP format definition) against torch's own float8 casts:
two independent implementations of e4m3fn / e5m2 round-to-nearest-even with saturation, over EVERY bf16 value and random f32 values."""
import torch

from oracle import fp8_emulation as F8

TORCH = {F8.E4M3: torch.float8_e4m3fn, F8.E5M2: torch.float8_e5m2}


def _all_bf16():
    bits = torch.arange(0, 1 << 16, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    return x[torch.isfinite(x)]


def test_quantize_bits_matches_torch_float8_on_every_bf16_value_and_random_floats():
    g = torch.Generator().manual_seed(0)
    rnd = torch.randn(200000, generator=g) * torch.exp(torch.randn(200000, generator=g) * 4)
    for fmt in (F8.E4M3, F8.E5M2):
        for x in (_all_bf16(), rnd):
            xc = x.clamp(-F8.FMAX[fmt], F8.FMAX[fmt])
            mine = F8.quantize_bits(x, fmt)
            ref = xc.to(TORCH[fmt]).view(torch.uint8)
            same = mine == ref
            # +0 / -0 have distinct codes in both; torch keeps the sign of a value that rounds to zero, so does the definition
            assert bool(same.all()), (fmt, x[~same][:5], mine[~same][:5], ref[~same][:5])
            back = F8.dequantize_bits(mine, fmt)
            assert torch.equal(back, ref.view(TORCH[fmt]).float())


def test_every_code_round_trips():
    for fmt in (F8.E4M3, F8.E5M2):
        codes = torch.arange(256, dtype=torch.uint8)
        vals = codes.view(TORCH[fmt]).float()
        ok = torch.isfinite(vals)
        assert torch.equal(F8.dequantize_bits(codes[ok], fmt), vals[ok])
        assert torch.equal(F8.quantize_bits(vals[ok], fmt), codes[ok])


def test_scale_book_is_delayed_by_one_step():
    b = F8.ScaleBook()
    x0, x1 = torch.tensor([1.0, -3.0]), torch.tensor([10.0])
    s0 = b.scale("t", x0, F8.E4M3)
    assert abs(s0 - 2 * 3.0 / 448) < 1e-8          # first use: its own amax
    b.end_step()
    s1 = b.scale("t", x1, F8.E4M3)
    assert s1 == s0                                 # step 1 still scales by step 0's amax
    b.end_step()
    assert abs(b.scale("t", x0, F8.E4M3) - 2 * 10.0 / 448) < 1e-8
