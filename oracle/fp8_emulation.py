"""TEST INFRASTRUCTURE (the CPU checker of the fp8 convolution path; never imported by cvcs_amd).

Restates, in plain torch on the CPU, the numerics of the MI355X fp8 path (cvcs_amd/csrc/conv_fp8.hip):
  * OCP fp8 formats e4m3fn (max 448, no infinity) and e5m2 (max 57344), round to nearest even, SATURATING quantisation
    q = fmt(clamp(x / scale, +-fmax)) of a bf16-stored tensor with a per-tensor scale;
  * delayed scaling: scale(step t) = MARGIN * amax(step t-1) / fmax; a tensor's FIRST use calibrates on its own amax;
  * products exact, f32 accumulation, the result multiplied by scale_x * scale_w and stored as bf16.
The reference (theElandor/CVCS) has no reduced-precision path (everything is f32, source/scripts/train.py:121), so there is no reference
line to follow: BASELINE.json configs[4] ("mixed bf16/fp8 convs") names the feature, this file defines what "fp8" means for the parity
tests.  `quantize_bits` is written from the format definition (bit arithmetic) and pinned against torch's own float8 casts in
tests/test_fp8_oracle_cpu.py - two independent implementations."""
import torch

E4M3, E5M2 = 0, 1
FMAX = {E4M3: 448.0, E5M2: 57344.0}
MARGIN = 2.0
_SPEC = {E4M3: (4, 3, 7), E5M2: (5, 2, 15)}     # exponent bits, mantissa bits, bias


def quantize_bits(x: torch.Tensor, fmt: int) -> torch.Tensor:
    """f32 tensor (already divided by the scale) -> uint8 codes of the OCP format, saturating, round to nearest even.
    Written from the format definition: value = (-1)^s 2^(e-bias) (1 + m / 2^M) for e > 0, (-1)^s 2^(1-bias) m / 2^M for e = 0."""
    E, M, bias = _SPEC[fmt]
    x = x.double()
    sign = (torch.signbit(x)).to(torch.int64)
    a = x.abs().clamp(max=FMAX[fmt])
    emin = 1 - bias
    # exponent of the binade a falls into (at least the subnormal binade)
    e = torch.floor(torch.log2(a.clamp(min=2.0 ** (emin - M - 2)))).clamp(min=emin)
    step = torch.pow(torch.tensor(2.0, dtype=torch.float64), e - M)       # spacing of representable values in that binade
    n = a / step                                                           # in [2^M, 2^(M+1)) for normals, [0, 2^M) for subnormals
    r = torch.floor(n)
    frac = n - r
    r = r + ((frac > 0.5) | ((frac == 0.5) & (r % 2 == 1))).to(torch.float64)   # round half to even
    v = r * step                                                            # may have rounded up into the next binade: re-derive the fields
    e2 = torch.floor(torch.log2(v.clamp(min=2.0 ** (emin - M - 2)))).clamp(min=emin)
    m_full = v / torch.pow(torch.tensor(2.0, dtype=torch.float64), e2 - M)
    is_sub = m_full < 2 ** M
    ef = torch.where(is_sub, torch.zeros_like(e2), e2 + bias).to(torch.int64)
    mf = torch.where(is_sub, m_full, m_full - 2 ** M).to(torch.int64)
    code = (sign << (E + M)) | (ef << M) | mf
    code = torch.where(torch.isnan(x), torch.full_like(code, 0x7F), code)
    return code.to(torch.uint8)


def dequantize_bits(code: torch.Tensor, fmt: int) -> torch.Tensor:
    E, M, bias = _SPEC[fmt]
    c = code.to(torch.int64)
    s = (c >> (E + M)) & 1
    e = (c >> M) & ((1 << E) - 1)
    m = c & ((1 << M) - 1)
    mag = torch.where(e == 0, m.double() * 2.0 ** (1 - bias - M), (1.0 + m.double() / 2 ** M) * torch.pow(torch.tensor(2.0, dtype=torch.float64), (e - bias).double()))
    return torch.where(s == 1, -mag, mag).float()


def scale_from_amax(amax: float, fmt: int) -> float:
    """f32 arithmetic of cvcs_fp8_update_scales"""
    return float(torch.tensor(MARGIN, dtype=torch.float32) * torch.tensor(amax, dtype=torch.float32) / torch.tensor(FMAX[fmt], dtype=torch.float32))


def fake_quant(x: torch.Tensor, scale: float, fmt: int) -> torch.Tensor:
    """the DEQUANTISED value of x stored as fp8 with `scale`: scale * fmt(clamp(x * (1/scale))) - f32 multiply by the reciprocal, as the kernel does"""
    inv = (torch.tensor(1.0, dtype=torch.float32) / torch.tensor(scale, dtype=torch.float32))
    q = quantize_bits((x.float() * inv), fmt)
    return dequantize_bits(q, fmt) * torch.tensor(scale, dtype=torch.float32)


class ScaleBook:
    """delayed-scaling state of the emulation: per named tensor, the amax seen in the previous step"""

    def __init__(self):
        self.prev, self.cur = {}, {}

    def scale(self, name, x: torch.Tensor, fmt: int) -> float:
        am = float(x.detach().abs().max())
        self.cur[name] = max(self.cur.get(name, 0.0), am)
        base = self.prev.get(name, am)            # first use: calibrate on this very tensor
        return scale_from_amax(base if base > 0 else 1.0, fmt) if base > 0 else 1.0

    def end_step(self):
        for k, v in self.cur.items():
            if v > 0:
                self.prev[k] = v
        self.cur = {}
