import torch


def bits(t):
    return t.contiguous().view(torch.int16).int()


def ulp_stats(a: torch.Tensor, b: torch.Tensor):
    """(max ulp distance, fraction differing) between two bf16 tensors (sign-magnitude aware)."""
    def key(t):
        x = bits(t)
        return torch.where(x < 0, -(x & 0x7FFF), x)
    d = (key(a.cpu()) - key(b.cpu())).abs()
    return int(d.max()), float((d > 0).float().mean())


def assert_close_bf16(a, b, max_ulp=1, max_frac=0.02, abs_floor=0.0, rel_floor=0.0, what=""):
    """bf16 tensors agree within max_ulp (elements whose abs diff is below abs_floor are exempt: near zero an
    ulp is meaninglessly small)."""
    a, b = a.cpu(), b.cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    af, bf = a.float(), b.float()
    assert torch.isfinite(af).all(), f"{what}: non-finite values"
    def key(t):
        x = bits(t)
        return torch.where(x < 0, -(x & 0x7FFF), x)
    d = (key(a) - key(b)).abs()
    # rel_floor: differences below rel_floor * mean|b| are exempt too -- outputs that cancel to ~0 carry the
    # absolute fp32 accumulation noise of the whole dot product, which is many ulps of a tiny value
    floor = max(abs_floor, rel_floor * float(bf.abs().mean()))
    small = (af - bf).abs() <= floor
    d = torch.where(small, torch.zeros_like(d), d)
    mx, frac = int(d.max()), float((d > 0).float().mean())
    assert mx <= max_ulp, f"{what}: max ulp diff {mx} (> {max_ulp}); max abs diff {(af - bf).abs().max().item():.4g}"
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ (> {max_frac})"
