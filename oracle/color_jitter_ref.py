"""ColorJitterPoints restated for the CPU -- TEST INFRASTRUCTURE (only tests/ import this module).

Reference call site: pyrl/utils/augmentations/pcd_aug.py:269-303 -- `ColorJitterPoints.process_single` reshapes the
rgb key [B, 3, N] to an image batch [B, 3, 1, N] and calls `torchvision.transforms.ColorJitter(brightness, contrast,
saturation, hue)` on it; the replay stores rgb as uint8, so the uint8 code path of torchvision runs.

The arithmetic lives in a THIRD-PARTY dependency that is not in /root/reference and not installable here:
torchvision (the reference pins torch 1.13.1, README.md:38, whose companion release is torchvision 0.14.1).  What
follows restates the published algorithm of that release (torchvision/transforms/transforms.py::ColorJitter and
torchvision/transforms/functional_tensor.py::{_blend, rgb_to_grayscale, adjust_brightness, adjust_contrast,
adjust_saturation, adjust_hue, _rgb2hsv, _hsv2rgb}) with plain torch ops, in the same order and dtypes.

PARITY UNPINNED against torchvision itself (it cannot be imported in this container, so no fixture could be generated
from it); anchored on the reference's call site above: one parameter draw per call shared by the whole batch, a
per-cloud grayscale mean for the contrast step, uint8 truncation after every step -- and held to hand-derived known
answers of the published algorithm (tests/test_color_jitter_oracle.py: brightness clamp / truncation, grayscale weights,
saturation and contrast end points, per-cloud contrast mean, hue rotations of the primaries, step order, draw ranges).
"""
import torch


def draw_params(brightness, contrast, saturation, hue):
    """ColorJitter.__init__ + get_params: ranges [max(0, 1 - x), 1 + x] (hue: [-x, x]), None when the range is empty;
    one randperm(4) and up to four uniform draws from torch's global generator, in this order."""
    def rng(value, center, clip=True):
        lo, hi = center - float(value), center + float(value)
        if clip:
            lo = max(lo, 0.0)
        return None if lo == hi == center else (lo, hi)
    b, c, s, h = rng(brightness, 1.0), rng(contrast, 1.0), rng(saturation, 1.0), rng(hue, 0.0, clip=False)
    order = torch.randperm(4)
    fac = [None if r is None else float(torch.empty(1).uniform_(r[0], r[1])) for r in (b, c, s, h)]
    return [int(i) for i in order], fac


def _blend(img1, img2, ratio):
    ratio = float(ratio)
    bound = 1.0 if img1.is_floating_point() else 255.0
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, bound).to(img1.dtype)


def _gray(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / cr_divisor, (maxc - g) / cr_divisor, (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * f), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (s * (1.0 - f))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6, device=i.device).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, factor):
    orig = img.dtype
    if img.dtype == torch.uint8:
        img = img.to(dtype=torch.float32) / 255.0
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + factor) % 1.0
    out = _hsv2rgb(torch.stack((h, s, v), dim=-3))
    return (out * 255.0).to(dtype=orig) if orig == torch.uint8 else out


def color_jitter(rgb, order, factors):
    """rgb [B, 3, N] uint8 (or float in [0, 1]) -> same shape / dtype; order: permutation of (0 brightness, 1 contrast,
    2 saturation, 3 hue); factors: the four factors, None to skip a step."""
    img = rgb[:, :, None, :]
    for op in order:
        f = factors[op]
        if f is None:
            continue
        if op == 0:
            img = _blend(img, torch.zeros_like(img), f)
        elif op == 1:
            dtype = img.dtype if torch.is_floating_point(img) else torch.float32
            mean = torch.mean(_gray(img).to(dtype), dim=(-3, -2, -1), keepdim=True)
            img = _blend(img, mean, f)
        elif op == 2:
            img = _blend(img, _gray(img), f)
        else:
            img = adjust_hue(img, f)
    return img.squeeze(-2)
