"""Test-time image transforms - the subset of dirtorch/utils/transforms.py the evaluation path uses
(create :11-37, Scale :133-185, Pad :47-75, PadSquare, CenterCrop, ToTensor/Normalize :617-623).

    trf = create('Scale(1050), CenterCrop(1024)', to_tensor=True, mean=..., std=...)
    trf(pil_image) -> tensor

The reference eval()s the comma-separated string; here it is parsed and only these transform names
with literal arguments are accepted.  With to_tensor='uint8' the chain ends in a raw uint8 HWC
tensor and normalisation runs on the GPU inside dir_forward (DIR_IMG_U8_NHWC): 4x less host->device
traffic than the reference's fp32 CHW feed.
"""
import ast

import numpy as np
import torch
from PIL import Image, ImageOps


class Scale(object):
    """Resize so that the smallest (largest=True: largest) side equals `size` (int), by a factor
    (float in ]0,4]) or to (w, h)."""

    def __init__(self, size, interpolation=Image.BILINEAR, largest=False, can_upscale=True, can_downscale=True):
        assert isinstance(size, (float, int)) or len(size) == 2
        if isinstance(size, float):
            assert 0 < size <= 4, 'bad float size, cannot be outside of range ]0,4]'
        self.size, self.interpolation, self.largest = size, interpolation, largest
        self.can_upscale, self.can_downscale = can_upscale, can_downscale

    def get_params(self, imsize):
        w, h = imsize
        if isinstance(self.size, int):
            w_is_ref = (w >= h) if self.largest else (w <= h)
            h_is_ref = (h >= w) if self.largest else (h <= w)
            if (w_is_ref and w == self.size) or (h_is_ref and h == self.size):
                return w, h
            if w_is_ref:
                return self.size, int(0.5 + self.size * h / w)
            return int(0.5 + self.size * w / h), self.size
        if isinstance(self.size, float):
            return int(0.5 + self.size * w), int(0.5 + self.size * h)
        return tuple(self.size)

    def target_size(self, imsize):
        """(w, h) the image has after this transform, honouring can_upscale / can_downscale."""
        imsize = tuple(imsize)
        size2 = tuple(self.get_params(imsize))
        grows, shrinks = min(imsize) < min(size2), min(imsize) > min(size2)
        if size2 != imsize and ((self.can_upscale and grows) or (self.can_downscale and shrinks)):
            return size2
        return imsize

    def __call__(self, img):
        size2 = self.target_size(img.size)
        return img if size2 == img.size else img.resize(size2, self.interpolation)


class Pad(object):
    """Pad the shortest side up to `size` (centred); larger images are untouched."""

    def __init__(self, size, color=(127, 127, 127)):
        self.size = size
        self.color = tuple(c if isinstance(c, int) else int(255 * c) for c in color)

    def __call__(self, img):
        w, h = img.size
        if w >= h:
            newh, neww = max(h, self.size), w
        else:
            newh, neww = h, max(w, self.size)
        if (neww, newh) != img.size:
            img = ImageOps.expand(img, border=((neww - w) // 2, (newh - h) // 2, neww - w - (neww - w) // 2,
                                               newh - h - (newh - h) // 2), fill=self.color)
        return img


class PadSquare(object):
    """Pad to a square of side max(w, h) (or `size` if larger), image centred."""

    def __init__(self, size=None, color=(127, 127, 127)):
        self.size = size
        self.color = tuple(c if isinstance(c, int) else int(255 * c) for c in color)

    def __call__(self, img):
        w, h = img.size
        s = self.size if self.size else max(w, h)
        neww, newh = max(w, s), max(h, s)
        if (neww, newh) != img.size:
            img = ImageOps.expand(img, border=((neww - w) // 2, (newh - h) // 2, neww - w - (neww - w) // 2,
                                               newh - h - (newh - h) // 2), fill=self.color)
        return img


class CenterCrop(object):
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        w, h = img.size
        tw, th = min(self.size[0], w), min(self.size[1], h)
        x0, y0 = int(round((w - tw) / 2.)), int(round((h - th) / 2.))
        return img.crop((x0, y0, x0 + tw, y0 + th))


class ToTensor(object):
    """PIL RGB -> float32 CHW in [0,1] (torchvision ToTensor semantics)."""

    def __call__(self, img):
        a = np.asarray(img, dtype=np.uint8)
        return torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div(255)


class Normalize(object):
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


class ToUint8HWC(object):
    """PIL RGB -> uint8 HWC tensor; ToTensor + Normalize then happen on the GPU (prep_input)."""

    def __call__(self, img):
        return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


class Compose(object):
    def __init__(self, trfs):
        self.transforms = list(trfs)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def device_scales(chains, **vars):
    """If every chain of `chains` is empty or one bilinear Scale(...), the list of those Scale
    objects (None for an empty chain): such chains can be run as ONE decode + upload per image and a
    bit-identical resize per scale on the GPU (ops.resize_bilinear_u8).  Otherwise None."""
    out = []
    for chain in chains:
        trfs = create(chain, to_tensor=False, **vars).transforms
        if len(trfs) == 0:
            out.append(None)
        elif len(trfs) == 1 and isinstance(trfs[0], Scale) and trfs[0].interpolation == Image.BILINEAR:
            out.append(trfs[0])
        else:
            return None
    return out


_ALLOWED = {c.__name__: c for c in (Scale, Pad, PadSquare, CenterCrop, ToTensor, Normalize)}


def create(cmd_line, to_tensor=False, **vars):
    """Build the transform chain from a string, e.g. "Scale(1050), CenterCrop(1024)".
    to_tensor: False | True (append ToTensor + Normalize(mean, std) like the reference)
               | 'uint8' (append ToUint8HWC; normalise on device)."""
    assert isinstance(cmd_line, str)
    chain = []
    if cmd_line.strip():
        try:
            tree = ast.parse('[%s]' % cmd_line, mode='eval').body
        except SyntaxError as e:
            raise SyntaxError("Cannot interpret this transform list: %s\nReason: %s" % (cmd_line, e))
        for node in tree.elts:
            if isinstance(node, ast.Name):
                node = ast.Call(func=node, args=[], keywords=[])
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _ALLOWED):
                raise SyntaxError("Cannot interpret this transform list: %s\nReason: only %s are supported"
                                  % (cmd_line, sorted(_ALLOWED)))

            def val(n):
                """Argument values: literals, the chain variables (mean / std / input_size),
                PIL's `Image.<FILTER>` constants and + - * / // of numbers - the subset of the
                reference's eval() that test-time chains use."""
                if isinstance(n, ast.Name) and n.id in vars:
                    return vars[n.id]
                if (isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == 'Image'
                        and n.attr.isupper()):
                    from PIL import Image
                    if hasattr(Image, n.attr):
                        return getattr(Image, n.attr)
                if isinstance(n, ast.BinOp) and isinstance(n.op, (ast.Add, ast.Sub, ast.Mult, ast.Div, ast.FloorDiv)):
                    a, b = val(n.left), val(n.right)
                    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
                        if isinstance(n.op, (ast.Div, ast.FloorDiv)) and not b:
                            raise SyntaxError("Cannot interpret this transform list: %s\nReason: division by zero" % cmd_line)
                        return {ast.Add: lambda: a + b, ast.Sub: lambda: a - b, ast.Mult: lambda: a * b,
                                ast.Div: lambda: a / b, ast.FloorDiv: lambda: a // b}[type(n.op)]()
                try:
                    return ast.literal_eval(n)
                except (ValueError, TypeError, SyntaxError) as e:
                    raise SyntaxError("Cannot interpret this transform list: %s\nReason: unsupported argument "
                                      "(%s)" % (cmd_line, e))
            chain.append(_ALLOWED[node.func.id](*[val(a) for a in node.args],
                                                **{k.arg: val(k.value) for k in node.keywords}))
    has_tensor = any(isinstance(t, ToTensor) for t in chain)
    if to_tensor == 'uint8' and not has_tensor:
        chain.append(ToUint8HWC())
    elif to_tensor and not has_tensor:
        chain += [ToTensor(), Normalize(mean=vars['mean'], std=vars['std'])]
    return Compose(chain)
