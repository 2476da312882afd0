"""Device-side `Scale` (dir_resize_bilinear_u8) against Pillow itself and the oracle: bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import RESIZE_CASES  # noqa: E402


@pytest.mark.parametrize('shape', RESIZE_CASES + [(1024, 1024, 724, 724), (600, 800, 849, 1131)],
                         ids=lambda c: '%dx%d_to_%dx%d' % c)
def test_resize_matches_pillow(shape):
    from PIL import Image
    import dir_oracle as O
    from dirtorch_amd import ops
    h, w, oh, ow = shape
    img = np.random.RandomState(h * 1000 + ow).randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    got = ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), (ow, oh)).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), int(np.abs(got.astype(int) - ref.astype(int)).max())
    if h * w <= 100000:
        assert np.array_equal(got, O.resize_bilinear_u8(img, ow, oh))


def test_resize_batch_and_smooth_images():
    """A batch shares one launch; smooth gradients exercise the rounding rather than the clipping."""
    from PIL import Image
    from dirtorch_amd import ops
    yy, xx = np.mgrid[0:150, 0:210]
    base = np.stack([(yy * 1.7) % 256, (xx * 1.2) % 256, ((xx + yy) * 0.6) % 256], -1).astype(np.uint8)
    batch = np.stack([base, base[::-1].copy(), np.roll(base, 17, 1)])
    for ow, oh in ((297, 212), (105, 75), (210, 75), (300, 150)):
        got = ops.resize_bilinear_u8(torch.from_numpy(batch).cuda(), (ow, oh)).cpu().numpy()
        for i in range(3):
            ref = np.asarray(Image.fromarray(batch[i]).resize((ow, oh), Image.BILINEAR))
            assert np.array_equal(got[i], ref), (ow, oh, i)


def test_scale_transform_on_device_equals_cpu_transform():
    """transforms.Scale.get_params + the device resize == the PIL Scale object (float and int sizes)."""
    from PIL import Image
    from dirtorch_amd import ops
    from dirtorch_amd.utils import transforms
    img = np.random.RandomState(4).randint(0, 256, (123, 167, 3)).astype(np.uint8)
    pil = Image.fromarray(img)
    for arg in (1.414, 0.707, 200, 64):
        trf = transforms.Scale(arg)
        ref = np.asarray(trf(pil))
        ow, oh = trf.get_params(pil.size)
        got = ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), (ow, oh)).cpu().numpy()
        assert np.array_equal(got, ref), arg
