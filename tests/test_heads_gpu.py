"""The FPN heads (dirtorch/nets/rmac_resnet_fpn.py) and the plain classifier (backbones/resnet.py) on
the MI355X, against the REFERENCE's own outputs (tests/golden/head_goldens.npz)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_oracle_golden import HEAD_CASES, head_case_inputs, head_oracle  # noqa: E402

FACTORY = {'fpn': '%s_fpn_rmac', 'fpn0': '%s_fpn0_rmac', 'cls': '%s'}


def make_net(head, arch, opts, sd, dtype):
    from dirtorch_amd import nets
    net = nets.create_model(FACTORY[head] % arch, pretrained='', **opts)
    net.load_state_dict(sd)
    net.compute_dtype = dtype
    net.cuda()
    return net.eval()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16', 'fp16p'])
@pytest.mark.parametrize('case', HEAD_CASES, ids=[c[0] for c in HEAD_CASES])
def test_head_vs_reference_golden(case, dtype, head_goldens):
    import dir_oracle as O
    tag, head, arch, opts, B, H, W = case
    sd, x = head_case_inputs(*case)
    net = make_net(head, arch, opts, sd, dtype)
    assert list(net.state_dict().keys()) == list(sd.keys())       # the reference's key set and order
    got = net(x.cuda()).cpu().numpy()
    gold = head_goldens[tag + '.desc']
    assert got.shape == gold.shape, (got.shape, gold.shape)
    assert np.isfinite(got).all()
    cos = O.cosine(got, gold)
    assert np.all(1 - cos < 1e-4), '%s %s: 1-cos = %s' % (tag, dtype, 1 - cos)
    if head == 'cls':
        # logits are not normalised: compare magnitudes too, against the 16-bit emulation
        emu = head_oracle(sd, head, arch, opts, x, quant=dtype).numpy()      # (fp16p: pairs in the head, fp16 after)
        tol = 2e-2 if dtype == 'bf16' else 3e-3
        assert np.abs(got - emu).max() < tol * np.abs(emu).max()
    else:
        np.testing.assert_allclose(np.linalg.norm(got.reshape(-1, got.shape[-1]), axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 7, 5, 4, 3, 64), (1, 14, 14, 7, 7, 256), (3, 9, 16, 5, 8, 8),
                                   (1, 64, 64, 32, 32, 1024)])
def test_upsample_add_matches_interpolate(shape, dtype):
    from dirtorch_amd import ops
    B, H, W, h, w, C = shape
    g = torch.Generator().manual_seed(H * 131 + w)
    x = torch.randn(B, H, W, C, generator=g).to(dtype)
    low = torch.randn(B, h, w, C, generator=g).to(dtype)
    up = torch.nn.functional.interpolate(low.float().permute(0, 3, 1, 2), size=(H, W), mode='nearest')
    ref = (x.float() + up.permute(0, 2, 3, 1)).to(dtype)
    got = ops.upsample_add(x.cuda(), low.cuda()).cpu()
    assert torch.equal(got, ref)        # one add in fp32, one rounding: bit-exact


def test_fpn_pooling_other_than_gem_fails_like_the_reference():
    from dirtorch_amd import nets
    net = nets.create_model('resnet18_fpn_rmac', pooling='max')    # constructs, as in the reference
    with pytest.raises(AttributeError):
        net(torch.zeros(1, 3, 64, 64, device='cuda'))


def test_fpn_batch_composition_is_irrelevant():
    import dir_oracle as O
    sd = O.synth_state_dict('resnet50', seed=9, out_dim=3072, gemp=2.6, head='fpn')
    net = make_net('fpn', 'resnet50', {}, sd, 'bf16')
    x = O.synth_images(3, 3, 96, 72).cuda()
    full = net(x).cpu()
    assert torch.equal(full, net(x).cpu())
    # The kernel a layer runs on depends on its pixel count (split-K, the small-map tiles - since round 6 conv_small.hip's two-accumulator
    # tile from 32 tiles of 64 x 64 up): a batch-1 forward re-associates some layers' fp32 sums, which flips a 16-bit rounding here and
    # there - the image alone and inside a batch agree to the storage format's noise, not bit for bit (bit-identity holds run to run and
    # stream to stream at equal shapes)
    for i in range(3):
        one = net(x[i:i + 1]).cpu()
        assert float(1 - torch.dot(one, full[i])) < 5e-6 and float((one - full[i]).abs().max()) < 2e-4
