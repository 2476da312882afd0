"""Model factory with the surface of dirtorch/nets/__init__.py:18-95.

    net = create_model('resnet101_rmac', pretrained='', out_dim=2048, pooling='gem', gemp=3, ...)
    net.load_state_dict(checkpoint['state_dict']); desc = net(x)      # x: [B,3,H,W] fp32 on the GPU

`model_names` holds the same 13 names as the reference: the *_rmac networks of the
descriptor hot path, the *_fpn_rmac two-level variants and the plain classification trunks, all
engine-backed.
"""
import os
from collections import OrderedDict

from . import rmac_resnet as _rmac

_FACTORIES = {name: getattr(_rmac, name) for name in (
    'resnet18', 'resnet50', 'resnet101', 'resnet152',
    'resnet18_rmac', 'resnet50_rmac', 'resnet101_rmac', 'resnet152_rmac',
    'resnet18_fpn_rmac', 'resnet50_fpn_rmac', 'resnet101_fpn_rmac', 'resnet101_fpn0_rmac',
    'resnet152_fpn_rmac')}
globals().update(_FACTORIES)          # `nets.resnet101_rmac(...)` works as in the reference
model_names = set(_FACTORIES)


def _strip_module_prefix(state_dict):
    """DataParallel checkpoints prefix every key with 'module.' (common.py:153)."""
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def create_model(arch, pretrained='', delete_fc=False, *args, **kwargs):
    """Instantiate architecture `arch` (NameError if unknown, like nets/__init__.py:37-39), attach
    the `preprocess` dict the loaders read, and optionally initialise from a checkpoint file."""
    factory = _FACTORIES.get(arch)
    if factory is None:
        raise NameError("unknown model architecture '%s'\nSelect one in %s"
                        % (arch, ','.join(sorted(model_names))))
    net = factory(*args, **kwargs)
    net.preprocess = {'mean': net.rgb_means, 'std': net.rgb_stds, 'input_size': max(net.input_size)}

    if pretrained:
        if not os.path.isfile(pretrained):
            # the reference would download ImageNet weights here (resnet.py:176-199): no network path
            raise NotImplementedError("pretrained='%s': only a checkpoint file path is supported" % pretrained)
        from ..utils.common import torch_load_trusted
        load_pretrained_weights(net, torch_load_trusted(pretrained)['state_dict'], delete_fc=delete_fc)
    return net


def load_pretrained_weights(net, state_dict, delete_fc=False):
    """Tolerant initialisation (nets/__init__.py:67-95): take every tensor of `state_dict` whose
    name and shape match the network; keep the network's own value for the rest and say so."""
    given = _strip_module_prefix(state_dict)
    merged = OrderedDict()
    for name, own in net.state_dict().items():
        cand = given.get(name)
        if cand is None:
            if not name.endswith('num_batches_tracked'):
                print("Loading weights for %s: Missing layer %s" % (type(net).__name__, name))
            merged[name] = own
        elif tuple(cand.shape) != tuple(own.shape):
            print("Loading weights for %s: Bad shape for layer %s, skipping" % (type(net).__name__, name))
            merged[name] = own
        else:
            merged[name] = cand
    # delete_fc: the reference deletes fc.* from its LOCAL new_dict after the merge (nets/__init__.py:90-93),
    # which changes nothing the caller can see; the caller's state_dict is left alone here too
    net.load_state_dict(merged)
