"""Model factory - same surface as dirtorch/nets/__init__.py:18-95.

    net = create_model('resnet101_rmac', pretrained='', out_dim=2048, pooling='gem', gemp=3, ...)
    net.load_state_dict(checkpoint['state_dict']); desc = net(x)      # x: [B,3,H,W] fp32 on the GPU
"""
import os
from collections import OrderedDict

import torch

internal_funcs = set(globals().keys())

from .rmac_resnet import (resnet18_rmac, resnet50_rmac, resnet101_rmac, resnet152_rmac,  # noqa: E402
                          resnet18, resnet50, resnet101, resnet152,
                          resnet18_fpn_rmac, resnet50_fpn_rmac, resnet101_fpn_rmac,
                          resnet101_fpn0_rmac, resnet152_fpn_rmac)

# same rule as the reference (dirtorch/nets/__init__.py:18-21): every lowercase callable above
model_names = {name for name in globals()
               if name.islower() and not name.startswith("__")
               and name not in internal_funcs
               and callable(globals()[name])}


def create_model(arch, pretrained='', delete_fc=False, *args, **kwargs):
    """Create an (uninitialised-from-checkpoint) network; dirtorch/nets/__init__.py:24-64."""
    if arch not in model_names:
        raise NameError("unknown model architecture '%s'\nSelect one in %s" % (
                        arch, ','.join(sorted(model_names))))
    model = globals()[arch](*args, **kwargs)

    model.preprocess = dict(
        mean=model.rgb_means,
        std=model.rgb_stds,
        input_size=max(model.input_size)
    )

    if os.path.isfile(pretrained or ''):
        from ..utils.common import torch_load_trusted
        weights = torch_load_trusted(pretrained)['state_dict']
        load_pretrained_weights(model, weights, delete_fc=delete_fc)
    elif pretrained:
        # the reference downloads ImageNet weights here (resnet.py:176-199); there is no network
        # path in this engine
        raise NotImplementedError("pretrained='%s': only a checkpoint file path is supported" % pretrained)

    return model


def load_pretrained_weights(net, state_dict, delete_fc=False):
    """Load what matches, keep the network's own value for what is missing or mis-shaped
    (dirtorch/nets/__init__.py:67-95)."""
    new_dict = OrderedDict()
    for k, v in list(state_dict.items()):
        if k.startswith('module.'):
            k = k.replace('module.', '')
        new_dict[k] = v

    d = net.state_dict()
    for k, v in list(d.items()):
        if k not in new_dict:
            if not k.endswith('num_batches_tracked'):
                print("Loading weights for %s: Missing layer %s" % (type(net).__name__, k))
            new_dict[k] = v
        elif v.shape != new_dict[k].shape:
            print("Loading weights for %s: Bad shape for layer %s, skipping" % (type(net).__name__, k))
            new_dict[k] = v

    net.load_state_dict(new_dict)

    if delete_fc:
        fc = net.fc_name
        del new_dict[fc + '.weight']
        del new_dict[fc + '.bias']
