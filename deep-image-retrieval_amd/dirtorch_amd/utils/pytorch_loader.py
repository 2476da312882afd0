"""Image loader for extraction - same call as dirtorch/utils/pytorch_loader.py:11-73.

    loader = get_loader(dataset, trf_chain='', iscuda=True, preprocess=net.preprocess,
                        output=['img'], batch_size=1, threads=8, shuffle=False)
    for inputs in loader: imgs = inputs[0]

Decode + geometric transforms run in DataLoader worker processes (PIL); the tensor handed to the
GPU is raw uint8 HWC by default (device_normalize=True) so that ToTensor/Normalize are fused into
the engine's first kernel.
"""
import torch
import torch.utils.data as data

from . import transforms


class PytorchLoader(data.Dataset):
    """dataset[i] -> [image tensor] (+ 'label'/'img_key' if asked), pytorch_loader.py:78-180."""

    def __init__(self, dataset, transform=None, output=('img',)):
        self.dataset, self.transform, self.output = dataset, transform, tuple(output)

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        res = []
        for o in self.output:
            if o == 'img':
                img = self.dataset.get_image(index)
                res.append(self.transform(img) if self.transform else img)
            elif o == 'img_key':
                res.append(self.dataset.get_key(index))
            elif o == 'label':
                res.append(self.dataset.get_label(index, toint=True))
            else:
                raise ValueError('unknown output %r' % o)
        return res


def get_loader(dataset, trf_chain, iscuda, preprocess={}, output=('img', 'label'), batch_size=None,
               threads=1, shuffle=True, device_normalize=True, **_unused):
    trf = transforms.create(trf_chain, to_tensor='uint8' if device_normalize else True, **preprocess)
    loader = PytorchLoader(dataset, transform=trf, output=output)
    # threads <= 1: load in the main process, but ALWAYS batched (the reference returns the bare
    # dataset for threads == 1, which breaks net(x)'s 4-D input; SURVEY.md §3.2 caveat)
    return data.DataLoader(loader, batch_size=batch_size or 1, shuffle=shuffle,
                           num_workers=threads if threads > 1 else 0, pin_memory=bool(iscuda))
