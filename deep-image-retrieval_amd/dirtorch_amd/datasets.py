"""The datasets the descriptor path needs, with the reference's interface
(dirtorch/datasets/: dataset.py:8-116, generic.py:13-30,124-250, create.py:5-29, oxford.py, paris.py).

    db = datasets.create('ROxford5K')                       # needs DB_ROOT
    db = datasets.create('ImageList("list.txt")')
    db.get_image(i) -> PIL.Image ; db.get_key(i) ; len(db) ; db.get_query_db()
    db.eval_query_AP(q, scores) -> {'easy','medium','hard'} (revisited protocol) or a float

Only evaluation-time datasets are here; training sets, splits and the wget downloader of the
reference are outside the hot path.
"""
import ast
import os
import pickle

import numpy as np


class Dataset(object):
    """Minimal base class: an indexable collection of image files."""
    root = ''
    img_dir = ''
    nimg = 0
    nclass = 0
    nquery = 0

    def __len__(self):
        return self.nimg

    def get_key(self, img_idx):
        raise NotImplementedError()

    def get_filename(self, img_idx, root=None):
        return os.path.join(root or self.root or '', self.img_dir, self.get_key(img_idx))

    def get_image(self, img_idx, resize=None):
        from PIL import Image
        img = Image.open(self.get_filename(img_idx)).convert('RGB')
        if resize:
            down = np.prod(resize) < np.prod(img.size)
            img = img.resize(resize, Image.LANCZOS if down else Image.BICUBIC)
        return img

    def get_query_db(self):
        raise NotImplementedError()

    def eval_query_AP(self, query_idx, scores):
        raise NotImplementedError()

    def eval_query_top(self, query_idx, scores, k=(1, 5, 10, 20, 50, 100)):
        raise NotImplementedError()   # no labels on the retrieval benchmarks (dataset.py:97)

    def original(self):
        return self

    def __repr__(self):
        res = 'Dataset: %s\n  %d images' % (type(self).__name__, len(self))
        try:
            res += ', %d queries' % self.get_query_db().nimg
        except NotImplementedError:
            pass
        return res + '\n  root: %s...' % self.root


class ImageList(Dataset):
    """A text file with one image path per row (generic.py:13-30)."""

    def __init__(self, img_list_path=None, root='', imgs=None):
        self.root = root
        self.imgs = list(imgs) if imgs is not None else [e.strip() for e in open(img_list_path) if e.strip()]
        self.nimg = len(self.imgs)

    def get_key(self, i):
        return self.imgs[i]


class ImageListROIs(Dataset):
    """Query images cropped to their bounding box (generic.py:227-250)."""

    def __init__(self, root, img_dir, imgs, rois):
        self.root, self.img_dir, self.imgs, self.rois = root, img_dir, imgs, rois
        self.nimg = len(imgs)

    def get_key(self, i):
        return self.imgs[i]

    def get_roi(self, i):
        return self.rois[i]

    def get_image(self, img_idx, resize=None):
        from PIL import Image
        img = Image.open(self.get_filename(img_idx)).convert('RGB').crop(self.rois[img_idx])
        if resize:
            down = np.prod(resize) < np.prod(img.size)
            img = img.resize(resize, Image.LANCZOS if down else Image.BICUBIC)
        return img


def compute_average_precision(positive_ranks):
    """Trapezoidal AP of the revisited Oxford/Paris protocol (utils/evaluation.py:46-82):
    positive_ranks = sorted zero-based ranks of the positives among the non-junk images."""
    n = len(positive_ranks)
    if not n:
        return 0.0
    ap = 0.0
    for i, rank in enumerate(positive_ranks):
        left = 1.0 if not rank else i / rank
        ap += (left + (i + 1) / (rank + 1)) / (2.0 * n)
    return ap


class ImageListRelevants(Dataset):
    """Images + queries + per-query relevant/junk index lists from a revisitop-style pickle
    (generic.py:124-224).  gt = {'imlist', 'qimlist', 'gnd': [{'bbx', 'easy', 'hard', 'junk'} or
    {'bbx', 'ok', 'junk'}]}."""

    def __init__(self, gt_file, root=None, img_dir='jpg', ext='.jpg'):
        self.root, self.img_dir = root, img_dir
        with open(gt_file, 'rb') as f:
            gt = pickle.load(f)

        def with_ext(e):
            return e if os.path.splitext(e)[1] else e + ext
        self.imgs = [with_ext(e) for e in gt['imlist']]
        self.qimgs = [with_ext(e) for e in gt['qimlist']]
        self.qroi = [tuple(e['bbx']) for e in gt['gnd']]
        if 'ok' in gt['gnd'][0]:
            self.relevants = [e['ok'] for e in gt['gnd']]
        else:
            self.relevants = None
            self.easy = [e['easy'] for e in gt['gnd']]
            self.hard = [e['hard'] for e in gt['gnd']]
        self.junk = [e['junk'] for e in gt['gnd']]
        self.nimg, self.nquery = len(self.imgs), len(self.qimgs)

    def get_key(self, i):
        return self.imgs[i]

    def get_query_key(self, i):
        return self.qimgs[i]

    def get_query_roi(self, i):
        return self.qroi[i]

    def get_query_db(self):
        return ImageListROIs(self.root, self.img_dir, self.qimgs, self.qroi)

    def get_relevants(self, q, mode='classic'):
        return {'classic': lambda: self.relevants[q], 'easy': lambda: self.easy[q],
                'medium': lambda: list(self.easy[q]) + list(self.hard[q]),
                'hard': lambda: self.hard[q]}[mode]()

    def get_junk(self, q, mode='classic'):
        return {'classic': lambda: self.junk[q],
                'easy': lambda: list(self.junk[q]) + list(self.hard[q]),
                'medium': lambda: self.junk[q],
                'hard': lambda: list(self.junk[q]) + list(self.easy[q])}[mode]()

    def get_query_groundtruth(self, q, what='AP', mode='classic'):
        res = -np.ones(self.nimg, dtype=np.int8)   # negatives
        res[self.get_relevants(q, mode)] = 1        # positives
        res[self.get_junk(q, mode)] = 0             # junk: removed before ranking
        return res

    def _ap(self, q, scores, mode):
        gt = self.get_query_groundtruth(q, 'AP', mode)
        assert gt.shape == scores.shape, "scores should have shape %s" % str(gt.shape)
        keep = gt != 0
        if mode != 'classic' and np.sum(gt[keep] > 0) == 0:
            return -1     # queries without positives are excluded from the mean
        gt, scores = gt[keep], scores[keep]
        order = np.argsort(scores)[::-1]            # ties: descending index, as the reference
        return compute_average_precision(np.where(gt[order] == 1)[0])

    def eval_query_AP(self, query_idx, scores):
        if self.relevants:
            return self._ap(query_idx, scores, 'classic')
        return {mode: self._ap(query_idx, scores, mode) for mode in ('easy', 'medium', 'hard')}


def _db_root():
    try:
        return os.environ['DB_ROOT']
    except KeyError:
        raise KeyError('DB_ROOT is not set: it must point at the directory holding oxford5k/ and paris6k/')


def _benchmark(cls_name, sub, pkl):
    def __init__(self):
        root = os.path.join(_db_root(), sub)
        ImageListRelevants.__init__(self, os.path.join(root, pkl), root=root)
    return type(cls_name, (ImageListRelevants,), {'__init__': __init__})


Oxford5K = _benchmark('Oxford5K', 'oxford5k', 'gnd_oxford5k.pkl')        # datasets/oxford.py
ROxford5K = _benchmark('ROxford5K', 'oxford5k', 'gnd_roxford5k.pkl')
Paris6K = _benchmark('Paris6K', 'paris6k', 'gnd_paris6k.pkl')            # datasets/paris.py
RParis6K = _benchmark('RParis6K', 'paris6k', 'gnd_rparis6k.pkl')

_REGISTRY = {c.__name__: c for c in (ImageList, ImageListRelevants, Oxford5K, ROxford5K, Paris6K, RParis6K)}


def create(dataset_cmd):
    """Instantiate a dataset from a string such as 'ROxford5K' or 'ImageList("imgs.txt")'
    (datasets/create.py:19-29).  The reference eval()s the string; here the call is parsed and only
    literal arguments are accepted."""
    if '(' not in dataset_cmd:
        dataset_cmd += '()'
    try:
        call = ast.parse(dataset_cmd.strip(), mode='eval').body
        assert isinstance(call, ast.Call) and isinstance(call.func, ast.Name)
        args = [ast.literal_eval(a) for a in call.args]
        kwargs = {k.arg: ast.literal_eval(k.value) for k in call.keywords}
    except (SyntaxError, ValueError, AssertionError) as e:
        raise SyntaxError('cannot interpret dataset command %r (%s)' % (dataset_cmd, e))
    if call.func.id not in _REGISTRY:
        raise NameError('unknown dataset %s; available: %s' % (call.func.id, ', '.join(sorted(_REGISTRY))))
    return _REGISTRY[call.func.id](*args, **kwargs)
