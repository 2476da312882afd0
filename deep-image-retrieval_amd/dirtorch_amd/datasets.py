"""The datasets the descriptor path needs, with the reference's interface
(dirtorch/datasets/: dataset.py:8-116, generic.py:13-30,124-250, create.py:5-29, oxford.py, paris.py).

    db = datasets.create('ROxford5K')                       # needs DB_ROOT
    db = datasets.create('ImageList("list.txt")')
    db = datasets.create('ImageListLabels("list_with_labels.txt")')   # class-label AP / top-k
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

    def get_label(self, img_idx, toint=False):
        raise NotImplementedError()

    def has_label(self):
        try:
            self.get_label(0)
            return True
        except NotImplementedError:
            return False

    def get_query_db(self):
        raise NotImplementedError()

    # ---- class-label ground truth (dataset.py:71-105): images of the query's class are relevant ---
    labels = None
    c_relevant_idx = None

    def get_query_groundtruth(self, query_idx, what='AP'):
        query_db = self.get_query_db()
        assert self.nclass == query_db.nclass
        if what == 'label':
            return query_db.get_label(query_idx)
        if what != 'AP':
            raise ValueError("Unknown ground-truth type: %s" % what)
        gt = -np.ones(self.nimg, dtype=np.int8)                              # negatives
        gt[self.c_relevant_idx.get(query_db.get_label(query_idx), [])] = 1   # same class (maybe none)
        if query_db is self:
            gt[query_idx] = 0                                                # the query itself: ignored
        return gt

    def eval_query_AP(self, query_idx, scores):
        """sklearn AP of `scores` against the class ground truth; -1 when the query has no relevant
        image (the caller leaves such queries out of the mean)."""
        if self.c_relevant_idx is None:
            raise NotImplementedError()
        from .utils.evaluation import compute_AP
        gt = self.get_query_groundtruth(query_idx, 'AP')
        assert gt.shape == scores.shape, "scores should have shape %s" % str(gt.shape)
        keep = gt != 0
        if not (gt[keep] > 0).any():
            return -1
        return compute_AP(gt[keep] > 0, scores[keep])

    def eval_query_top(self, query_idx, scores, k=(1, 5, 10, 20, 50, 100)):
        """1.0 / 0.0 per k: is an image of the query's class among the k best-scored ones."""
        if not self.labels:
            raise NotImplementedError()   # no labels on the retrieval benchmarks (dataset.py:97)
        q_label = self.get_query_groundtruth(query_idx, 'label')
        correct = np.array([l == q_label for l in self.labels], dtype=bool)[np.argsort(-scores)]
        return {k_: float(correct[:k_].any()) for k_ in k if k_ < len(correct)}

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


from .utils.evaluation import compute_average_precision  # noqa: E402,F401  (kept importable from here)


def find_and_list_classes(labels, cls_idx=None):
    """classes = [name of class 0, name of class 1, ...], cls_idx = {name: index}: classes are numbered
    in order of first appearance, after the indices forced through `cls_idx` (generic_func.py:8-44)."""
    assert not isinstance(labels, set), 'labels must be ordered'
    cls_idx = dict(cls_idx) if cls_idx else {}
    present = set(labels)
    for label in cls_idx:
        assert label in present, "error: missing forced label '%s'" % str(label)
    free = (i for i in range(len(present)) if i not in set(cls_idx.values()))
    for label in labels:
        if label not in cls_idx:
            cls_idx[label] = next(free)
    by_index = {i: c for c, i in cls_idx.items()}
    assert sorted(by_index) == list(range(len(by_index))), 'class indices must be 0..n-1'
    return [by_index[i] for i in range(len(by_index))], cls_idx


def find_relevants(labels):
    """{label: [indices of the images carrying it]} (generic_func.py:46-60)."""
    assert not isinstance(labels, set), 'labels must be ordered'
    rel = {}
    for i, label in enumerate(labels):
        rel.setdefault(label, []).append(i)
    return rel


class LabelledDataset(Dataset):
    """Per-image class labels (generic.py:33-41)."""

    def find_classes(self, *arg, **cls_idx):
        labels = arg[0] if arg else self.labels
        self.classes, self.cls_idx = find_and_list_classes(labels, cls_idx=cls_idx)
        self.nclass = len(self.classes)
        self.c_relevant_idx = find_relevants(self.labels)


def _read_pairs(path):
    rows = [e.strip().split(' ') for e in open(path) if e.strip()]
    return [r[0] for r in rows], [r[1] for r in rows]


class ImageListLabels(LabelledDataset):
    """'<image path> <label>' per row of a .txt file, or a {path: label} .json; the images are their
    own queries (generic.py:44-77)."""

    def __init__(self, img_list_path, root=None):
        self.root = root
        if os.path.splitext(img_list_path)[1] == '.json':
            import json
            pairs = json.load(open(img_list_path))
            self.imgs, self.labels = list(pairs.keys()), list(pairs.values())
        else:
            self.imgs, self.labels = _read_pairs(img_list_path)
        self.find_classes()
        self.nimg = len(self.imgs)
        self.nquery = 0

    def get_key(self, i):
        return self.imgs[i]

    def get_label(self, i, toint=False):
        return self.cls_idx[self.labels[i]] if toint else self.labels[i]

    def get_query_db(self):
        return self


class ImagesAndLabels(ImageListLabels):
    """In-memory image and label lists sharing another dataset's class indices (generic.py:108-121)."""

    def __init__(self, imgs, labels, cls_idx, root=None):
        self.root, self.imgs, self.labels, self.cls_idx = root, imgs, labels, cls_idx
        self.nclass = len(cls_idx)
        self.nimg = len(imgs)
        self.nquery = 0


class ImageListLabelsQ(ImageListLabels):
    """Database and query lists in two '<path> <label>' files (generic.py:80-105)."""

    def __init__(self, img_list_path, query_list_path, root=None):
        self.root = root
        self.imgs, self.labels = _read_pairs(img_list_path)
        self.qimgs, self.qlabels = _read_pairs(query_list_path)
        self.find_classes()
        self.nimg = len(self.imgs)
        self.nquery = len(self.qimgs)

    def find_classes(self, *arg, **cls_idx):
        labels = arg[0] if arg else self.labels + self.qlabels
        self.classes, self.cls_idx = find_and_list_classes(labels, cls_idx=cls_idx)
        self.nclass = len(self.classes)
        self.c_relevant_idx = find_relevants(self.labels)

    def get_query_db(self):
        return ImagesAndLabels(self.qimgs, self.qlabels, self.cls_idx, root=self.root)


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



def _listed(cls_name, base, sub, lst):
    """A named list dataset under DB_ROOT/<sub>/ (datasets/landmarks.py, landmarks18.py)."""
    def __init__(self):
        root = os.path.join(_db_root(), sub)
        base.__init__(self, os.path.join(root, lst), root + '/')
    return type(cls_name, (base,), {'__init__': __init__})


# name -> (labelled?, sub-directory, list file): extraction / whitening-set / distractor lists
_LISTS = {
    'Landmarks_clean': (True, 'landmarks', 'annotations/annotation_clean_train.txt'),
    'Landmarks_clean_val': (True, 'landmarks', 'annotations/annotation_clean_val.txt'),
    'Landmarks_lite': (True, 'landmarks', 'annotations/extra_landmark_images.txt'),
    'Landmarks18_train': (True, 'landmarks18', 'lists/train.txt'),
    'Landmarks18': (True, 'landmarks18', 'lists/train_all.txt'),
    'Landmarks18_lite': (True, 'landmarks18', 'lists/train_lite.txt'),
    'Landmarks18_mid': (True, 'landmarks18', 'lists/train_mid.txt'),
    'Landmarks18_5K': (True, 'landmarks18', 'lists/train_5K.txt'),
    'Landmarks18_val': (True, 'landmarks18', 'lists/val.txt'),
    'Landmarks18_valdstr': (True, 'landmarks18', 'lists/val_distractors.txt'),
    'Landmarks18_index': (False, 'landmarks18', 'lists/index.txt'),
    'Landmarks18_new_index': (False, 'landmarks18', 'lists/index_new.txt'),
    'Landmarks18_test': (False, 'landmarks18', 'lists/test.txt'),
    'Landmarks18_pca': (False, 'landmarks18', 'lists/train_pca.txt'),
    'Landmarks18_missing_index': (False, 'landmarks18', 'lists/missing_index.txt'),
}
_LISTED = {name: _listed(name, ImageListLabels if lab else ImageList, sub, lst)
           for name, (lab, sub, lst) in _LISTS.items()}
globals().update(_LISTED)

_REGISTRY = {c.__name__: c for c in (ImageList, ImageListLabels, ImageListLabelsQ, ImageListRelevants,
                                     Oxford5K, ROxford5K, Paris6K, RParis6K) + tuple(_LISTED.values())}


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
