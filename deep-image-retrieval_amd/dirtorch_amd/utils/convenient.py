"""mkdir helper with the reference's signature (dirtorch/utils/convenient.py:11-23)."""
import os


def mkdir(d, isfile='auto'):
    """Create directory d (or the directory of file d).  isfile='auto': d is a file path when it has an
    extension (the reference's splitext rule)."""
    if isfile == 'auto':
        isfile = bool(os.path.splitext(d)[1])
    if isfile:
        d = os.path.split(d)[0]
    if d and not os.path.isdir(d):
        os.makedirs(d)
