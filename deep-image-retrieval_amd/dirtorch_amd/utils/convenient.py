"""mkdir helper with the reference's signature (dirtorch/utils/convenient.py:11-23)."""
import os


def mkdir(d, isfile=False):
    if isfile:
        d = os.path.split(d)[0]
    if d and not os.path.isdir(d):
        os.makedirs(d)
