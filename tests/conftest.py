import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'deep-image-retrieval_amd'), os.path.join(ROOT, 'oracle'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _cpus_allotted():
    """CPUs this process may actually run on: the cgroup quota when there is one, else the affinity mask.  (The GPU box shows
    256 logical CPUs and allots 16: torch's default thread count oversubscribes them and the CPU oracle - most of the GPU
    suite's wall time - runs several times slower than it has to.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(round(float(quota) / float(period)))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    try:
        import torch
        torch.set_num_threads(_cpus_allotted())
    except Exception:      # noqa: BLE001 - the thread count is an optimisation, never a reason to fail collection
        pass
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'experiments: exercises a kernel that only an experiments build of the library has '
                                       '(DIR_EXPERIMENTS=1 csrc/build.sh + DIRTORCH_AMD_LIB=.../libdir_engine_exp.so); '
                                       'DESELECTED - not skipped - when the library in use is a default build')


def _experiments_build():
    try:
        from dirtorch_amd import ops
        return '128x256_ring1x1' in ops.conv_variant_names()      # host-only query
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _experiments_build():
        return
    drop = [it for it in items if it.get_closest_marker('experiments')]
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = [it for it in items if not it.get_closest_marker('experiments')]


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment(monkeypatch):
    """The library reads its DIRTORCH_AMD_* A/B switches ONCE (dir_reload_env re-reads them; an engine copies them at
    dir_engine_create).  Tests flip them through monkeypatch: every such change is followed by a reload here, and the
    switches are restored together with the environment at teardown."""
    def reload():
        mod = sys.modules.get('dirtorch_amd._lib')
        if mod is not None:
            mod.reload_env()
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        orig_set(name, value, prepend)
        if name.startswith('DIRTORCH_AMD_'):
            reload()

    def delenv(name, raising=True):
        orig_del(name, raising)
        if name.startswith('DIRTORCH_AMD_'):
            reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    reload()


@pytest.fixture(scope='session')
def model_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'model_goldens.npz'))


@pytest.fixture(scope='session')
def postproc_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'postproc_goldens.npz'), allow_pickle=True)


@pytest.fixture(scope='session')
def head_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'head_goldens.npz'))


@pytest.fixture(scope='session')
def label_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'label_goldens.npz'))


@pytest.fixture(scope='session')
def qe_goldens():
    import numpy as np
    return np.load(os.path.join(GOLDEN, 'qe_goldens.npz'))
