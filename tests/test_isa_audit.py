"""Emitted-code audit of the LDS-DMA ring kernels (no GPU needed: hipcc cross-compiles to gfx950 assembly).

Every ring kernel hands a stage from its loading side to its reading side with `s_waitcnt vmcnt(N)` + `s_barrier`.
hipcc 7.2 does not treat that pair as a fence for LDS reads: in `conv_patch3x3_kernel<64>` it had hoisted the first
weight-fragment reads of stage t above the wait and the barrier (found in round 3 as a one-in-2400 irreproducible launch
when forwards overlapped on several HIP streams - csrc/dir_common.h `ring_barrier`).  The fix is a barrier bracketed by
`__builtin_amdgcn_sched_barrier(0)`; these tests keep a raw `s_barrier` from coming back: no kernel source calls the
builtin directly, and in the assembly of every ring kernel at least one barrier sits between two
`; sched_barrier mask(0x00000000)` markers while no barrier that follows a hand-written `s_waitcnt` (an inline-asm
block) is left without them."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'deep-image-retrieval_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
RING_SOURCES = ['conv_igemm', 'conv_pair', 'conv_patch', 'conv_patchlc', 'conv_patchw', 'conv_patchs2', 'conv_persist', 'conv_persistlc', 'conv_ring', 'conv_seam3', 'conv_wreg', 'conv_wregd', 'conv_c3c1lc',
                'sim_split', 'stem_pool', 'stem_u8']
RING_KERNELS = re.compile(r'conv_igemm_kernel|conv_patch3x3\w*_kernel|conv_patch64_lc_kernel|conv1x1_persist_kernel|conv1x1_ring_kernel|'
                          r'conv1x1_wreg_kernel|conv1x1_lc_kernel|conv1x1_wregd_kernel|conv_c3c1ds_lc_kernel|sim_split\w*_kernel|whiten_split_kernel|stem_pool_persist_kernel|conv_pair_kernel|conv_pair_patch64_kernel|conv_seam3_kernel|stem_pool_pair_persist_kernel|stem_pool_u8_kernel')


def _asm(name, out_dir):
    out = os.path.join(out_dir, name + '.s')
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only',
                    os.path.join(CSRC, name + '.hip'), '-o', out], check=True, capture_output=True)
    return open(out).read()


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason='hipcc not available')
def test_ring_kernels_fence_their_hand_off_barriers(tmp_path):
    with ThreadPoolExecutor(max_workers=len(RING_SOURCES)) as ex:
        texts = list(ex.map(lambda n: _asm(n, str(tmp_path)), RING_SOURCES))
    checked, kernels = 0, 0
    for name, text in zip(RING_SOURCES, texts):
        func, body = None, []
        funcs = []
        for line in text.split('\n'):
            m = re.match(r'^(_ZN3dir\w+):', line)
            if m:
                func, body = m.group(1), []
                funcs.append((func, body))
            elif func is not None:
                body.append(line.strip())
        for func, body in funcs:
            if not RING_KERNELS.search(func):
                continue
            kernels += 1
            lines = [l for l in body if l and not l.startswith('.')]
            fenced_here = 0
            for i, l in enumerate(lines):
                if not l.startswith('s_barrier'):
                    continue
                prev = lines[i - 1] if i else ''
                nxt = lines[i + 1] if i + 1 < len(lines) else ''
                fenced = prev.startswith('; sched_barrier mask(0x00000000)') and nxt.startswith('; sched_barrier mask(0x00000000)')
                after_asm_wait = any(t.startswith(';;#ASMEND') for t in lines[max(0, i - 3):i])
                assert fenced or not after_asm_wait, '%s (%s.hip): raw s_barrier behind a hand-written wait' % (func, name)
                fenced_here += int(fenced)
            assert fenced_here >= 1, '%s (%s.hip): no fenced hand-off barrier' % (func, name)
            checked += fenced_here
    assert kernels >= 40 and checked >= 60, (kernels, checked)


def test_no_kernel_calls_the_raw_barrier_builtin():
    import glob
    for f in glob.glob(os.path.join(CSRC, '*.hip')):
        assert '__builtin_amdgcn_s_barrier' not in open(f).read(), '%s: use ring_barrier() (dir_common.h)' % os.path.basename(f)
