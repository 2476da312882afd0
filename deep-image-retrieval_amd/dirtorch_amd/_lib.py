"""ctypes binding of libdir_engine.so (C ABI in include/dir_engine.h).

There is no fallback: if the shared library is missing or a call fails, this raises.  The product
path never routes through PyTorch ops or the CPU oracle.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char, c_char_p, c_double, c_float, c_int, c_int64,
                    c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIRTORCH_AMD_LIB: an alternative build of the same library (kernel experiments); no other effect
LIB_PATH = os.environ.get('DIRTORCH_AMD_LIB') or os.path.join(_HERE, 'libdir_engine.so')

DIR_BF16, DIR_FP16, DIR_F32, DIR_FP16P = 0, 1, 2, 3
DIR_ERR_RANGE = -7      # dir_status: a finite fp32 weight does not fit the chosen 16-bit format
DTYPES = {'bf16': DIR_BF16, 'fp16': DIR_FP16, 'f32': DIR_F32, 'fp16p': DIR_FP16P}
DIR_IMG_F32_NCHW, DIR_IMG_U8_NHWC = 0, 1
DIR_POOL_GEM, DIR_POOL_MAX, DIR_POOL_AVG = 0, 1, 2
POOLING = {'gem': DIR_POOL_GEM, 'max': DIR_POOL_MAX, 'avg': DIR_POOL_AVG}
DIR_HEAD_RMAC, DIR_HEAD_FPN, DIR_HEAD_FPN0, DIR_HEAD_CLASSIFIER = 0, 1, 2, 3


class DirError(RuntimeError):
    """A dir_* call returned a negative status; carries the code and dir_last_error()."""

    def __init__(self, code, msg):
        super().__init__('dir_engine error %d: %s' % (code, msg))
        self.code = code


class ModelDesc(Structure):
    _fields_ = [('bottleneck', c_int), ('layers', c_int * 4), ('out_dim', c_int),
                ('norm_features', c_int), ('pooling', c_int), ('without_fc', c_int),
                ('center_bias', c_float), ('mean', c_float * 3), ('std', c_float * 3),
                ('head', c_int)]


class ProfRecord(Structure):
    _fields_ = [('name', c_char * 48), ('kernel', c_char * 48), ('flops', c_double),
                ('bytes', c_double), ('ms', c_float)]


# name -> (restype, argtypes); every symbol include/dir_engine.h declares
SIGNATURES = {
    'dir_last_error': (c_char_p, []),
    'dir_version': (c_char_p, []),
    'dir_engine_create': (c_int, [POINTER(ModelDesc), c_int, POINTER(c_void_p)]),
    'dir_engine_destroy': (c_int, [c_void_p]),
    'dir_engine_set_tensor': (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    'dir_engine_finalize': (c_int, [c_void_p, c_int]),
    'dir_engine_out_dim': (c_int, [c_void_p, POINTER(c_int)]),
    'dir_workspace_bytes': (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_size_t)]),
    'dir_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                            c_size_t, c_void_p]),
    'dir_forward_features': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p,
                                     c_size_t, c_void_p]),
    'dir_engine_autotune': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'dir_engine_tuning_export': (c_int, [c_void_p, c_char_p, c_size_t, POINTER(c_size_t)]),
    'dir_engine_tuning_import': (c_int, [c_void_p, c_char_p]),
    'dir_engine_set_profiling': (c_int, [c_void_p, c_int]),
    'dir_engine_profile_pause': (c_int, [c_void_p, c_int]),
    'dir_engine_get_profile': (c_int, [c_void_p, POINTER(ProfRecord), c_int, POINTER(c_int)]),
    'dir_conv_variant_count': (c_int, []),
    'dir_reload_env': (c_int, []),
    'dir_conv_variant_name': (c_int, [c_int, c_char_p, c_int]),
    'dir_conv_bn_act': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 14
                        + [c_void_p]),
    'dir_conv_bn_act_f32': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    'dir_conv_bn_act_pair': (c_int, [c_void_p] * 4 + [c_void_p] + [c_void_p] * 4 + [c_int] * 12 + [c_void_p]),
    'dir_conv_pair_dual': (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p]),
    'dir_prep_input_pair': (c_int, [c_void_p, c_int, POINTER(c_float), POINTER(c_float), c_void_p, c_void_p,
                                    c_int, c_int, c_int, c_void_p]),
    'dir_stem_pool_pair': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    'dir_stem_pool_u8': (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p]),
    'dir_engine_overflow': (c_int, [c_void_p, c_void_p, POINTER(c_int)]),
    'dir_conv_heuristic': (c_int, [c_int] * 12 + [c_char_p, c_int, POINTER(c_int)]),
    'dir_conv_variant_admissible': (c_int, [c_int] * 13 + [POINTER(c_int)]),
    'dir_conv_bn_act_splitk': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 15 +
                               [c_void_p, c_size_t, POINTER(c_int), c_void_p]),
    'dir_conv_c3c1': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
                      + [c_int] * 8 + [c_void_p]),
    'dir_conv_c3c1_ds': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
                         + [c_int] * 6 + [c_void_p]),
    'dir_conv_c3c1_wpair': (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_void_p]),
    'dir_conv_c3c1_ds_wpair': (c_int, [c_void_p] * 11 + [c_int] * 5 + [c_void_p]),
    'dir_conv_dual': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    'dir_conv_bn_act_naive': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
                              + [c_int] * 13 + [c_void_p]),
    'dir_prep_input': (c_int, [c_void_p, c_int, POINTER(c_float), POINTER(c_float), c_void_p,
                               c_int, c_int, c_int, c_int, c_void_p]),
    'dir_stem_pool': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                              c_int, c_void_p]),
    'dir_maxpool_3x3s2': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dir_global_pool': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                c_float, c_float, c_int, c_void_p]),
    'dir_resize_workspace_bytes': (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    'dir_resize_bilinear_u8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_size_t, c_void_p]),
    'dir_upsample_add': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p]),
    'dir_l2norm_rows': (c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    'dir_gemm_splitk_factor': (c_int, [c_int, c_int, c_int]),
    'dir_gemm_nt_f32': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dir_fc_l2': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'dir_pca_whiten_l2': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p]),
    'dir_pca_whiten_l2_unit': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p]),
    'dir_similarity': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dir_similarity_unit': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dir_rank_counts': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p]),
    'dir_revisitop_ap': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'dir_expand_descriptors': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    'dir_comm_init_all': (c_int, [c_int, POINTER(c_int), POINTER(c_void_p)]),
    'dir_comm_size': (c_int, [c_void_p, POINTER(c_int)]),
    'dir_comm_destroy': (c_int, [c_void_p]),
    'dir_allgather_desc': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_size_t, c_int,
                                   POINTER(c_void_p)]),
    'dir_multiscale_pool': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                    c_void_p]),
}

_lib = None


def load():
    """Load libdir_engine.so (once) and bind every declared symbol.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            'HIP extension %s is missing - build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or '
            'deep-image-retrieval_amd/csrc/build.sh; there is no CPU fallback.' % LIB_PATH)
    # torch first: its bundled HIP runtime (libamdhip64) must be the one this library binds to,
    # otherwise the process ends up with two runtimes and device pointers are not shared.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def reload_env():
    """The library reads its DIRTORCH_AMD_* A/B switches from the environment once; after changing one inside a running
    process (tests, A/B scripts) call this, then build a new engine.  No-op when the library has not been loaded yet."""
    if _lib is not None:
        _lib.dir_reload_env()


def check(code):
    if code != 0:
        raise DirError(code, load().dir_last_error().decode('utf-8', 'replace'))


def call(name, *args):
    """Call a status-returning entry point and raise DirError on failure."""
    check(getattr(load(), name)(*args))


def ptr(t):
    """Device (or host) address of a torch tensor / None as c_void_p."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
