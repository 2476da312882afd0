"""dirtorch_amd - MI355X-native drop-in for the descriptor-extraction + ranking path of
naver/deep-image-retrieval (`dirtorch`).

Same names as the reference on this path:
    dirtorch_amd.nets.create_model / model_names         (dirtorch/nets/__init__.py)
    dirtorch_amd.utils.common.pool / whiten_features / matmul / tonumpy / load_checkpoint ...
    dirtorch_amd.test_dir.extract_image_features / eval_model / load_model
    dirtorch_amd.extract_features.extract_features
All arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of include/dir_engine.h;
PyTorch only owns device memory, streams and torch.distributed.
"""
__version__ = '0.1'
