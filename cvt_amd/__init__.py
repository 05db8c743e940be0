"""cvt_amd -- MI355X-native OPQ encode + ADC search path of willard-yuan/cvt.

The product is the C-ABI shared library ``cvt_amd/lib/libcvtmi.so`` (include/cvtmi.h), built from
the hand-written HIP kernels in ``cvt_amd/csrc`` and wrapped by the C++ mirror of the reference
classes in ``cvt_amd/host``.  This Python package is only the thin ctypes binding the tests and
bench.py drive it through; torch is used for device memory, streams and torch.distributed.
"""
from .capi import (Comm, CvtmiError, FlatIndex, HnswIndex, OpqIndex, kmeans, lib, load_library, opq_learn_rotation, opq_train, pca_project, pinned_empty, set_tuning, sq8_decode, sq8_decode_faiss,  # noqa: F401
                   sq8_encode, sq8_train, search_sharded_all, shard_range, topk_merge, topk_select)

IP, L2F, L2U8 = 0, 1, 2
