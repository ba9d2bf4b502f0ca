"""mccnn_amd -- MI355X-native Monte-Carlo convolution layer (drop-in for MCCNN's MCConvBuilder path).

Only the hot path lives here: csrc/ (hand-written HIP kernels + the C-ABI of include/mccnn.h),
MCConvModule (the reference's Python op surface, tf_ops/MCConvModuleSrc), MCConvBuilder
(PointHierarchy / ConvolutionBuilder, utils/MCConvBuilder.py) and dist (per-cloud sharding +
RCCL all-reduce of the kernel-MLP weight gradients).
"""
__version__ = "0.1.0"
