"""sar-ssl_amd: MI355X-native implementation of SAR-SSL's cross-channel-reconstruction pretraining path.

Layout: csrc/ (HIP kernels + C ABI, include/sarssl_hip.h), hip.py (ctypes tensor wrappers), engine.py (hand-written
forward/backward), runtime.py (precision modes, flat parameter store, fused Adam), dist.py (RCCL data parallel), and the
reference-named host modules model.py / learner.py / common/ / dataset.py / opt.py / run_pretrain.py.
"""
__all__ = ["hip", "engine", "runtime", "dist", "model", "learner", "dataset", "opt", "synth"]
