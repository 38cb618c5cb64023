"""deepi2p_amd -- MI355X-native DeepI2P registration hot path (HIP kernels behind a C ABI).

Importing this package does not load the HIP library; the first operator call does, and raises
``DeepI2PHipError`` if it is missing.  There is no CPU fallback anywhere in this package.
"""
__all__ = ["ops", "torch_ops", "index_max", "ball_query", "FrustumRegistration", "networks", "registration", "pipeline", "synthetic"]
