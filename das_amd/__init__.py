"""das_amd — MI355X-native (gfx950) implementation of the DAS data-parallel hot path.

Host side mirrors the reference's registry/config/method-name protocol (mmdet3d fork);
device side is libdas_hip.so (das_amd/csrc, C ABI in include/das_hip.h).
"""
__version__ = '0.1.0'
