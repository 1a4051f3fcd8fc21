"""MI355X-native hot path of pix2pixHD audio super-resolution.

Mirrors the reference's module surface for that path (``models.mdct``, ``models.networks``,
``models.pix2pixHD_model``, ``models.models``, ``util.util``) over hand-written HIP kernels
reached through the C ABI of ``libp2phd_hip.so`` (include/p2phd.h).
"""
__version__ = "0.1.0"
