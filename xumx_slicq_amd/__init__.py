"""MI355X-native demix hot path of xumx-sliCQ-V2 (sliCQT -> CDAE -> phasemix |
Wiener-EM -> isliCQT) behind the reference's Separator / Unmix / NSGT_SL /
INSGT_SL module API.  The arithmetic lives in ``csrc/`` (hand-written HIP for
gfx950 behind a C ABI, ``include/xumx_slicq_hip.h``); this package holds the
host-side plan builder and the ``nn.Module`` mirrors of the reference surface.
"""
__version__ = "0.1.0"
