"""orca_amd: MI355X-native (gfx950) inference engine for the Orca
sequence-to-3D-genome models.  The hot path (Encoder -> Encoder2/Encoder3 ->
Decoder cascade) runs in hand-written HIP kernels behind a C-ABI shared
library (include/orca_hip.h); this package is the thin Python host that keeps
the reference's module / model / predict API.
"""
__version__ = "0.1.0"
