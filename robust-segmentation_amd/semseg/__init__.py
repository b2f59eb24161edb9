"""semseg: MI355X-native drop-in for the attack hot path of nmndeep/Robust-Segmentation.

Same module names as the reference package (semseg.attacker, semseg.val, semseg.metrics,
semseg.losses, semseg.utils.utils, semseg.models) so that ``tools/infer.py``-style callers switch by
putting ``robust-segmentation_amd/`` on PYTHONPATH.  Device work goes through libsea_hip.so
(hand-written HIP for gfx950, see include/sea_hip.h); there is no CPU fallback.
"""
__all__ = ["attacker", "val", "metrics", "losses", "models", "utils"]
