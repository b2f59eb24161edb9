"""Minimal stand-in for the `timm` package (own code, test infrastructure only).

The reference imports timm for DropPath / trunc_normal_ / registry decorators
(semseg/models/backbones/convnext_orig.py:14, vit_encoder.py). None of that
changes attack arithmetic; this shim only lets `/root/reference` import in the
build container so golden vectors can be generated from it.
"""
from . import optim, scheduler  # noqa: F401
