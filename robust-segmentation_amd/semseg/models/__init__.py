from .convnext_upernet import CONVNEXT_SETTINGS, ConvNeXt, UperNetForSemanticSegmentation  # noqa: F401

__all__ = ["UperNetForSemanticSegmentation", "ConvNeXt", "CONVNEXT_SETTINGS"]
