from .convnext_upernet import CONVNEXT_SETTINGS, ConvNeXt, UperNetForSemanticSegmentation  # noqa: F401
from .segmenter import MaskTransformer, SegMenter, VisionTransformer, create_segmenter  # noqa: F401

__all__ = ["UperNetForSemanticSegmentation", "ConvNeXt", "CONVNEXT_SETTINGS", "SegMenter", "VisionTransformer",
           "MaskTransformer", "create_segmenter"]
