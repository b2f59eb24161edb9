def load_pretrained(*a, **k):
    raise NotImplementedError("timm shim")


def load_custom_pretrained(*a, **k):
    raise NotImplementedError("timm shim")
