def create_scheduler(*a, **k):
    raise NotImplementedError("timm shim: schedulers are outside the hot path")
