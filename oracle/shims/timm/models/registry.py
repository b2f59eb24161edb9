def register_model(fn):
    return fn
