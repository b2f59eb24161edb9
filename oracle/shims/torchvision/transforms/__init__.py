class ToTensor:
    pass
