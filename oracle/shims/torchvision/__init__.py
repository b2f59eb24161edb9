"""Stand-in for torchvision (own code): the reference only names transforms.ToTensor."""
from . import transforms  # noqa: F401
