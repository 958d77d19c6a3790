from . import hrnet  # noqa: F401
