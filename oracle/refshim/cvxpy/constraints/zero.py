from . import Zero  # noqa: F401
