from . import NonPos  # noqa: F401
