from .dictionary import Dictionary  # noqa: F401
