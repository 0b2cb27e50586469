"""strutopy_amd -- MI355X-native E-step for the Structural Topic Model (drop-in for strutopy's STM hot path)."""
__all__ = ["STM"]


def __getattr__(name):
    if name == "STM":
        from .stm import STM
        return STM
    raise AttributeError(name)
