from .logit import LogitICARGibbs

__all__ = ('LogitICARGibbs',)
