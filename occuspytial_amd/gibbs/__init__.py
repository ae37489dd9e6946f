from .logit import LogitICARGibbs, LogitRSRGibbs

__all__ = ('LogitICARGibbs', 'LogitRSRGibbs')
