from .ffmlp import FFMLP, ffmlp_forward
