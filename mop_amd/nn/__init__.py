"""nn.Module surface mirroring `mop.models` of the reference (names as in mop/models/__init__.py)."""
from .attention_variants import (BaselineMSA, CrossViewMixerMSA, EdgewiseGateHead,  # noqa: F401
                                 EdgewiseMSA, MultiHopMSA, UnifiedMSA)
