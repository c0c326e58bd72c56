"""nn.Module surface mirroring `mop.models` of the reference (names as in mop/models/__init__.py)."""
from .attention_variants import (BaselineMSA, CrossViewMixerMSA, EdgewiseGateHead,  # noqa: F401
                                 EdgewiseMSA, MultiHopMSA, UnifiedMSA)
from .components import (MLP, MSA, Block, DropPath, FuseExcInh, Kernels3, PatchEmbed,  # noqa: F401
                         ViewsLinear, ViTEncoder)
from .quartet_attn_patch import CausalSelfAttention, TransformerConfig  # noqa: F401
from .vit_mop import ViT_MoP  # noqa: F401
from .whisper_mop import (EncoderBlock, FuseExcInh2D, Kernels2D, MoP2D, MultiheadSelfAttention,  # noqa: F401
                          ViewsConv2D, WhisperConfig)
from .vit_edgewise import BlockEdgewise, ViTEdgewise  # noqa: F401
