# same export list as the reference's bubbleformer/layers/__init__.py:1-5 (U-Net conv layers are out of scope)
from .positional_encoding import ContinuousPositionBias1D, RelativePositionBias
from .linear_layers import GeluMLP, SirenMLP, FiLMMLP
from .patching import HMLPEmbed, HMLPDebed
from .attention import AxialAttentionBlock, AttentionBlock
