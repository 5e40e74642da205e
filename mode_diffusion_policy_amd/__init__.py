"""MI355X-native MoDE denoising hot path (drop-in for the reference's MoDeDiT / GCDenoiser / sample_ddim)."""
from .modedit import MoDeDiT, NoiseBlockMoE  # noqa: F401
from .score_wrappers import GCDenoiser  # noqa: F401
from .gc_sampling import get_sigmas_exponential, sample_ddim  # noqa: F401
from .training import diffusion_loss, training_step  # noqa: F401
from .lang_buffer import AdvancedLangEmbeddingBuffer  # noqa: F401
from .perceptual_encoders import (FiLMResNet18Policy, FiLMResNet34Policy, FiLMResNet50Policy, GraphedVisualEncoder, ResNetEncoderWithFiLM,  # noqa: F401
                                  embed_visual_obs)

__all__ = ["MoDeDiT", "NoiseBlockMoE", "GCDenoiser", "sample_ddim", "get_sigmas_exponential", "diffusion_loss", "training_step", "AdvancedLangEmbeddingBuffer",
           "FiLMResNet18Policy", "FiLMResNet34Policy", "FiLMResNet50Policy", "ResNetEncoderWithFiLM", "embed_visual_obs"]
