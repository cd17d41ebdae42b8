"""nf_distillation_amd: MI355X-native engine for the Glow coupling-flow hot path of
vklyukin/nf_distillation (forward / inverse / log_prob + the distillation training step).

Drop-in names (the reference's ``models/__init__.py:1-3``)::

    from nf_distillation_amd import (create_glow_model, inherit_permutation_matrix, FlowStep,
                                     gaussian_sample, SqueezeLayer)
"""
from .modules import (ActNorm1d, ActNorm2d, Conv2d, Conv2dZeros, FlowNet, FlowNetGetAllOutputs, FlowStep, Glow, MADEStep,
                      GlowGetAllOutputs, InvertibleConv1x1, LinearZeros, Permute2d, Split2d, SqueezeLayer,
                      create_glow_model, gaussian_sample, inherit_permutation_matrix, split_feature)
from .checkpoint import load_checkpoint
from .data import postprocess_images, preprocess_images

__all__ = ["create_glow_model", "inherit_permutation_matrix", "FlowStep", "gaussian_sample", "SqueezeLayer",
           "Glow", "GlowGetAllOutputs", "FlowNet", "FlowNetGetAllOutputs", "Split2d", "ActNorm1d", "ActNorm2d",
           "InvertibleConv1x1", "Conv2d", "Conv2dZeros", "LinearZeros", "Permute2d", "split_feature", "load_checkpoint", "preprocess_images", "postprocess_images"]
