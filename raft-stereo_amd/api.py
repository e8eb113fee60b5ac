"""Public names of the package."""
from raft_stereo_amd.corr import (CorrBlock1DHip, AltCorrBlock1DHip, CorrSamplerHip)
from raft_stereo_amd import sampler as corr_sampler_hip
from raft_stereo_amd.update import (ConvGRUHip, bias_act_, conv_bias_act, gru_gate_r, gru_update, upsample_flow_hip, sequence_loss_hip,
                                    pool2x_hip, interp_hip)
from raft_stereo_amd.update_block import BasicMultiUpdateBlockHip, MotionEncoderHip
from raft_stereo_amd.refine import refine_disparity
from raft_stereo_amd.dispatch import (CORR_IMPLEMENTATIONS, HIP_CORR_CHOICES, make_corr_fn,
                                      wants_autocast, shard_rows, shard_pairs, global_batch_loss)

__all__ = ["CorrBlock1DHip", "AltCorrBlock1DHip", "CorrSamplerHip",
           "ConvGRUHip", "bias_act_", "conv_bias_act", "gru_gate_r", "gru_update", "upsample_flow_hip", "sequence_loss_hip", "pool2x_hip", "interp_hip", "BasicMultiUpdateBlockHip", "MotionEncoderHip", "refine_disparity",
           "corr_sampler_hip", "CORR_IMPLEMENTATIONS", "HIP_CORR_CHOICES", "make_corr_fn",
           "wants_autocast", "shard_rows", "shard_pairs", "global_batch_loss"]
