"""demf_amd - MI355X-native (gfx950) implementation of the DeMF fusion hot path.

``demf_amd.ops``      operator API (mmdet3d.ops / mmcv.ops signatures) on libdemf_hip.so
``demf_amd.modules``  nn.Modules mirroring the reference's DeMFVoteHead /
                      DeMFTransformerDecoderLayer (+ the PointNet++ backbone they sit on)
"""
__version__ = "0.1.0"
