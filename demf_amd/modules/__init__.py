from .coder import DeMFClassAgnosticBBoxCoder
from .detector import DeMFHotPath, DeMFVoteNet, head_kwargs
from .head import DeMFVoteHead
from .image_stream import ChannelMapper, DeformableDetrEncoder, ImageStream, ResNet50
from .pointnet2 import PointFPModule, PointNet2SASSG, PointSAModule, build_sa_module
from .transformer import (DeMFTransformerDecoderLayer, MultiScaleDeformableAttention,
                          PositionEmbeddingLearned)
from .vote import BaseConvBboxHead, VoteModule

__all__ = ["DeMFClassAgnosticBBoxCoder", "DeMFHotPath", "DeMFVoteNet", "head_kwargs", "DeMFVoteHead",
           "PointFPModule", "PointNet2SASSG", "PointSAModule", "build_sa_module",
           "DeMFTransformerDecoderLayer", "MultiScaleDeformableAttention",
           "PositionEmbeddingLearned", "BaseConvBboxHead", "VoteModule", "ImageStream",
           "ResNet50", "ChannelMapper", "DeformableDetrEncoder"]
