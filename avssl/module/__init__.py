from speechclip_amd.module import *  # noqa: F401,F403
from speechclip_amd.module import (ClipModel, FairseqSpeechEncoder_Hubert, MaskedContrastiveLoss, MLPLayers,  # noqa: F401
                                   S3prlSpeechEncoderPlus, WeightedSumLayer, losses, mutualRetrieval)
