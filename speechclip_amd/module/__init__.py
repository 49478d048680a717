from .clip_official import ClipModel
from .losses import MaskedContrastiveLoss
from .projections import *  # noqa: F401,F403
from .retrieval import mutualRetrieval
from .speech_encoder_plus import FairseqSpeechEncoder_Hubert, S3prlSpeechEncoderPlus
from .weighted_sum import WeightedSumLayer
from . import losses  # noqa: F401
