"""Building blocks of the plugin surface (`avssl.module.<name>`): encoders, loss, projections, retrieval, layer mix.
Pooling layers and SupConLoss of the reference are not on the SpeechCLIP path."""
from . import clip_official as _clip
from . import losses
from . import projections as _proj
from . import retrieval as _ret
from . import speech_encoder_plus as _enc
from . import weighted_sum as _ws

ClipModel = _clip.ClipModel
MaskedContrastiveLoss = losses.MaskedContrastiveLoss
mutualRetrieval = _ret.mutualRetrieval
FairseqSpeechEncoder_Hubert = _enc.FairseqSpeechEncoder_Hubert
S3prlSpeechEncoderPlus = _enc.S3prlSpeechEncoderPlus
WeightedSumLayer = _ws.WeightedSumLayer
MLPLayers = _proj.MLPLayers

__all__ = ["ClipModel", "MaskedContrastiveLoss", "mutualRetrieval", "FairseqSpeechEncoder_Hubert", "S3prlSpeechEncoderPlus", "WeightedSumLayer",
           "MLPLayers", "losses"]
