"""mutualRetrieval -- same signature and return triple as avssl/module/retrieval.py:6-121 (recall@K in percent,
A->B, B->A and their mean).  Implemented with a rank test instead of a full argsort + Python row loops:
item i is a hit@K iff fewer than K candidates score above its best candidate carrying its answer id (sc_retrieval_ranks on the device)."""
from typing import Sequence, Tuple

import torch


def _recall_one_way(score: torch.Tensor, own_ids: torch.Tensor, cand_ids: torch.Tensor, recall_at: Sequence[int]) -> dict:
    n, m = score.shape
    # There is ONE implementation: sc_retrieval_ranks on the MI355X.  Host score matrices (the reference's validation_epoch_end hands CPU tensors
    # to mutualRetrieval, kwClip.py:487-491) are moved to the device; without a GPU / the library this raises -- no host re-implementation.
    from .. import ops
    from .._lib import SpeechClipHipError
    if not score.is_cuda:
        if not torch.cuda.is_available():
            raise SpeechClipHipError("mutualRetrieval ranks on the MI355X (sc_retrieval_ranks); no GPU is visible and there is no host fallback")
        # the device the answer ids live on if they are device tensors (a multi-rank run's model device), else this process's CURRENT device --
        # never a bare .cuda() (device 0 of whatever is visible)
        dev = next((t.device for t in (own_ids, cand_ids) if torch.is_tensor(t) and t.is_cuda), torch.device("cuda", torch.cuda.current_device()))
        score = score.to(dev)
    rank = ops.retrieval_ranks(score.float().contiguous(), own_ids, cand_ids)
    out = {}
    for k in recall_at:
        if k > m:
            print("recall@{} is not eligible for #{} samples".format(k, m))
        out["recall@{}".format(k)] = (rank < min(k, m)).float().mean().item() * 100
    return out


def mutualRetrieval(score_per_A: torch.Tensor, score_per_B: torch.Tensor, AB_answers: torch.Tensor, BA_answers: torch.Tensor,
                    recall_at: list, modality_A_title: str = "audio", modality_B_title: str = "image") -> Tuple[dict, dict, dict]:
    assert score_per_A.dim() == 2 and score_per_B.dim() == 2 and AB_answers.dim() == 1 and BA_answers.dim() == 1
    assert score_per_A.shape == (len(AB_answers), len(BA_answers)), (score_per_A.shape, (len(AB_answers), len(BA_answers)))
    assert score_per_B.shape == (len(BA_answers), len(AB_answers)), (score_per_B.shape, (len(BA_answers), len(AB_answers)))
    ab = _recall_one_way(score_per_A, AB_answers, BA_answers, recall_at)
    ba = _recall_one_way(score_per_B, BA_answers, AB_answers, recall_at)
    mean = {k: (ab[k] + ba[k]) / 2.0 for k in ab}
    return ab, ba, mean
