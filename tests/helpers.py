"""Shared test helpers: the shipped configurations live in the package (speechclip_amd/util/shipped_configs.py); the embedding-parity
metric used by every end-to-end test lives here."""
import torch
import torch.nn.functional as F

from speechclip_amd.util.shipped_configs import make_config  # noqa: F401


def centred_cos(got: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """Per-row cosine between `got` and `ref` after subtracting the REFERENCE's batch mean from both.

    Why not the raw cosine: with random-init (or any) towers the embeddings of different utterances share a large common component
    (pairwise raw cosine 0.988-0.9986 in the fixtures), so `cos(got_b, ref_b) > 0.999` would also pass for the WRONG utterance's embedding.
    Removing the common component leaves what distinguishes the rows; rows of different utterances then have cosine around -1/(B-1)."""
    got, ref = got.detach().float().cpu().reshape(got.shape[0], -1), ref.detach().float().cpu().reshape(ref.shape[0], -1)
    mu = ref.mean(0, keepdim=True)
    return F.cosine_similarity(got - mu, ref - mu, dim=-1)


def assert_rows_match(got: torch.Tensor, ref: torch.Tensor, min_ccos: float, what: str = "embedding", max_wrong_ccos: float = 0.9):
    """The parity assertion of the end-to-end tests plus its negative control: every row matches its reference row in centred cosine, and
    the SAME metric rejects the output with its rows rotated by one (so the metric can tell utterances apart on this very batch)."""
    cc = centred_cos(got, ref)
    assert cc.min().item() >= min_ccos, (what, "centred cosine per row", cc.tolist())
    if got.shape[0] > 1:
        wrong = centred_cos(torch.roll(got.detach().float().cpu(), 1, dims=0), ref)
        assert wrong.max().item() < min(max_wrong_ccos, min_ccos), (what, "negative control: rotated rows still pass", wrong.tolist())
    return cc
