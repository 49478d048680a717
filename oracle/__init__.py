"""CPU oracle for the SpeechCLIP forward/contrastive hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``speechclip_amd/`` (the product) may
import this package.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker /
the timed CPU baseline, never as the thing shipped.

What it is: a plain fp32 PyTorch-on-CPU restatement of the arithmetic the
reference executes on the path named by BASELINE.json:north_star.  The
reference (atosystem/SpeechCLIP) implements only the glue in Python; the
arithmetic lives in third-party packages that are NOT vendored under
/root/reference and are not installed here:

  * fairseq @ b5a039c292facba9c73f59ff34621ec131d82341 (requirements.txt:6):
    HubertModel / ConvFeatureExtractionModel / TransformerEncoder /
    TransformerSentenceEncoderLayer / MultiheadAttention  -> oracle/hubert_ref.py
  * openai/CLIP, unpinned HEAD (requirements.txt:4): VisionTransformer /
    Transformer / ResidualAttentionBlock / QuickGELU       -> oracle/clip_ref.py
  * the reference's own glue (avssl/model/kwClip.py, avssl/module/*.py)
                                                           -> oracle/speechclip_ref.py

Pinning: the reference's own tests hold no golden vector
for this path (they are stale and need network weights).  The oracle is pinned by
  (i)  outputs of the reference's own glue code, imported in the build container
       with stubs for the absent third-party packages (tests/golden/make_golden.py,
       fixtures under tests/golden/*.npz), including the two known-answer loss
       values recorded in BASELINE.md, and
  (ii) an independent cross-check of the third-party restatements against the
       architecture-equivalent `transformers` implementations (HubertModel,
       CLIPVisionModelWithProjection, CLIPTextModelWithProjection) with weights
       copied by key mapping (tests/test_oracle_vs_hf.py).
"""
