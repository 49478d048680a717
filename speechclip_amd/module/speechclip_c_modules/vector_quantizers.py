"""SimpleVectorQuantizer (avssl/module/speechclip_c_modules/my_vector_quantizer.py:12-165), hard (non-gumbel) mode on HIP:
special-token masking, arg-max one-hot, code/prob perplexities and per-keyword entropy in one pass over the
[B*K, V] score matrix (sc_vq_fwd).  `subword_prob` is returned lazily (a dense one-hot is only materialised if read)."""
import ast

import torch
import torch.nn as nn

from ... import ops

__all__ = ["SimpleVectorQuantizer"]


class _LazyOneHot(dict):
    """vq_results dict whose 'subword_prob' ([B,K,V] one-hot) is built on first access."""

    def __missing__(self, key):
        if key == "subword_prob":
            t = self["targets"]
            B, K, _ = t.shape
            v = torch.zeros(B, K, self["num_vars"], device=t.device, dtype=torch.float32).scatter_(-1, t, 1.0)
            self[key] = v
            return v
        raise KeyError(key)


class SimpleVectorQuantizer(nn.Module):
    def __init__(self, temp, groundTruthPerplexity=None, time_first=True, use_gumbel=False, hard=True):
        super().__init__()
        if use_gumbel or not hard or not time_first or groundTruthPerplexity is not None:
            raise NotImplementedError("MI355X path supports the shipped VQ settings: time_first, hard, no gumbel")
        self.time_first, self.use_gumbel, self.hard = time_first, use_gumbel, hard
        if isinstance(temp, str) and temp.startswith("learnable="):
            self.temp_type = "learnable"
            self.curr_temp = nn.parameter.Parameter(torch.FloatTensor([ast.literal_eval(temp.replace("learnable=", ""))]))
        elif isinstance(temp, str) and temp.startswith("fixed="):
            self.temp_type = "fixed"
            self.register_buffer("curr_temp", torch.FloatTensor([ast.literal_eval(temp.replace("fixed=", ""))]))
            self._fixed_temp = float(ast.literal_eval(temp.replace("fixed=", "")))   # host copy: reading the buffer would sync the stream every step
        else:      # "(max, min, decay)": scheduled (my_vector_quantizer.py:45-52); nothing in the reference calls set_num_updates, so it stays at max
            self.temp_type = "scheduled"
            t3 = ast.literal_eval(temp) if isinstance(temp, str) else tuple(temp)
            assert len(t3) == 3, f"{t3}, {len(t3)}"
            self.max_temp, self.min_temp, self.temp_decay = t3
            self.curr_temp = self.max_temp
        self.groundTruthPerplexity = None
        if self.temp_type == "fixed":   # a checkpoint may carry another value in the buffer: refresh the host copy once, at load time
            self.register_load_state_dict_post_hook(lambda mod, keys: setattr(mod, "_fixed_temp", float(mod.curr_temp.detach().cpu().item())))

    def set_num_updates(self, num_updates):
        if self.temp_type == "scheduled":       # my_vector_quantizer.py:58-62
            self.curr_temp = max(self.max_temp * self.temp_decay ** num_updates, self.min_temp)

    def temperature_value(self) -> float:
        if self.temp_type == "fixed":
            return self._fixed_temp
        return float(self.curr_temp) if self.temp_type == "scheduled" else float(self.curr_temp.item())

    def forward(self, x, prob_msk=[0, 2, 3], produce_targets=True):
        # train mode: same statistics and hard targets; the straight-through gradient (softmax(x / temp), :133-141) is applied where the
        # sub-word embeddings are formed (train_tail.KeywordSTFn via KW_CascadedBranch), `subword_prob` stays the hard one-hot value.
        # (a learnable temperature, `temp: "learnable=..."`, gets its gradient there too: d loss / d T = -(1 / T) sum dcos . cos)
        B, K, V = x.shape
        targets, stats, ent = ops.vq_fwd(x.reshape(B * K, V), K, prob_msk)
        res = _LazyOneHot()
        res["num_vars"] = V
        res["code_perplexity"] = stats[0]
        res["prob_perplexity"] = stats[1]
        res["ent_per_t"] = ent
        res["temp"] = self.temperature_value()
        res["diversity_loss"] = (V - stats[1]) / V
        res["targets"] = targets.view(B, K, 1)
        return res

    @staticmethod
    def embed(vq_results, emb_weight):
        """subword_prob @ E for a hard one-hot = gather of the chosen rows (kwClip.py:909)."""
        t = vq_results["targets"]
        B, K, _ = t.shape
        return ops.gather_rows(emb_weight, t.reshape(-1)).view(B, K, emb_weight.shape[1])
