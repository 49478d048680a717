#!/usr/bin/env python3
"""LayerNorm backward / bf16 column-sum timing at the fine-tuning shape [81664, 768]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from speechclip_amd import ops
x=torch.randn(81664,768,device="cuda").to(torch.bfloat16); dy=torch.randn_like(x); g=torch.ones(768,device="cuda")
for f,name in ((lambda: ops.layernorm_bwd_bf16(x,dy,g), "ln_bwd"), (lambda: ops.colsum_bf16(x), "colsum")):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(name, "ms", (time.time()-t)/20*1e3)
