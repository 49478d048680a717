#!/usr/bin/env python3
"""Soak test: 300 forward steps + 150 training steps at the bench shape; step time per 50-step window and memory growth."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, bench
from speechclip_amd import parallel
model = bench.build_model().cuda()
B,L=256,160000
g=torch.Generator().manual_seed(1)
batch={"wav":(0.1*torch.randn(B,L,generator=g)).cuda(),"wav_len":torch.full((B,),L),"image":torch.randn(B,3,224,224,generator=g).cuda(),"id":torch.arange(B).cuda()}
def fwd():
    with torch.no_grad():
        lf,_,_=model(batch); return model.compute_loss(parallel.gather_loss_feats(lf))["loss"]
for _ in range(3): l0=fwd()
torch.cuda.synchronize(); m0=torch.cuda.memory_allocated(); r0=torch.cuda.memory_reserved()
t=time.perf_counter(); times=[]
for i in range(300):
    l=fwd()
    if i%50==49:
        torch.cuda.synchronize(); t1=time.perf_counter(); times.append((t1-t)/50*1e3); t=t1
print("fwd ms/step per 50-step window:", [round(x,2) for x in times], "loss drift", float(l-l0), "alloc delta MB", (torch.cuda.memory_allocated()-m0)/1e6, "reserved delta MB", (torch.cuda.memory_reserved()-r0)/1e6)
model.train(); (opt,),(sch,)=model.configure_optimizers()
def tr():
    opt.zero_grad(); loss=model.training_step_end(model.training_step(batch,0))["loss"]; loss.backward(); opt.step(); sch["scheduler"].step(); return loss.detach()
for _ in range(3): tr()
torch.cuda.synchronize(); m0=torch.cuda.memory_allocated(); r0=torch.cuda.memory_reserved(); t=time.perf_counter(); times=[]; ls=[]
for i in range(150):
    l=tr()
    if i%50==49:
        torch.cuda.synchronize(); t1=time.perf_counter(); times.append((t1-t)/50*1e3); t=t1; ls.append(float(l))
print("train ms/step per 50-step window:", [round(x,2) for x in times], "losses", [round(x,4) for x in ls], "alloc delta MB", (torch.cuda.memory_allocated()-m0)/1e6, "reserved delta MB", (torch.cuda.memory_reserved()-r0)/1e6)
# ---- full-encoder fine-tuning (audio_encoder.trainable: true, with the train-mode dropouts): 40 steps at B = 64
del model, opt
torch.cuda.empty_cache()
ft = bench.build_model(finetune_all=True).cuda().train()
Bf = 64
fb = {k: (v[:Bf] if torch.is_tensor(v) else v) for k, v in batch.items()}
(opt,), (sch,) = ft.configure_optimizers()
def trf():
    opt.zero_grad(); loss = ft.training_step_end(ft.training_step(fb, 0))["loss"]; loss.backward(); opt.step(); sch["scheduler"].step(); return loss.detach()
for _ in range(2): trf()
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved(); t = time.perf_counter(); times = []; ls = []
for i in range(40):
    l = trf()
    if i % 20 == 19:
        torch.cuda.synchronize(); t1 = time.perf_counter(); times.append((t1 - t) / 20 * 1e3); t = t1; ls.append(float(l))
print("full-encoder train ms/step per 20-step window (B=64):", [round(x, 2) for x in times], "losses", [round(x, 4) for x in ls], "alloc delta MB",
      (torch.cuda.memory_allocated() - m0) / 1e6, "reserved delta MB", (torch.cuda.memory_reserved() - r0) / 1e6)
