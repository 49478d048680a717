#!/usr/bin/env python3
"""bench.py -- speech-image pairs/s of the Parallel SpeechCLIP base forward + InfoNCE step on MI355X.

Workload (BASELINE.json configs[1] / SURVEY.md section 8d "C2"): HuBERT-base + CLIP ViT-B/32 + parallel CLS head, bf16 MFMA
compute, per-GPU batch 256, every wave exactly 160000 samples (10 s @ 16 kHz => T = 499), 224^2 images, unique ids,
random-init weights (no network), synthetic inputs resident in HBM before the timed region.  One "step" = the full
forward of both towers + head + L2 norms + (N > 1: RCCL all-gather of the embeddings) + masked InfoNCE on the global batch.
N > 1: one process per GPU (torchrun), weak scaling (256 pairs per GPU), value = all pairs / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  "roofline":     dominant kernel (the bf16 MFMA GEMM): algorithmic FLOPs of all its launches / their HIP-event time
  "cpu_baseline": the fp32 CPU oracle (oracle/, a port of the reference's CPU path) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md


def conv_lens(L):
    out = []
    for k, s in [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2:
        L = (L - k) // s + 1
        out.append(L)
    return out


def algorithmic_gflop_per_pair(L=160000, d=768, ffn=3072, layers=12, vit_w=768, vit_layers=12, patch=32, res=224, E=512):
    """SURVEY.md section 8(d) formulas (2*MACs of the dense contractions; softmax/LN/GELU excluded)."""
    Ts = conv_lens(L)
    T = Ts[-1]
    cnn0 = 2 * Ts[0] * 512 * 1 * 10
    cnn = sum(2 * Ts[i] * 512 * 512 * k for i, k in zip(range(1, 7), [3, 3, 3, 3, 2, 2]))
    proj = 2 * T * 512 * d
    pos = 2 * T * d * (d // 16) * 128
    lin = layers * 2 * T * (4 * d * d + 2 * d * ffn)
    att = layers * 4 * T * T * d
    n = (res // patch) ** 2 + 1
    vit_lin = 2 * (n - 1) * 3 * patch * patch * vit_w + vit_layers * 2 * n * (4 * vit_w * vit_w + 8 * vit_w * vit_w) + 2 * vit_w * E
    vit_att = vit_layers * 4 * n * n * vit_w
    branch_lin = 2 * (T + 1) * d * 2 * d + 2 * (d * d + d * d + 2 * d * ffn) + 2 * d * E   # K/V of all frames + CLS-row-only rest
    branch_att = 4 * (T + 1) * d
    total = cnn + proj + pos + lin + vit_lin + branch_lin + cnn0 + att + vit_att + branch_att     # 158.1 GF
    # FLOPs that actually run on the MFMA GEMM kernel (roofline numerator).  The pooling head is evaluated in its algebraic form
    # (score = x.u_r, value projection of 8 pooled vectors), so its K/V GEMM over all frames (1.18 GF/pair) is NOT executed and
    # is not credited to the kernel: 8 score columns + per-head value projection + out_proj + FFN + final projection.
    branch_gemm = 2 * T * d * 8 + 2 * d * d + 2 * (d * d + 2 * d * ffn) + 2 * d * E
    # The grouped positional conv (`pos`, 4.7 GF/pair) runs in its own windowed MFMA kernel (csrc/posconv.hip), not in the GEMM kernel: not credited.
    gemm = cnn + proj + lin + vit_lin + branch_gemm
    return total / 1e9, gemm / 1e9


LARGE = dict(d=1024, ffn=4096, layers=24, vit_w=1024, vit_layers=24, patch=14, E=768)   # HuBERT-large + ViT-L/14 (BASELINE configs[4])


def build_model(seed=7122, large=False, cascaded=False):
    from speechclip_amd.util.shipped_configs import make_config
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(seed)
    if cascaded:      # C-base (BASELINE configs[2]): 8 keyword queries -> BatchNorm -> VQ over the 49408 sub-words -> CLIP text tower
        cfg = make_config(parallel=False, cascaded=True)
    elif large:
        cfg = make_config(d_model=1024, branch_heads=8, hubert_name="hubert_large_ll60k", clip_name="ViT-L/14", normalize_hiddenstates=True,
                          temperature_trainable=True)
    else:
        cfg = make_config()
    return KWClip_GeneralTransformer(cfg).eval()


def cpu_baseline(model_sd, n_pairs, L):
    """fp32 CPU oracle (port of the reference's CPU path) on n_pairs of the same workload, all host cores."""
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from oracle.speechclip_ref import SpeechClipRef
    ncpu = os.cpu_count() or 1
    ref = SpeechClipRef(HubertRefConfig.base(), ClipRefConfig.vit_b32(), parallel=True, branch_heads=8).eval()
    ref.encoder.load_state_dict({k[len("audio_encoder.encoder."):]: v for k, v in model_sd.items() if k.startswith("audio_encoder.encoder.")})
    ref.clip.load_state_dict({k[len("clip.model."):]: v for k, v in model_sd.items() if k.startswith("clip.model.")})
    ref.parallel_branch.load_state_dict({k[len("parallel_branch."):]: v for k, v in model_sd.items() if k.startswith("parallel_branch.")})
    g = torch.Generator().manual_seed(7122)

    def mk(b):
        return {"wav": 0.1 * torch.randn(b, L, generator=g), "wav_len": torch.full((b,), L), "image": torch.randn(b, 3, 224, 224, generator=g),
                "id": torch.arange(b)}
    # torch's CPU kernels do not scale to hundreds of threads on this workload (measured on the 2 x 64-core box: 16 threads beat
    # 128 by 3.5x): pick the best of a few thread counts on a 2-pair probe, report the count actually used as `cores`.
    best, cores = 0.0, 1
    with torch.no_grad():
        for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(n)
            ref(mk(1))
            t0 = time.perf_counter()
            ref(mk(2))
            r = 2 / (time.perf_counter() - t0)
            if r > best:
                best, cores = r, n
    torch.set_num_threads(cores)
    with torch.no_grad():
        ref.compute_loss(ref(mk(1)))                         # warm-up
        t0 = time.perf_counter()
        o = ref(mk(n_pairs))
        loss = ref.compute_loss(o)["loss"].item()
        dt = time.perf_counter() - t0
    return {"value": round(n_pairs / dt, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n_pairs} pairs (10 s audio + 224^2 image) through oracle/speechclip_ref.py fp32 with {cores} threads "
                      f"(best of 8/16/32/64 on a probe; box has {ncpu} hw threads), {dt:.1f} s, loss {loss:.4f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="pairs per GPU (default 256; 64 with --model large)")
    ap.add_argument("--model", choices=["base", "large", "cascaded"], default="base", help="base = the headline workload (BASELINE configs[1]); large = "
                    "HuBERT-large + ViT-L/14 (configs[4]), informational")
    ap.add_argument("--audio-len", type=int, default=160000)
    ap.add_argument("--cpu-pairs", type=int, default=32, help="pairs for the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--train", action="store_true", help="time the TRAINING step of the trainable tail instead (forward in train mode + loss.backward() "
                    "+ grad all-reduce + clip + Adam + LR schedule); not the headline metric, reported with config.mode = 'train'")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"

    from speechclip_amd import ops, parallel
    large = args.model == "large"
    if args.batch is None:
        args.batch = 64 if large else 256
    casc = args.model == "cascaded"
    model = build_model(large=large, cascaded=casc)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if (rank == 0 and args.cpu_pairs > 0 and world == 1 and not args.train and not large and not casc) else None
    model = model.to(dev)
    B, L = args.batch, args.audio_len
    g = torch.Generator(device="cpu").manual_seed(7122 + rank)
    wav = (0.1 * torch.randn(B, L, generator=g)).to(dev)
    batch = {"wav": wav, "wav_len": torch.full((B,), L, dtype=torch.long), "image": torch.randn(B, 3, 224, 224, generator=g).to(dev),
             "id": (torch.arange(B) + rank * B).to(dev)}

    if args.train:
        model.train()
        (opt,), (sch,) = model.configure_optimizers()

        def step():
            opt.zero_grad()
            loss = model.training_step_end(model.training_step(batch, 0))["loss"]
            loss.backward()
            opt.step()
            sch["scheduler"].step()
            return loss.detach()
    else:
        def step():
            with torch.no_grad():
                lf, _, _ = model(batch)
                return model.compute_loss(parallel.gather_loss_feats(lf))["loss"]

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    # roofline instrumentation: HIP events around every GEMM launch of every THIRD timed step (two events per launch, ~220 launches per
    # step: on every step they cost ~1 % of the step time they are meant to explain)
    prof = None if args.no_roofline_events else []
    ops.PROFILE_SIDE = []
    ev_steps = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        instrumented = prof is not None and (i % 3 == 2 or args.steps < 3)
        ops.PROFILE = prof if instrumented else None
        ev_steps += int(instrumented)
        loss = step()
    ops.PROFILE = None
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    pairs_per_s = world * B * args.steps / dt
    # train mode crops every utterance to audio_encoder.max_audio_len (102400 samples, T = 319) exactly as the reference trains
    mal = int(getattr(model.audio_encoder, "max_audio_len", -1))
    L_eff = min(L, mal) if (args.train and mal > 0) else L
    total_gf, gemm_gf = algorithmic_gflop_per_pair(L_eff, **(LARGE if large else {}))
    if rank == 0:
        roof = None
        if prof:
            ms = sum(e[0].elapsed_time(e[1]) for e in prof)
            launches = len(prof)
            alg = gemm_gf * 1e9 * B * ev_steps                 # algorithmic GEMM FLOPs of this rank's instrumented launches
            ach = alg / (ms * 1e-3) / 1e12
            # HBM traffic of the kernel cannot be measured from inside this process (PMC counters need rocprofv3 and their own passes):
            # report the committed per-launch figure of the same command, with its source, or null if it is not there / not this workload
            traffic, tsrc = None, None
            tf = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic.json")
            if os.path.exists(tf) and not args.train and not large and not casc and B == 256 and L == 160000:
                tj = json.load(open(tf))
                traffic, tsrc = tj["hand_written_main_stream"]["bytes_per_launch"], tj["source"] + " (separate rocprofv3 --pmc passes, FETCH_SIZE x2; gemm256_kernel + gemm_bf16_kernel launches)"
            # the entry serves two kernels (vendor_gemm.hip): the hand-written gemm256_kernel family (everything fused / overlapping rows /
            # small) and hipBLASLt (plain GEMMs).  The DOMINANT kernel of the step is the hand-written one: `achieved` is ITS flops / ITS time;
            # the whole entry and the library part are reported beside it.
            exe = sum(e[2] for e in prof)
            # The image tower runs on a side stream beside the speech tower (its small kernels are slotted in between the speech tower's
            # for most of the step).  The roofline counts the launches of the MAIN stream -- the speech tower, 97 % of the GEMM flops --,
            # whose event time is the kernel's own apart from the few CUs a side-stream kernel holds at that moment; the side stream's
            # launches wait for CUs inside their event window (stretched, overlapping in wall time) and are reported as `side_stream`.
            main_stream = torch.cuda.current_stream().cuda_stream

            def concurrent(e):
                return e[4] != main_stream
            flags_c = [concurrent(e) for e in prof]
            part = {}
            for path, name in ((0, "hand_written"), (1, "vendor")):
                sel = [e for e, c in zip(prof, flags_c) if e[3][6] == path and not c]
                pms = sum(e[0].elapsed_time(e[1]) for e in sel)
                pfl = sum(e[2] for e in sel) * (alg / exe)     # algorithmic share (executed flops include <0.2 % tile padding)
                part[name] = {"launches_per_step": len(sel) // ev_steps, "ms_per_step": round(pms / ev_steps, 3),
                              "avg_launch_ms": round(pms / max(1, len(sel)), 4), "achieved": round(pfl / max(pms, 1e-9) / 1e9, 1) if sel else None}
            tot_sel = [e for e, c in zip(prof, flags_c) if not c]
            tot_ms = sum(e[0].elapsed_time(e[1]) for e in tot_sel)
            tot_fl = sum(e[2] for e in tot_sel) * (alg / exe)
            csel = [e for e, c in zip(prof, flags_c) if c]
            cms = sum(e[0].elapsed_time(e[1]) for e in csel)
            conc = {"launches_per_step": len(csel) // ev_steps, "event_ms_per_step": round(cms / ev_steps, 3),
                    "windows_ms_per_step": round(sum(w0.elapsed_time(w1) for w0, w1 in ops.PROFILE_SIDE) / ev_steps, 3),
                    "note": "image-tower GEMMs on the side stream: they wait for CUs inside their event window (stretched, overlapping the main stream in "
                            "wall time); not counted in achieved.  SC_OVERLAP_VIT=0 serialises the towers"} if csel else None
            hw = part["hand_written"]
            roof = {"bound": "mfma", "kernel": "gemm256_kernel family (hand-written HIP: fused-GELU / QuickGELU epilogues, conv-as-GEMM with overlapping rows, small shapes)",
                    "achieved": hw["achieved"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(hw["achieved"] / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": tsrc, "launches_per_step": hw["launches_per_step"], "avg_launch_ms": hw["avg_launch_ms"],
                    "ms_per_step": hw["ms_per_step"],
                    "vendor_plain_gemms": dict(part["vendor"], kernel="hipBLASLt (plain QKV / out-proj / fc2 / ViT projections behind the same sc_gemm_bf16 entry)",
                                               frac=round(part["vendor"]["achieved"] / PEAK_BF16_TFLOPS, 4) if part["vendor"]["achieved"] else None),
                    "gemm_entry_total": {"achieved": round(tot_fl / max(tot_ms, 1e-9) / 1e9, 1), "frac": round(tot_fl / max(tot_ms, 1e-9) / 1e9 / PEAK_BF16_TFLOPS, 4),
                                         "launches_per_step": len(tot_sel) // ev_steps, "avg_launch_ms": round(tot_ms / max(1, len(tot_sel)), 4),
                                         "ms_per_step": round(tot_ms / ev_steps, 3)},
                    "side_stream": conc,
                    "gemm_ms_per_step": round(tot_ms / ev_steps, 3),
                    "instrumented_steps": ev_steps,
                    "executed_over_algorithmic": round(exe / alg, 4)}
        out = {"metric": "speech-image pairs/sec/node (%s)" % ("Cascaded SpeechCLIP base" if casc else "Parallel SpeechCLIP %s" % args.model), "value": round(pairs_per_s, 2), "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": ("Cascaded SpeechCLIP base (HuBERT-base + ViT-B/32 + CLIP text tower; flop model = the encoders' GEMMs, the keyword head adds < 1 %)" if casc else
                                       "Parallel SpeechCLIP large (HuBERT-large + ViT-L/14)" if large else "Parallel SpeechCLIP base (HuBERT-base + ViT-B/32)")
                          + " forward + InfoNCE, 10 s/16 kHz audio + 224^2 images",
                          "pairs_per_gpu": B, "global_batch": world * B, "audio_samples": L_eff, "frames": conv_lens(L_eff)[-1],
                          "parallelism": f"dp{world}" if world > 1 else "single", "weights": "random-init (no network)",
                          "algorithmic_gflop_per_pair": round(total_gf, 2), "mode": "train (tail: branch + layer-mix weights)" if args.train else "forward + loss"},
               "e2e_tflops_per_gpu": round(total_gf * 1e9 * pairs_per_s / world / 1e12, 1),
               "e2e_frac_of_bf16_peak": round(total_gf * 1e9 * pairs_per_s / world / 1e12 / PEAK_BF16_TFLOPS, 4),
               "loss": round(float(loss), 5), "roofline": roof, "cpu_baseline": None}
        if sd_cpu is not None:
            del model
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(sd_cpu, args.cpu_pairs, L)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
