#!/usr/bin/env python3
"""tools/probes/pk_f32_probe.hip on a side stream beside a load on the main stream: do packed fp32 results ever differ from the single-instruction results?
usage: python tools/pk_f32_probe.py [load: none|front|conv0|gemm|speech] [rounds]"""
import ctypes
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from speechclip_amd import ops  # noqa: E402

load = sys.argv[1] if len(sys.argv) > 1 else "front"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
so = os.path.join(ROOT, "tools", "probes", "pk_f32_probe_bin.so")
if not os.path.exists(so):
    os.system("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared %s -o %s" % (os.path.join(ROOT, "tools", "probes", "pk_f32_probe.hip"), so))
L = ctypes.CDLL(so)
L.pk_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
model = bench.build_model().cuda().eval()
batch, lens = bench.make_batch(256, 160000, 0, "cuda")
hub = model.audio_encoder.encoder
counts = torch.zeros(16, dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
a_big = torch.randn(128000, 768, device="cuda").to(torch.bfloat16); w_big = torch.randn(3072, 768, device="cuda").to(torch.bfloat16)
o_big = torch.empty(128000, 3072, device="cuda", dtype=torch.bfloat16)
lens_l = [int(v) for v in batch["wav_len"].tolist()]
total = 0
for r in range(rounds):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for k in range(40):                       # 40 launches x 2048 blocks x 256 lanes x 512 iterations x 8 compared results
            L.pk_probe_launch(counts.data_ptr(), 2048, 512, r * 1000 + k, side.cuda_stream)
    total += 40 * 2048 * 256 * 512
    if load == "front":
        for _ in range(3):
            hub.extract_all_layers(batch["wav"], lens_l, stop_layer=0)
    elif load == "speech":
        with torch.no_grad():
            model.forward_audio(batch["wav"], batch["wav_len"])
    elif load == "gemm":
        for _ in range(20):
            ops.gemm(a_big, w_big, None, 1, out=o_big)
    cur.wait_stream(side)
    torch.cuda.synchronize()
c = counts.tolist()
print("load %s: %d lane-iterations; mismatches packed vs single: fma lo/hi %d/%d  mul lo/hi %d/%d  add lo/hi %d/%d  cross-half add lo/hi %d/%d  dependent-chain accumulators lo/hi %d/%d"
      % (load, total, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9]))
