#!/usr/bin/env python3
"""BASELINE configs[4] tail timing: (B,4,128,64) latents -> VQ -> VQGAN decoder -> ISTFT+ / iSTFT audio on one MI355X.
    python tools/tail_bench.py --batch 64 --dtype bf16"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusynth_amd.synth import synth_input  # noqa: E402
from diffusynth_amd.vocoder import latents_to_audio  # noqa: E402
from diffusynth_amd.vqgan import PRODUCTION_CONFIG, VQGAN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--ops", type=int, default=0, help="1: per-op timeline of the decoder plan (events after every op)")
    a = ap.parse_args()
    torch.manual_seed(0)
    vae = VQGAN(**PRODUCTION_CONFIG).cuda()
    vae._decoder.set_compute_dtype(a.dtype)
    z = synth_input("tail_bench_z", (a.batch, 4, 128, 64)).cuda()

    def run():
        q = vae._vq_vae(z)[0]
        return latents_to_audio(vae._decoder, q)

    audio = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        audio = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    assert torch.isfinite(audio).all()
    if a.ops:
        for pl in vae._decoder._engine.plans.values():
            pl.prof = []
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            torch.cuda.synchronize()
            print(f"  before the decoder plan (VQ search, layout) {e0.elapsed_time(pl.prof_start) * 1e3:8.1f} us")
            prev = pl.prof_start
            for k, name, ev in pl.prof:
                meta = pl.conv_meta.get(k)
                print(f"  op{k:3d} {name:22s} {prev.elapsed_time(ev) * 1e3:8.1f} us  {meta[2] if meta else ''}")
                prev = ev
            pl.prof = None
    print(f"tail B={a.batch} {a.dtype}: {ms:.2f} ms per batch -> audio {tuple(audio.shape)}; {a.batch / ms * 1e3:.0f} clips/s")


if __name__ == "__main__":
    main()
