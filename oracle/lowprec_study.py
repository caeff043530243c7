"""ORACLE-SIDE STUDY -- TEST INFRASTRUCTURE ONLY (never imported by m3pc_amd).

What would block-scaled FP8 operands (the `v_mfma_scale_f32_32x32x64_f8f6f4` path of gfx950: OCP e4m3 elements with one E8M0
power-of-two scale per 32 consecutive K elements, 2x the bf16 MFMA rate) do to the certified re-score of a plan step?
BASELINE.md's arithmetic says north_star's 10 k plan-steps/s "needs FP8 MFMA or fewer tokens"; the bf16 candidate pass is
certified by re-scoring the few candidates whose bf16 score comes within the bound delta of the best (m3pc_amd/certificate.py:
delta ~ 4.8, 8-14 candidates at N = 1024).  The question is what delta and that set become in FP8.

Method: the CPU oracle's candidate pass (oracle/mtm_oracle.py: learner.py:288-316 restated) with the operands of chosen GEMMs
rounded on the way in -- activations and weights alike, fp32 accumulation, everything else fp32:
    bf16      every Linear's operands rounded to bf16                    (what the HIP candidate pass does today)
    fp8_ffn   linear1 / linear2 of every transformer block in MX-FP8, every other Linear in bf16
    fp8_all   every Linear in MX-FP8
For each: d_j = score_lowprec_j - score_fp32_j over ALL candidates, its deviation from the median (the common shift the merge
removes), delta = 1.5 x the largest deviation (the planner's calibration), and need = #{j : b_j > f* + c - delta} -- the
candidates the arg-max certificate would have to re-score in fp32.  (Attention products stay fp32 here: in the HIP path they are
bf16 MFMAs with fp32 softmax; their share of the deviation is part of what the HIP path measures and this emulation does not.)

Round 6 adds the question behind the step's 102x HBM traffic: the fp32 residual stream crosses HBM between the
fused layer tails (2 KB per token row out + in per layer: a third of a tile's input bytes, half of its output bytes).  What if
that buffer were bf16?
    bf16_res  as bf16, and the residual stream is rounded to bf16 wherever it would cross HBM in the HIP candidate pass: the
              input rows of every encoder layer (the embedding kernel's X; the layer-1 tail's X''), used both by that layer's
              norm1 and by its residual add.  (Inside a layer X' never leaves the registers; the last encoder layer's X'' feeds
              encoder.norm inside the tail; the decoder's scored rows come from a candidate-independent fp32 table.)

    python -m oracle.lowprec_study [N] [seeds] [modes, comma-separated]      -> a table on stdout, JSON to gpurun_out/ or /tmp
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from m3pc_amd import synth  # noqa: E402  (the synthetic recipe weights / windows: plain numpy + torch CPU, no HIP)
from oracle import mtm_oracle as O  # noqa: E402


def round_bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def round_mxfp8(x: torch.Tensor, block: int = 32) -> torch.Tensor:
    """OCP MX e4m3: blocks of `block` consecutive elements along the last (K) dimension share a power-of-two scale
    2^(floor(log2(amax)) - 8) (8 = e4m3's largest exponent); elements are e4m3 with saturation at +-448."""
    shp = x.shape
    K = shp[-1]
    assert K % block == 0
    xb = x.reshape(-1, K // block, block)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax)))) - 8.0
    scale = torch.exp2(e)
    q = torch.clamp(xb / scale, -448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).reshape(shp)


def make_linear(sd, mode: str):
    """An F.linear stand-in that rounds its operands according to `mode` (the weight tensor's identity tells which layer)."""
    names = {id(v): k for k, v in sd.items()}
    cache = {}

    def q_weight(W, fn):
        key = (id(W), fn.__name__)
        if key not in cache:
            cache[key] = fn(W)
        return cache[key]

    def linear(x, W, b=None):
        name = names.get(id(W), "")
        K = W.shape[-1]
        if mode == "fp32" or K % 32 != 0:  # (the tiny-K encoder embeddings stay fp32: K = 11 / 3 / 1 -- not MFMA work in the HIP path either)
            return F.linear(x, W, b)
        is_ffn = ".linear1." in name or ".linear2." in name
        fn = round_mxfp8 if (mode == "fp8_all" or (mode == "fp8_ffn" and is_ffn)) else round_bf16  # (bf16_res: bf16 operands)
        return F.linear(fn(x), q_weight(W, fn), b)

    return linear


def scores(sd, stats, cfg, win, h, acts, mode):
    keep, keep_block = O.F, O._block
    O.F = types.SimpleNamespace(linear=make_linear(sd, mode), layer_norm=F.layer_norm, gelu=F.gelu)
    if mode == "bf16_res":
        def block(x, sd_, prefix, n_head):
            return keep_block(round_bf16(x) if prefix.startswith("encoder.layers.") else x, sd_, prefix, n_head)
        O._block = block
    try:
        out = []
        for c0 in range(0, acts.shape[0], 256):
            out.append(O.plan_candidates(sd, stats, cfg, win, h, acts[c0 : c0 + 256], "rtg", 0.6))
        return torch.cat(out)
    finally:
        O.F, O._block = keep, keep_block


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16", "fp8_ffn", "fp8_all", "bf16_res"]
    T, H = 32, 16
    dims = synth.Dims(11, 3, T)
    rows = []
    for ws in range(seeds):
        sd, st = synth.make_state_dict(dims, ws), synth.make_tokenizer_stats(dims, ws)
        stats = O.make_stats(st)
        cfg = O.PlanCfg(T, H, N, 0.99, 0.01, 0.6)
        hist = synth.make_history(dims, 0)
        win, h = O.assemble_window(cfg, hist, 500, 3.0)
        eps = synth.make_eps(N, dims, 1)
        loc, std = O.policy_pass(sd, stats, cfg, win, h)
        acts = O.sample_candidates(loc, std, eps, T, h) if hasattr(O, "sample_candidates") else torch.tanh(loc + std * eps)[:, 0, T - h :, 0, :]
        t0 = time.time()
        f = scores(sd, stats, cfg, win, h, acts, "fp32")
        t_f = time.time() - t0
        fbest = float(f.max())
        for mode in modes:
            b = scores(sd, stats, cfg, win, h, acts, mode)
            d = b - f
            c = float(d.median())
            dev = (d - c).abs()
            delta = 1.5 * float(dev.max())
            need = int((b > fbest + c - delta).sum())
            rows.append(dict(weight_seed=ws, mode=mode, N=N, shift=round(c, 3), dev_rms=round(float(dev.pow(2).mean().sqrt()), 3),
                             dev_max=round(float(dev.max()), 3), delta=round(delta, 3), need=need,
                             argmax_match=int(torch.argmax(b)) == int(torch.argmax(f)), score_sigma=round(float(f.std()), 2)))
            print(rows[-1], flush=True)
        print(f"# seed {ws}: fp32 pass {t_f:.1f} s on {torch.get_num_threads()} threads", flush=True)
    out_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp"
    with open(os.path.join(out_dir, f"r06_lowprec_study_N{N}.json"), "w") as fh:
        json.dump(rows, fh, indent=1)
    print("| weight seed | mode | shift | dev rms | dev max | delta = 1.5 max | need (of %d) | score sigma |" % N)
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['weight_seed']} | {r['mode']} | {r['shift']} | {r['dev_rms']} | {r['dev_max']} | {r['delta']} | {r['need']} | {r['score_sigma']} |")


if __name__ == "__main__":
    main()
