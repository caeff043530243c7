#!/usr/bin/env python3
"""How deep in the bf16 ranking does the fp32 arg-max sit?  (sizing of the fp32 re-score set; GPU tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from m3pc_amd import capi, synth
from hip_util import make_handle
from oracle import mtm_oracle as O

def main():
    worst = 0; errs = []
    for env, T, H, N in (("hopper", 32, 16, 1024), ("walker2d", 32, 16, 1024)):
        S, A = synth.ENV_DIMS[env]
        dims = synth.Dims(S, A, T)
        for wseed in (0, 1, 2):
            h, sd, stats, critic = make_handle(dims, max_candidates=N, max_batch=1, seed=wseed)
            cfg = O.PlanCfg(T, H, N)
            for trial in range(8):
                hist = synth.make_history(dims, 10 + trial)
                win, hh = O.assemble_window(cfg, hist, 200 + 37 * trial, 3.0)
                eps = synth.make_eps(N, dims, 100 + trial)[:, 0, :, 0, :].cuda()
                s, a, r = win["states"][0].cuda(), win["actions"][0].cuda(), win["rewards"][0].cuda()
                f = h.plan_step(capi.MODE_RTG, s, a, r, eps, hh, 3.0, 0.6, 0.99, N)["expect_return"].clone()
                b = h.plan_step(capi.MODE_RTG, s, a, r, eps, hh, 3.0, 0.6, 0.99, N, precision=capi.PREC_BF16)["expect_return"].clone()
                am = int(torch.argmax(f))
                rank = int((b > b[am]).sum())
                worst = max(worst, rank)
                errs.append(float((b - f).abs().max()) / float(f.abs().max()))
                top2 = torch.topk(f, 2).values
                print(f"{env} w{wseed} t{trial}: fp32 argmax has bf16 rank {rank}; max|dE|/scale {errs[-1]:.2e}; fp32 top-1 margin {float(top2[0]-top2[1]):.3f}; spread(std) {float(f.std()):.2f}")
            h.close()
    print("worst rank", worst, "max rel err", max(errs))

if __name__ == "__main__":
    main()
