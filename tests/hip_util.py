"""Shared helpers of the GPU parity tests: build a libm3pc_hip handle loaded with the synthetic
recipe weights, and the matching oracle-side objects."""
import numpy as np
import torch

from m3pc_amd import capi, synth
from oracle import mtm_oracle as O


def lab_library():
    """libm3pc_hip_lab.so (-DM3PC_LAB): the product ABI plus the kernel-level hooks of include/m3pc_hip_debug.h and the
    environment A/B switches.  Built on demand (it travels to the GPU box with the snapshot like the product library)."""
    from m3pc_amd import build
    return capi.load_library(build.build_library(lab=True))


def make_handle(dims: synth.Dims, max_candidates=64, max_batch=4, seed=0):
    h = capi.Handle(dims.state_dim, dims.action_dim, dims.traj_length, dims.n_embd, dims.n_head, dims.n_enc_layer,
                    dims.n_dec_layer, max_candidates=max_candidates, max_batch=max_batch, critic_hidden=256)
    sd = synth.make_state_dict(dims, seed)
    h.load_weights(sd)
    stats = synth.make_tokenizer_stats(dims, seed)
    for k, name in enumerate(synth.KEYS):
        h.set_tokenizer(k, stats[name]["mean"], stats[name]["std"], normalize=(name != "actions"))
    qsd, om, os_ = synth.make_critic(dims, seed)
    h.set_critic(qsd, om, os_)
    return h, sd, O.make_stats(stats), (qsd, om, os_)


def window_dev(win):
    """oracle window dict -> raw fp32 device tensors (T,S) (T,A) (T,1) the C ABI takes."""
    return (win["states"][0].cuda(), win["actions"][0].cuda(), win["rewards"][0].cuda())


def maxerr(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max())
