"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A CPU restatement (plain PyTorch fp32 tensor ops, no ``nn.Module``, no reference imports)
of the m3pc test-time MPC plan step.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; ``m3pc_amd`` never does.

Parity pin: every function here is checked against golden vectors produced by importing the
real reference (``/root/reference``) in the build container -- ``tests/golden/make_golden.py``
(generator, committed) and ``tests/golden/*.npz`` (vectors, committed);
``tests/test_oracle_golden.py`` is the pin.  The reference's own test-suite holds no vector
for this path.

Third-party arithmetic: the transformer block of the reference is
``torch.nn.TransformerEncoderLayer(batch_first=True, norm_first=True, activation="gelu")``
(reference pins pytorch==1.12.1, README.md:24; fixtures were generated with torch 2.10.0).
Its published algorithm is restated in ``_block`` below: pre-LN, eps=1e-5, packed in-proj with
rows [Q;K;V], 1/sqrt(head_dim) scaling, exact-erf GELU.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

KEYS = ("states", "actions", "rewards", "returns")


# ----------------------------------------------------------------------------------------------
# masks   (research/finetune_omtm/masks.py:7-44, research/zeroshot_omtm/masks.py:30-108)
# ----------------------------------------------------------------------------------------------
def _mask_dict(T: int, s, a, r, g) -> Dict[str, np.ndarray]:
    out = {}
    for k, v in zip(KEYS, (s, a, r, g)):
        m = np.zeros(T, dtype=np.float64)
        m[v] = 1.0
        out[k] = m
    return out


def rcbc_mask(T: int, idx: int):
    """finetune_omtm/masks.py:7-27: states[:idx+1], actions[:idx], all returns, no rewards."""
    return _mask_dict(T, slice(0, idx + 1), slice(0, max(idx, 0)), slice(0, 0), slice(0, T))


def fd_mask(T: int, idx: int):
    """finetune_omtm/masks.py:30-44: states[:idx+1], every action, nothing else."""
    return _mask_dict(T, slice(0, idx + 1), slice(0, T), slice(0, 0), slice(0, 0))


def fid_mask(T: int, idx: int):
    """zeroshot_omtm/masks.py:30-47: every state, actions[:idx]."""
    return _mask_dict(T, slice(0, T), slice(0, max(idx, 0)), slice(0, 0), slice(0, 0))


def pi_mask(T: int, idx: int):
    """zeroshot_omtm/masks.py:72-91 (== create_gid_mask 50-69): every state except idx+1..T-2
    (only when idx>0), actions[:idx]."""
    m = fid_mask(T, idx)
    if idx > 0:
        m["states"][idx + 1 : -1] = 0.0
    return m


gid_mask = pi_mask


# ----------------------------------------------------------------------------------------------
# tokenizer   (research/omtm/tokenizers/continuous.py:68-94)
# ----------------------------------------------------------------------------------------------
class Stats:
    """mean/std/min/max of one key + the 'normalize' flag (actions: False, continuous.py:59-61)."""

    def __init__(self, mean, std, lo, hi, normalize: bool):
        self.mean = torch.as_tensor(np.asarray(mean), dtype=torch.float32)
        self.std = torch.as_tensor(np.asarray(std), dtype=torch.float32)
        self.min = np.asarray(lo, dtype=np.float32)
        self.max = np.asarray(hi, dtype=np.float32)
        self.normalize = normalize


def make_stats(stats: Dict[str, Dict[str, np.ndarray]]) -> Dict[str, Stats]:
    return {k: Stats(v["mean"], v["std"], v["min"], v["max"], normalize=(k != "actions")) for k, v in stats.items()}


def tok_encode(x: torch.Tensor, st: Stats) -> torch.Tensor:
    """continuous.py:68-79.  (B,T,D) -> (B,T,1,D) fp32; a float64 input is normalised in
    float64 first (type promotion with the fp32 mean/std), exactly like the reference."""
    if st.normalize:
        x = (x - st.mean) / st.std
    return x.unsqueeze(2).to(torch.float32)


def tok_decode(y: torch.Tensor, st: Stats) -> torch.Tensor:
    """continuous.py:81-94.  (B,T,1,D) -> (B,T,D)."""
    if st.normalize:
        return y.squeeze(2) * st.std + st.mean
    return y


def encode_all(traj: Dict[str, torch.Tensor], stats: Dict[str, Stats]):
    """tokenizers/base.py:69-83."""
    return {k: tok_encode(v, stats[k]) for k, v in traj.items()}


# ----------------------------------------------------------------------------------------------
# model   (research/omtm/models/mtm_model.py)
# ----------------------------------------------------------------------------------------------
def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _block(x: torch.Tensor, sd, prefix: str, n_head: int) -> torch.Tensor:
    """One nn.TransformerEncoderLayer(norm_first=True, gelu) in eval mode (mtm_model.py:379-409)."""
    B, L, d = x.shape
    hd = d // n_head
    h = _ln(x, sd, prefix + ".norm1")
    qkv = F.linear(h, sd[prefix + ".self_attn.in_proj_weight"], sd[prefix + ".self_attn.in_proj_bias"])
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(B, L, n_head, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = k.reshape(B, L, n_head, hd).transpose(1, 2)
    v = v.reshape(B, L, n_head, hd).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, L, d)
    x = x + F.linear(o, sd[prefix + ".self_attn.out_proj.weight"], sd[prefix + ".self_attn.out_proj.bias"])
    h = _ln(x, sd, prefix + ".norm2")
    h = F.gelu(F.linear(h, sd[prefix + ".linear1.weight"], sd[prefix + ".linear1.bias"]))
    return x + F.linear(h, sd[prefix + ".linear2.weight"], sd[prefix + ".linear2.bias"])


def _n_layers(sd, stack: str) -> int:
    n = 0
    while f"{stack}.layers.{n}.norm1.weight" in sd:
        n += 1
    return n


def mtm_forward(sd, tokens: Dict[str, torch.Tensor], masks: Dict[str, np.ndarray], n_head: int,
                want: Optional[Sequence[str]] = None, taps: Optional[dict] = None):
    """omtm.forward (mtm_model.py:593-607) for P=1 token per timestep.

    tokens[k]: (B,T,1,D_k) fp32 (already tokenised); masks[k]: (T,) 0/1.
    Returns dict k -> (B,T,1,D_k) tensors; for "actions" a (loc, std) pair
    (DiagGaussianActor, mtm_model.py:313-321).  ``taps`` (optional dict) receives intermediates.
    """
    keys = list(tokens.keys())
    pos = sd["pos_embed"]  # (1,T,1,d)
    # trajectory_encoding 546-557
    emb = {}
    for k in keys:
        x = tokens[k].to(torch.float32)
        e = F.linear(x, sd[f"encoder_embed_dict.{k}.weight"], sd[f"encoder_embed_dict.{k}.bias"])
        e = e + sd[f"encoder_per_dim_encoding.{k}"] + pos[:, : x.shape[1]]
        emb[k] = e.reshape(e.shape[0], -1, e.shape[-1])
    # forward_encoder 619-644 with _index 534-544
    kept, restore = {}, {}
    feats = []
    for k in keys:
        m = torch.as_tensor(np.asarray(masks[k]))
        ids = (m == 1).nonzero(as_tuple=True)[0]
        zero_ids = (m == 0).nonzero(as_tuple=True)[0]
        restore[k] = torch.argsort(torch.hstack((ids, zero_ids)))
        kept[k] = len(ids)
        feats.append(emb[k][:, ids])
    x = torch.cat(feats, dim=1)
    if taps is not None:
        taps["enc_in"] = x
    for i in range(_n_layers(sd, "encoder")):
        x = _block(x, sd, f"encoder.layers.{i}", n_head)
    x = _ln(x, sd, "encoder.norm")
    if taps is not None:
        taps["enc_out"] = x
    # forward_decoder 663-716 with _decoder_trajectory_encoding 646-661
    dec_in = []
    off = 0
    for k in keys:
        z = x[:, off : off + kept[k]]
        off += kept[k]
        n_masked = restore[k].shape[0] - kept[k]
        mt = sd[f"mask_token_dict.{k}"].expand(z.shape[0], n_masked, -1)
        z = torch.cat([z, mt], dim=1)[:, restore[k]]
        e = F.linear(z, sd[f"decoder_embed_dict.{k}.weight"], sd[f"decoder_embed_dict.{k}.bias"])
        e = e.unsqueeze(2) + sd[f"decoder_per_dim_encoding.{k}"] + pos[:, : e.shape[1]]
        dec_in.append(e.reshape(e.shape[0], -1, e.shape[-1]))
    y = torch.cat(dec_in, dim=1)
    if taps is not None:
        taps["dec_in"] = y
    for i in range(_n_layers(sd, "decoder")):
        y = _block(y, sd, f"decoder.layers.{i}", n_head)
    y = _ln(y, sd, "decoder.norm")
    if taps is not None:
        taps["dec_out"] = y
    out = {}
    off = 0
    for k in keys:
        t_p = restore[k].shape[0]
        seg = y[:, off : off + t_p].unsqueeze(2)  # (B,T,1,d)
        off += t_p
        if want is not None and k not in want:
            continue
        if k == "actions":
            mu = F.linear(seg, sd["output_head_dict.actions.mu.weight"], sd["output_head_dict.actions.mu.bias"])
            ls = F.linear(seg, sd["output_head_dict.actions.log_std.weight"], sd["output_head_dict.actions.log_std.bias"])
            ls = torch.tanh(ls)
            ls = -5.0 + 0.5 * (2.0 - (-5.0)) * (ls + 1.0)
            out[k] = (mu, ls.exp())
        else:
            h = _ln(seg, sd, f"output_head_dict.{k}.0")
            h = F.gelu(F.linear(h, sd[f"output_head_dict.{k}.1.weight"], sd[f"output_head_dict.{k}.1.bias"]))
            out[k] = F.linear(h, sd[f"output_head_dict.{k}.3.weight"], sd[f"output_head_dict.{k}.3.bias"])
    return out


def twinq(qsd, obs_mean, obs_std, state: torch.Tensor, action: torch.Tensor) -> torch.Tensor:
    """TwinQ.forward = min(q1,q2) (finetune_omtm/model.py:163-171, MLP 72-104)."""
    s = (state - obs_mean) / obs_std
    sa = torch.cat([s, action], 1)
    qs = []
    for q in ("q1", "q2"):
        h = sa
        for li in (0, 2):
            h = torch.relu(F.linear(h, qsd[f"{q}.net.{li}.weight"], qsd[f"{q}.net.{li}.bias"]))
        qs.append(F.linear(h, qsd[f"{q}.net.4.weight"], qsd[f"{q}.net.4.bias"]).squeeze(-1))
    return torch.min(qs[0], qs[1])


# ----------------------------------------------------------------------------------------------
# plan step   (research/finetune_omtm/learner.py)
# ----------------------------------------------------------------------------------------------
class PlanCfg:
    """The cfg fields the path reads (finetune.py RunConfig; learner.py:276,319,342)."""

    def __init__(self, traj_length, horizon, action_samples, discount=0.99, temperature=1.0, lmbda=0.6,
                 plan_guidance="rtg_guiding", n_head=4):
        self.traj_length = traj_length
        self.horizon = horizon
        self.action_samples = action_samples
        self.discount = discount
        self.temperature = temperature
        self.lmbda = lmbda
        self.plan_guidance = plan_guidance
        self.n_head = n_head


def assemble_window(cfg: PlanCfg, hist: Dict[str, np.ndarray], path_length: int, rtg: float):
    """Window assembly of action_sample (learner.py:342-385).  Returns (traj dict, horizon).
    states/actions/rewards are fp32 (1,T,D); returns is float64 (1,T,1) filled with rtg."""
    T = cfg.traj_length
    horizon = cfg.horizon
    end = int(path_length)
    if end + horizon < T:
        horizon = T - end
    hl = T - horizon + 1
    win = {}
    for src, dst in (("observations", "states"), ("actions", "actions"), ("rewards", "rewards")):
        z = np.zeros((1, T, hist[src].shape[-1]))
        z[0, :hl] = hist[src][end - hl + 1 : end + 1]
        win[dst] = torch.tensor(z, dtype=torch.float32)
    win["returns"] = torch.from_numpy(float(rtg) * np.ones((1, T, 1)))
    return win, horizon


def explore_rtg(stats: Dict[str, Stats], percentage: float) -> float:
    """learner.py:377-381."""
    lo, hi = stats["returns"].min, stats["returns"].max
    return float(np.asarray(lo + (hi - lo) * percentage).reshape(-1)[0])


def policy_pass(sd, stats, cfg: PlanCfg, traj, h: int):
    """PASS 1: rcbc forward at batch 1 -> (loc, std) of shape (1,T,1,A) (learner.py:278-284)."""
    T = cfg.traj_length
    out = mtm_forward(sd, encode_all(traj, stats), rcbc_mask(T, T - h), cfg.n_head, want=("actions",))
    return out["actions"]


def td_lambda(cfg: PlanCfg, rewards: torch.Tensor, boot: torch.Tensor, lmbda: float) -> torch.Tensor:
    """The scoring loop (learner.py:300-316 / 240-257), in the reference's operation order.
    rewards (N,h): predicted rewards; boot (N,h): bootstrap value at each step."""
    N, h = rewards.shape
    er = torch.zeros((N,))
    for t in range(h):
        values = torch.zeros((N, t + 1))
        values[:, t] = boot[:, t]
        disc = torch.cumprod(cfg.discount * torch.ones((t + 1,)), dim=0)
        if t > 0:
            values[:, :t] = rewards[:, :t]
        values *= disc[None, :]
        if t < h - 1:
            er += values.sum(dim=-1) * (1 - lmbda) * (lmbda**t)
        else:
            er += values.sum(dim=-1) * (lmbda**t)
    return er


def select(cfg: PlanCfg, expect_return: torch.Tensor, a0: torch.Tensor):
    """learner.py:318-323: p and the softmax-weighted first action."""
    er = expect_return - torch.max(expect_return)
    score = er * cfg.temperature
    p = torch.exp(score) / torch.exp(score).sum()
    eval_action = (a0 * p[:, None]).sum(dim=0) / p.sum()
    return p, eval_action


def plan_candidates(sd, stats, cfg: PlanCfg, traj, h: int, sample_actions: torch.Tensor, mode: str,
                    lmbda: float, critic=None, taps: Optional[dict] = None) -> torch.Tensor:
    """PASS 2 + scoring for a given (n,h,A) block of candidate action sequences
    (learner.py:288-316 rtg / 228-257 critic).  Returns expect_return (n,) BEFORE the max shift."""
    T = cfg.traj_length
    n = sample_actions.shape[0]
    batch = {k: v.repeat(n, 1, 1) for k, v in traj.items()}
    batch["actions"][:, T - h :, :] = sample_actions
    want = ("rewards", "returns") if mode == "rtg" else ("states", "rewards")
    out = mtm_forward(sd, encode_all(batch, stats), fd_mask(T, T - h), cfg.n_head, want=want, taps=taps)
    dec = {k: tok_decode(v, stats[k]) for k, v in out.items()}
    rewards = dec["rewards"][:, T - h :, 0]
    if mode == "rtg":
        boot = dec["returns"][:, T - h :, 0] * 1000
    else:
        qsd, om, os_ = critic
        fs = dec["states"][:, T - h :, :]
        boot = torch.stack([twinq(qsd, om, os_, fs[:, t], sample_actions[:, t]) for t in range(h)], dim=1)
    if taps is not None:
        taps["rewards"] = rewards
        taps["boot"] = boot
        for k, v in dec.items():
            taps["dec_" + k] = v
    return td_lambda(cfg, rewards, boot, lmbda)


def sample_candidates(loc, std, eps: torch.Tensor, T: int, h: int) -> torch.Tensor:
    """SquashedNormal.sample((N,)) with explicit normals (mtm_model.py:263-269, learner.py:285-287):
    eps (N,1,T,1,A) -> tanh(loc + std*eps)[:, 0, T-h:, 0, :]  (N,h,A)."""
    return torch.tanh(eps * std + loc)[:, 0, T - h :, 0, :]


def guiding(sd, stats, cfg: PlanCfg, traj, h: int, lmbda: float, eps: torch.Tensor, mode: str,
            critic=None, generator: Optional[torch.Generator] = None, chunk: int = 0, taps=None):
    """rtg_guiding (learner.py:271-327), critic_lambda_guiding (211-268) and, with mode="noise",
    noise_adding_lambda (142-208).  ``eps`` replaces the reference's internal normal draw:
    (N,1,T,1,A) for rtg/critic, (N,h,A) for noise.  Returns a dict of everything observable."""
    T, N = cfg.traj_length, cfg.action_samples
    loc, std = policy_pass(sd, stats, cfg, traj, h)
    if mode == "noise":
        mean = torch.tanh(loc)[0, T - h :, 0, :]
        acts = torch.clamp(mean + eps * 0.09, -0.99999, 0.99999)
        score_mode = "critic"
    else:
        acts = sample_candidates(loc, std, eps, T, h)
        score_mode = mode
    if chunk and chunk < N:
        er = torch.cat([plan_candidates(sd, stats, cfg, traj, h, acts[i : i + chunk], score_mode, lmbda, critic)
                        for i in range(0, N, chunk)])
    else:
        er = plan_candidates(sd, stats, cfg, traj, h, acts, score_mode, lmbda, critic, taps=taps)
    p, eval_action = select(cfg, er, acts[:, 0])
    res = {"loc": loc, "std": std, "sample_actions": acts, "expect_return": er, "p": p,
           "eval_action": eval_action, "argmax": int(torch.argmax(er))}
    if generator is not None:
        idx = torch.multinomial(p, 1, generator=generator)
        res["sample_idx"] = idx
        res["sample_action"] = acts[idx, 0]
    return res


def mtm_sampling(sd, stats, cfg: PlanCfg, traj, h: int, eps: Optional[torch.Tensor] = None):
    """learner.py:103-115: no-plan path; eps (1,T,1,A) optional normals for the sampled action."""
    T = cfg.traj_length
    loc, std = policy_pass(sd, stats, cfg, traj, h)
    eval_action = torch.tanh(loc)[0, T - h]
    sample_action = None if eps is None else torch.tanh(eps * std + loc)[0, T - h]
    return sample_action, eval_action


# ----------------------------------------------------------------------------------------------
# zero-shot goal reaching   (research/zeroshot_omtm/learner.py:60-261)
# ----------------------------------------------------------------------------------------------
def assemble_goal_window(cfg: PlanCfg, hist: Dict[str, np.ndarray], path_length: int, rtg: float):
    """zeroshot learner.py:164-223: the history window plus FUTURE observations (waypoints)
    copied over the whole window, shortened near the 1000-step episode end."""
    T = cfg.traj_length
    horizon = cfg.horizon
    end = int(path_length)
    if end + horizon < T:
        horizon = T - end
    smart = T
    if end + horizon > 1000:
        smart = smart - (end + horizon - 1000)
    hl = T - horizon + 1
    win = {}
    for src, dst in (("observations", "states"), ("actions", "actions"), ("rewards", "rewards")):
        z = np.zeros((1, T, hist[src].shape[-1]))
        z[0, :hl] = hist[src][end - hl + 1 : end + 1]
        if src == "observations":
            z[0, :smart] = hist[src][end - hl + 1 : end - hl + 1 + T]
        win[dst] = torch.tensor(z, dtype=torch.float32)
    win["returns"] = torch.from_numpy(float(rtg) * np.ones((1, T, 1)))
    return win, horizon


def goal_piid(sd, stats, cfg: PlanCfg, traj, h: int):
    """action_piid_sample's two chained forwards (zeroshot learner.py:225-256).
    Returns (loc, std) of the second pass and the path-inference states."""
    T = cfg.traj_length
    idx = T - h
    out = mtm_forward(sd, encode_all(traj, stats), pi_mask(T, idx), cfg.n_head, want=("states",))
    inferred = tok_decode(out["states"], stats["states"])
    traj = {k: v.clone() for k, v in traj.items()}
    traj["states"][:, idx + 2 : -1, :] = inferred[:, idx + 2 : -1, :]
    traj["states"][:, : idx + 1, :] = inferred[:, : idx + 1, :]
    out2 = mtm_forward(sd, encode_all(traj, stats), fid_mask(T, idx), cfg.n_head, want=("actions",))
    loc, std = out2["actions"]
    return loc, std, inferred


def goal_piid_list(sd, stats, cfg: PlanCfg, traj, h: int):
    """action_piid_list_sample (zeroshot learner.py:263-370; goal_mask "piid_allout"): the same two chained forwards; what the
    reference leaves in ``self.action_list`` is [mean of the action distribution at T-h] -- tanh(loc), whatever ``eval`` says
    (the entries for T-h+1, T-h+2 are commented out at 366-370).  Returns that list."""
    loc, _, _ = goal_piid(sd, stats, cfg, traj, h)
    return [torch.tanh(loc[0, cfg.traj_length - h])]


def goal_id(sd, stats, cfg: PlanCfg, traj, h: int):
    """action_id_sample's single forward under the gid mask (zeroshot learner.py:135-149)."""
    T = cfg.traj_length
    out = mtm_forward(sd, encode_all(traj, stats), gid_mask(T, T - h), cfg.n_head, want=("actions",))
    return out["actions"]


def cem_guiding(sd, stats, cfg: PlanCfg, traj, h: int, lmbda: float, noise: torch.Tensor, mode: str = "rtg", critic=None,
                iterations: int = 2, top_k: int = 128, init_std: float = 0.1):
    """Cross-entropy refinement of the plan: the algorithm of the legacy ``sample_action_cem``
    (research/omtm/datasets/sequence_dataset.py:919-1000: noisy copies of the policy mean, score, keep top_k, refit mean
    and std, resample, clamp to [-1, 1]) on this model's own pieces -- policy_pass for the mean, plan_candidates
    (learner.py:288-316) for the score.  The legacy function itself predates the four-key model and does not run on it:
    this restatement is the parity pin of HipPlanner.cem_guiding.  noise: (iterations+1, N, h, A) standard normals."""
    T = cfg.traj_length
    loc, _ = policy_pass(sd, stats, cfg, traj, h)
    mean = torch.tanh(loc[0, T - h :, 0, :])
    std = torch.full_like(mean, init_std)
    cand = torch.clamp(mean[None] + std[None] * noise[0], -1.0, 1.0)
    trace = []
    k = min(top_k, cand.shape[0])
    for it in range(iterations):
        er = plan_candidates(sd, stats, cfg, traj, h, cand, mode, lmbda, critic)
        top = torch.topk(er, k).indices
        elite = cand[top]
        mean = elite.mean(dim=0)
        std = elite.std(dim=0) if k > 1 else torch.zeros_like(mean)
        trace.append(dict(expect_return=er, top=top, mean=mean, std=std))
        cand = torch.clamp(mean[None] + std[None] * noise[it + 1], -1.0, 1.0)
    return dict(sample_action=cand[0, 0][None], eval_action=mean[0], trace=trace, candidates=cand)
