"""ctypes binding of libm3pc_hip.so (C ABI: include/m3pc_hip.h).

``import torch`` happens first on purpose: the library's DT_NEEDED ``libamdhip64.so.7`` then resolves to
the HIP runtime torch already loaded, so device pointers and streams are shared with PyTorch-ROCm.
The HIP extension is mandatory: if the shared object is missing this module raises -- there is no CPU
fallback anywhere in ``m3pc_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M3PC_LIB") or os.path.join(_HERE, "libm3pc_hip.so")  # (M3PC_LIB: tools/ point at the lab build)

STATES, ACTIONS, REWARDS, RETURNS = 0, 1, 2, 3
KEYS = ("states", "actions", "rewards", "returns")
MODE_RTG, MODE_CRITIC, MODE_NOISE = 0, 1, 2
PREC_FP32, PREC_BF16 = 0, 1
PROF_LAYER_TAIL = 16  # m3pc_profile_read: the fused layer-tail launches only
ABI_VERSION = 6
GOAL_PIID, GOAL_ID = 0, 1  # m3pc_goal_step_batch goal_mode
SLOTS = 4  # M3PC_SLOTS: plan steps in flight per handle

EXPORTS = (
    "m3pc_last_error", "m3pc_abi_version", "m3pc_create", "m3pc_destroy", "m3pc_load_weights", "m3pc_load_stats",
    "m3pc_set_tokenizer", "m3pc_set_critic", "m3pc_tokenize", "m3pc_detokenize", "m3pc_forward", "m3pc_goal_step",
    "m3pc_goal_step_batch",
    "m3pc_policy_pass", "m3pc_candidate_pass", "m3pc_candidate_join", "m3pc_policy_pass_batch",
    "m3pc_plan_step", "m3pc_plan_step_batch", "m3pc_score_actions", "m3pc_rescore", "m3pc_rescore_topk", "m3pc_topk_window",
    "m3pc_rescore_listed", "m3pc_rescore_merge", "m3pc_topk_race_window", "m3pc_rescore_merge_race", "m3pc_merge_race_select", "m3pc_select",
    "m3pc_profile_enable",
    "m3pc_profile_read",
)


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "state_dim", "action_dim", "traj_length", "n_embd", "n_head", "n_enc_layer", "n_dec_layer",
        "max_candidates", "max_batch", "critic_hidden", "max_rescore", "max_goal_batch")]


class NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_longlong), ("on_device", C.c_int)]


class PlanArgs(C.Structure):
    _fields_ = [("mode", C.c_int), ("precision", C.c_int), ("horizon", C.c_int), ("n_total", C.c_int),
                ("n_begin", C.c_int), ("n_count", C.c_int), ("lmbda", C.c_double), ("discount", C.c_double),
                ("rtg", C.c_double), ("slot", C.c_int), ("returns_f64", C.c_int), ("returns", C.c_void_p),
                ("flags", C.c_int), ("window", C.c_int)]


PLAN_DEFER_JOIN = 1
PLAN_PRUNED_POLICY = 2


class M3pcError(RuntimeError):
    pass


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen the library (no GPU needed) and declare the prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise M3pcError(f"{p} not found: build it with `python -m m3pc_amd.build` "
                        "(m3pc_amd has no fallback path without its HIP library)")
    lib = C.CDLL(p)
    vp, i, ll, f, d = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double
    lib.m3pc_last_error.restype = C.c_char_p
    lib.m3pc_last_error.argtypes = []
    lib.m3pc_abi_version.restype = i
    protos = {
        "m3pc_create": [C.POINTER(Dims), i, C.POINTER(vp)],
        "m3pc_destroy": [vp],
        "m3pc_load_weights": [vp, C.POINTER(NamedTensor), i, vp],
        "m3pc_load_stats": [vp, C.POINTER(ll)],
        "m3pc_set_tokenizer": [vp, i, C.POINTER(f), C.POINTER(f), i, i],
        "m3pc_set_critic": [vp, C.POINTER(NamedTensor), i, C.POINTER(f), C.POINTER(f), vp],
        "m3pc_tokenize": [vp, i, vp, i, vp, ll, vp],
        "m3pc_detokenize": [vp, i, vp, vp, ll, vp],
        "m3pc_forward": [vp, i, C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp, vp, i, vp],
        "m3pc_goal_step": [vp, i, vp, vp, vp, C.POINTER(d), C.POINTER(vp), C.POINTER(vp), i, vp, vp, vp, vp, vp],
        "m3pc_goal_step_batch": [vp, i, vp, vp, i, i, i, vp, vp, vp, vp],
        "m3pc_policy_pass": [vp, C.POINTER(PlanArgs), vp, vp, vp, vp, vp, vp],
        "m3pc_candidate_pass": [vp, C.POINTER(PlanArgs), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "m3pc_candidate_join": [vp, i, vp],
        "m3pc_policy_pass_batch": [vp, C.POINTER(PlanArgs), i, vp, vp, vp, C.POINTER(d), vp, vp, vp],
        "m3pc_plan_step": [vp, C.POINTER(PlanArgs), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "m3pc_plan_step_batch": [vp, C.POINTER(PlanArgs), i, vp, vp, vp, C.POINTER(d), vp, vp, vp, vp, vp, vp, vp],
        "m3pc_score_actions": [vp, C.POINTER(PlanArgs), i, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "m3pc_rescore": [vp, C.POINTER(PlanArgs), vp, vp, vp, vp, vp, i, vp, vp, vp],
        "m3pc_rescore_topk": [vp, C.POINTER(PlanArgs), vp, vp, vp, vp, vp, i, vp, vp],
        "m3pc_topk_window": [vp, vp, i, i, i, f, vp, vp, vp, vp, f, vp],
        "m3pc_rescore_merge": [vp, vp, i, vp, i, vp, vp, f, vp, vp, vp, f, vp],
        "m3pc_topk_race_window": [vp, vp, vp, f, i, i, i, i, vp, vp, vp, vp, f, vp],
        "m3pc_rescore_merge_race": [vp, vp, vp, f, i, vp, i, i, vp, vp, f, vp, vp, vp, f, vp],
        "m3pc_merge_race_select": [vp, vp, vp, f, i, vp, i, i, vp, vp, f, vp, vp, vp, f, vp, ll, vp, vp, vp, vp, vp, vp],
        "m3pc_rescore_listed": [vp, C.POINTER(PlanArgs), vp, vp, vp, vp, vp, i, vp, vp],
        "m3pc_select": [vp, vp, vp, ll, i, f, vp, vp, vp, vp, vp, vp, vp],
        "m3pc_profile_enable": [vp, i],
        "m3pc_profile_read": [vp, i, C.POINTER(ll), C.POINTER(d), C.POINTER(d), i],
    }
    for name, args in protos.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i
    if lib.m3pc_abi_version() != ABI_VERSION:
        raise M3pcError(f"ABI mismatch: library {lib.m3pc_abi_version()} vs binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise M3pcError(f"m3pc error {rc}: {load_library().m3pc_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _named(sd: Dict[str, torch.Tensor]):
    """state_dict -> (NamedTensor array, keep-alive list).  Tensors are made fp32-contiguous."""
    keep = []
    arr = (NamedTensor * len(sd))()
    for j, (k, v) in enumerate(sd.items()):
        t = v.detach().to(torch.float32).contiguous()
        nb = k.encode()
        keep.append((t, nb))
        arr[j] = NamedTensor(nb, t.data_ptr(), t.numel(), 1 if t.is_cuda else 0)
    return arr, keep


class HostStats:
    """A pinned, host-mapped float buffer a kernel writes its statistics to, followed (system scope, release) by a sequence
    number at [4]: the host reads them by spinning on the sequence number -- visible a few microseconds after the kernel
    ends, where a blocking device-to-host copy costs tens -- without synchronising a stream."""

    def __init__(self):
        self.buf = torch.zeros(8, dtype=torch.float32).pin_memory()
        self.np = self.buf.numpy()
        self.seq = 0

    def next_seq(self) -> float:
        self.seq = self.seq % 1000000 + 1
        return float(self.seq)

    def ready(self, seq: float) -> bool:
        return self.np[4] == seq

    def read(self):
        """[0..3] the four statistics every kernel writes; [4..6] = slots 5..7 of the buffer (m3pc_rescore_merge_race: need_race,
        the best listed race key, its threshold; zero for the kernels that write four)."""
        hs = self.np
        return [float(hs[0]), float(hs[1]), float(hs[2]), float(hs[3]), float(hs[5]), float(hs[6]), float(hs[7])]

    def wait(self, seq: float, fallback: Optional[torch.Tensor] = None, timeout: float = 10.0):
        import time as _t
        hs = self.np
        t0 = _t.perf_counter()
        spins = 0
        while hs[4] != seq:
            spins += 1
            if (spins & 1023) == 0 and _t.perf_counter() - t0 > timeout:  # never hang on the mapped buffer
                if fallback is not None:
                    # the device copy of the statistics is written on another stream than the current one: order behind
                    # everything on the device before reading it (this path is taken when the mapped buffer misbehaved)
                    torch.cuda.synchronize(fallback.device)
                    fb = [float(x) for x in fallback.cpu()]
                    return fb[:4] + (fb[5:8] if len(fb) >= 8 else [0.0, 0.0, 0.0])
                raise M3pcError("timed out waiting for kernel statistics in host-mapped memory")
        return self.read()


class Handle:
    """RAII wrapper of one m3pc_handle (one per process and GPU)."""

    def __init__(self, state_dim, action_dim, traj_length, n_embd=512, n_head=4, n_enc_layer=2, n_dec_layer=1,
                 max_candidates=1024, max_batch=1, critic_hidden=256, device: int = 0, max_rescore: int = 64,
                 max_goal_batch: int = 0):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise M3pcError("no HIP device visible: m3pc_amd runs only on a GPU (there is no CPU path)")
        self.device = torch.device("cuda", device)
        self.dims = Dims(state_dim, action_dim, traj_length, n_embd, n_head, n_enc_layer, n_dec_layer,
                         max_candidates, max_batch, critic_hidden, max(int(max_rescore), 1), max(int(max_goal_batch), 0))
        self.max_rescore = max(int(max_rescore), 1)
        self.max_goal_batch = max(int(max_goal_batch), 0)
        self._h = C.c_void_p()
        torch.cuda.init()
        with torch.cuda.device(self.device):
            check(self.lib.m3pc_create(C.byref(self.dims), device, C.byref(self._h)))
        self.T, self.S, self.A = traj_length, state_dim, action_dim

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.m3pc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup -------------------------------------------------------------------------------------
    def load_weights(self, state_dict: Dict[str, torch.Tensor]):
        """All tensors (first call) or any subset of them (later calls: only what depends on them is re-derived)."""
        arr, keep = _named(state_dict)
        check(self.lib.m3pc_load_weights(self._h, arr, len(state_dict), _stream(self.device)))
        del keep

    def load_stats(self):
        """{tensors, tail_streams, kv_streams, tables_invalidated} of the last ``load_weights``."""
        out = (C.c_longlong * 4)()
        check(self.lib.m3pc_load_stats(self._h, out))
        return dict(tensors=out[0], tail_streams=out[1], kv_streams=out[2], tables_invalidated=bool(out[3]))

    def set_tokenizer(self, key: int, mean, std, normalize: bool):
        m = torch.as_tensor(mean, dtype=torch.float32).contiguous().cpu().reshape(-1)
        s = torch.as_tensor(std, dtype=torch.float32).contiguous().cpu().reshape(-1)
        fp = C.POINTER(C.c_float)
        check(self.lib.m3pc_set_tokenizer(self._h, key, C.cast(m.data_ptr(), fp), C.cast(s.data_ptr(), fp),
                                          m.numel(), int(bool(normalize))))

    def set_critic(self, q_state_dict: Dict[str, torch.Tensor], obs_mean, obs_std):
        arr, keep = _named(q_state_dict)
        m = torch.as_tensor(obs_mean, dtype=torch.float32).contiguous().cpu().reshape(-1)
        s = torch.as_tensor(obs_std, dtype=torch.float32).contiguous().cpu().reshape(-1)
        fp = C.POINTER(C.c_float)
        check(self.lib.m3pc_set_critic(self._h, arr, len(q_state_dict), C.cast(m.data_ptr(), fp),
                                       C.cast(s.data_ptr(), fp), _stream(self.device)))
        del keep

    # -- tokenizer ---------------------------------------------------------------------------------
    def tokenize(self, key: int, x: torch.Tensor) -> torch.Tensor:
        assert x.is_cuda and x.dtype in (torch.float32, torch.float64)
        x = x.contiguous()
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        rows = x.numel() // x.shape[-1]
        check(self.lib.m3pc_tokenize(self._h, key, _ptr(x), int(x.dtype == torch.float64), _ptr(out), rows,
                                     _stream(self.device)))
        return out

    def detokenize(self, key: int, y: torch.Tensor) -> torch.Tensor:
        y = y.contiguous()
        out = torch.empty_like(y)
        check(self.lib.m3pc_detokenize(self._h, key, _ptr(y), _ptr(out), y.numel() // y.shape[-1],
                                       _stream(self.device)))
        return out

    # -- model -------------------------------------------------------------------------------------
    def forward(self, tokens: Sequence[Optional[torch.Tensor]], masks: Sequence, want=KEYS, precision=PREC_FP32):
        """tokens[k]: (B,T,D_k) fp32 cuda; masks[k]: (T,) 0/1.  Returns dict of raw head outputs;
        'actions' -> (mu, std)."""
        B = next(t.shape[0] for t in tokens if t is not None)
        toks = [None if t is None else t.to(torch.float32).contiguous() for t in tokens]
        tp = (C.c_void_p * 4)(*[None if t is None else t.data_ptr() for t in toks])
        mb = [bytes(bytearray(int(v != 0) for v in m)) for m in masks]
        mbuf = [C.create_string_buffer(b, len(b)) for b in mb]
        mp = (C.c_void_p * 4)(*[C.addressof(b) for b in mbuf])
        dev = self.device
        feat = (self.S, self.A, 1, 1)
        out = {}
        for k, name in enumerate(KEYS):
            if name in want and k != ACTIONS:
                out[name] = torch.empty((B, self.T, feat[k]), dtype=torch.float32, device=dev)
        mu = sd = None
        if "actions" in want:
            mu = torch.empty((B, self.T, self.A), dtype=torch.float32, device=dev)
            sd = torch.empty_like(mu)
            out["actions"] = (mu, sd)
        check(self.lib.m3pc_forward(self._h, B, tp, mp, _ptr(out.get("states")), _ptr(out.get("rewards")),
                                    _ptr(out.get("returns")), _ptr(mu), _ptr(sd), precision, _stream(dev)))
        return out

    @staticmethod
    def _mask_ptrs(masks):
        mb = [bytes(bytearray(int(v != 0) for v in m)) for m in masks]
        mbuf = [C.create_string_buffer(b, len(b)) for b in mb]
        return (C.c_void_p * 4)(*[C.addressof(b) for b in mbuf]), mbuf

    def goal_step(self, states, actions, rewards, rtg, masks_pi, masks_fid, idx: int):
        """Both forwards of the zero-shot piid call on raw windows: states (E,T,S), actions (E,T,A), rewards (E,T,1) fp32 cuda,
        rtg (E,) floats.  Returns (mu, std) (E,T,A) of the inverse-dynamics forward, the de-tokenised inferred states (E,T,S) and
        the observation rows the second forward saw (E,T,S)."""
        E = states.shape[0]
        f32 = dict(dtype=torch.float32, device=self.device)
        ins = [self._f32(t) for t in (states, actions, rewards)]
        inferred = torch.empty((E, self.T, self.S), **f32)
        window = torch.empty((E, self.T, self.S), **f32)
        mu = torch.empty((E, self.T, self.A), **f32)
        sd = torch.empty_like(mu)
        rt = (C.c_double * E)(*[float(v) for v in rtg])
        pi, keep1 = self._mask_ptrs(masks_pi)
        fid, keep2 = self._mask_ptrs(masks_fid)
        check(self.lib.m3pc_goal_step(self._h, E, _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), rt, pi, fid, int(idx), _ptr(inferred),
                                      _ptr(window), _ptr(mu), _ptr(sd), _stream(self.device)))
        return mu, sd, inferred, window

    def goal_step_batch(self, states, actions, idx: int, goal_mode: int = GOAL_PIID, precision: int = PREC_BF16,
                        want_window: bool = False, out=None):
        """The zero-shot call for many windows, exactly pruned (m3pc_goal_step_batch): states (E,T,S), actions (E,T,A) raw fp32
        cuda.  Returns (mu, std) (E,A) of the action token at ``idx`` [and the observation rows (E,T,S) the inverse-dynamics
        forward saw].  ``out``: optional preallocated (mu, std)."""
        E = states.shape[0]
        f32 = dict(dtype=torch.float32, device=self.device)
        s, a = self._f32(states), self._f32(actions)
        assert s.shape == (E, self.T, self.S) and a.shape == (E, self.T, self.A)
        mu, sd = out if out is not None else (torch.empty((E, self.A), **f32), torch.empty((E, self.A), **f32))
        window = torch.empty((E, self.T, self.S), **f32) if want_window else None
        check(self.lib.m3pc_goal_step_batch(self._h, E, _ptr(s), _ptr(a), int(idx), int(goal_mode), int(precision), _ptr(window),
                                            _ptr(mu), _ptr(sd), _stream(self.device)))
        return (mu, sd, window) if want_window else (mu, sd)

    # -- plan step ---------------------------------------------------------------------------------
    @staticmethod
    def _f32(t):
        return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()

    def _args(self, mode, precision, horizon, n_total, n_begin, n_count, lmbda, discount, rtg, slot=0, returns=None):
        a = PlanArgs(mode, precision, horizon, n_total, n_begin, n_count, lmbda, discount, rtg, slot, 0, None, 0, 0)
        if returns is not None:
            assert returns.is_cuda and returns.is_contiguous() and returns.numel() == self.T
            assert returns.dtype in (torch.float32, torch.float64)
            a.returns_f64 = int(returns.dtype == torch.float64)
            a.returns = returns.data_ptr()
        return a

    def policy_pass(self, mode: int, states, actions, rewards, horizon: int, rtg: float, slot: int = 0, returns=None,
                    loc=None, std=None, pruned: bool = False):
        """PASS 1 of a plan step on the current stream (chain workspace): leaves the policy head in ``slot``.
        returns: optional (T,) device row of raw returns (float32 / float64) instead of the constant ``rtg``.
        pruned (with loc = std = None): the head at the h action tokens the candidates are sampled from only, through the
        exactly pruned decoder (M3PC_PLAN_PRUNED_POLICY); rows t < T - h of the slot's loc / std are zero."""
        args = self._args(mode, PREC_FP32, horizon, 1, 0, 1, 0.0, 0.0, rtg, slot, returns)
        if pruned and loc is None and std is None:
            args.flags = PLAN_PRUNED_POLICY
        check(self.lib.m3pc_policy_pass(self._h, C.byref(args), _ptr(self._f32(states)), _ptr(self._f32(actions)),
                                        _ptr(self._f32(rewards)), _ptr(loc), _ptr(std), _stream(self.device)))

    def candidate_pass(self, mode: int, states, actions, rewards, eps, horizon: int, lmbda: float, discount: float,
                       n_total: int, n_begin: int = 0, n_count: Optional[int] = None, precision: int = PREC_FP32, slot: int = 0,
                       want_debug: bool = False, out=None, defer_join: bool = False, window: int = 0):
        """Candidates + PASS 2 + scores of a plan step on the current stream (candidate workspace) from ``slot``'s policy
        head.  ``out``: optional dict of preallocated loc / std / sample_actions / expect_return.  ``defer_join``: the
        current stream does not wait for the parts of the pass that run on the handle's own streams; the consumer of the
        scores calls ``candidate_join(slot)`` on its stream (M3PC_PLAN_DEFER_JOIN)."""
        n_count = n_total - n_begin if n_count is None else n_count
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        out = out or {}
        loc = out.get("loc")
        std = out.get("std")
        if loc is None:
            loc = torch.empty((self.T, self.A), **f32)
            std = torch.empty((self.T, self.A), **f32)
        acts = out.get("sample_actions")
        if acts is None:
            acts = torch.empty((n_count, horizon, self.A), **f32)
        er = out.get("expect_return")
        if er is None:
            er = torch.empty((n_count,), **f32)
        pr = torch.empty((n_count, horizon), **f32) if want_debug else None
        pb = torch.empty((n_count, horizon), **f32) if want_debug else None
        args = self._args(mode, precision, horizon, n_total, n_begin, n_count, lmbda, discount, 0.0, slot)
        args.flags = PLAN_DEFER_JOIN if defer_join else 0
        args.window = int(window)  # which of the slot's policy heads (policy_pass_batch)
        ins = [self._f32(t) for t in (states, actions, rewards, eps)]
        check(self.lib.m3pc_candidate_pass(self._h, C.byref(args), _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]),
                                           _ptr(loc), _ptr(std), _ptr(acts), _ptr(er), _ptr(pr), _ptr(pb), _stream(dev)))
        res = dict(loc=loc, std=std, sample_actions=acts, expect_return=er)
        if want_debug:
            res["pred_rewards"], res["pred_boot"] = pr, pb
        return res

    def policy_pass_batch(self, mode: int, states, actions, rewards, horizon: int, rtg, slot: int = 0, loc=None, std=None):
        """PASS 1 of E windows as one pass at batch E (policy workspace): states (E,T,S), actions (E,T,A), rewards (E,T,1),
        rtg (E,) floats.  Leaves E policy heads in ``slot``: ``candidate_pass(..., slot=slot, window=w)`` plans window w."""
        E = states.shape[0]
        args = self._args(mode, PREC_FP32, horizon, 1, 0, 1, 0.0, 0.0, 0.0, slot)
        rt = (C.c_double * E)(*[float(v) for v in rtg])
        check(self.lib.m3pc_policy_pass_batch(self._h, C.byref(args), E, _ptr(self._f32(states)), _ptr(self._f32(actions)),
                                              _ptr(self._f32(rewards)), rt, _ptr(loc), _ptr(std), _stream(self.device)))

    def candidate_join(self, slot: int):
        """Order the current stream behind every part of ``slot``'s last candidate pass enqueued with ``defer_join``."""
        check(self.lib.m3pc_candidate_join(self._h, int(slot), _stream(self.device)))

    def plan_step(self, mode: int, states, actions, rewards, eps, horizon: int, rtg: float, lmbda: float,
                  discount: float, n_total: int, n_begin: int = 0, n_count: Optional[int] = None,
                  precision: int = PREC_FP32, want_debug: bool = False, slot: int = 0, returns=None):
        n_count = n_total - n_begin if n_count is None else n_count
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        loc = torch.empty((self.T, self.A), **f32)
        std = torch.empty((self.T, self.A), **f32)
        acts = torch.empty((n_count, horizon, self.A), **f32)
        er = torch.empty((n_count,), **f32)
        pr = torch.empty((n_count, horizon), **f32) if want_debug else None
        pb = torch.empty((n_count, horizon), **f32) if want_debug else None
        args = self._args(mode, precision, horizon, n_total, n_begin, n_count, lmbda, discount, rtg, slot, returns)
        ins = [self._f32(t) for t in (states, actions, rewards, eps)]
        check(self.lib.m3pc_plan_step(self._h, C.byref(args), _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]),
                                      _ptr(loc), _ptr(std), _ptr(acts), _ptr(er), _ptr(pr), _ptr(pb), _stream(dev)))
        res = dict(loc=loc, std=std, sample_actions=acts, expect_return=er)
        if want_debug:
            res["pred_rewards"], res["pred_boot"] = pr, pb
        return res

    def plan_step_batch(self, mode: int, states, actions, rewards, rtg, eps, horizon: int, lmbda: float, discount: float,
                        n_total: int, precision: int = PREC_FP32, slot: int = 0):
        """E windows x n_total candidates in one pass: states (E,T,S), actions (E,T,A), rewards (E,T,1), rtg (E,) floats,
        eps (E, n_total, T|h, A).  Returns expect_return (E, n_total), sample_actions (E, n_total, h, A), loc/std (E,T,A)."""
        E = states.shape[0]
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        loc = torch.empty((E, self.T, self.A), **f32)
        std = torch.empty((E, self.T, self.A), **f32)
        acts = torch.empty((E, n_total, horizon, self.A), **f32)
        er = torch.empty((E, n_total), **f32)
        widx = torch.arange(E, dtype=torch.int32, device=dev).repeat_interleave(n_total).contiguous()
        args = self._args(mode, precision, horizon, n_total, 0, n_total, lmbda, discount, 0.0, slot)
        ins = [self._f32(t) for t in (states, actions, rewards, eps)]
        assert ins[3].numel() == E * n_total * (horizon if mode == MODE_NOISE else self.T) * self.A
        rt = (C.c_double * E)(*[float(v) for v in rtg])
        check(self.lib.m3pc_plan_step_batch(self._h, C.byref(args), E, _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), rt, _ptr(ins[3]),
                                            _ptr(widx), _ptr(loc), _ptr(std), _ptr(acts), _ptr(er), _stream(dev)))
        return dict(loc=loc, std=std, sample_actions=acts, expect_return=er, window_index=widx)

    def score_actions(self, mode: int, states, actions, rewards, cand: torch.Tensor, window_index: Optional[torch.Tensor],
                      horizon: int, lmbda: float, discount: float, precision: int = PREC_FP32, want_debug: bool = False,
                      slot: int = 0):
        """TD(lambda) scores of caller-supplied candidates cand (n,h,A); states/actions/rewards (E,T,.) or (T,.) windows,
        window_index (n,) int32 (None: one window).  slot: whose returns tokens (and, for few-row fp32 calls, which of the two
        chain workspaces: slot & 1) the pass uses."""
        dev = self.device
        n = cand.shape[0]
        ins = [self._f32(t) for t in (states, actions, rewards, cand)]
        E = 1 if ins[0].dim() == 2 else ins[0].shape[0]
        er = torch.empty((n,), dtype=torch.float32, device=dev)
        pr = torch.empty((n, horizon), dtype=torch.float32, device=dev) if want_debug else None
        pb = torch.empty((n, horizon), dtype=torch.float32, device=dev) if want_debug else None
        wi = None if window_index is None else window_index.to(torch.int32).contiguous()
        args = self._args(mode, precision, horizon, n, 0, n, lmbda, discount, 0.0, slot)
        check(self.lib.m3pc_score_actions(self._h, C.byref(args), E, _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]),
                                          _ptr(wi), _ptr(er), _ptr(pr), _ptr(pb), _stream(dev)))
        return (er, pr, pb) if want_debug else er

    def rescore(self, mode: int, states, actions, rewards, eps, index: torch.Tensor, horizon: int, rtg: float,
                lmbda: float, discount: float, n_total: int, slot: int = 0, out: Optional[torch.Tensor] = None, want_actions: bool = True):
        """fp32 scores (and candidates) of rows ``index`` (int32 cuda) of the plan step that owns ``slot``.
        ``out``: optional (n,) fp32 destination of the scores (a slice of a larger buffer is fine)."""
        n = index.numel()
        dev = self.device
        acts = torch.empty((n, horizon, self.A), dtype=torch.float32, device=dev) if want_actions else None
        er = out if out is not None else torch.empty((n,), dtype=torch.float32, device=dev)
        assert er.numel() == n and er.is_contiguous() and er.dtype == torch.float32
        args = self._args(mode, PREC_FP32, horizon, n_total, 0, n, lmbda, discount, rtg, slot)
        ins = [self._f32(t) for t in (states, actions, rewards, eps)]
        assert index.dtype == torch.int32 and index.is_contiguous()
        check(self.lib.m3pc_rescore(self._h, C.byref(args), _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]),
                                    _ptr(index), n, _ptr(acts), _ptr(er), _stream(dev)))
        return er, acts

    def rescore_topk(self, mode: int, states, actions, rewards, eps, expect_return: torch.Tensor, k: int,
                     horizon: int, rtg: float, lmbda: float, discount: float, slot: int = 0):
        """Replace the k largest entries of ``expect_return`` (all N candidates, contiguous fp32 cuda) by
        their fp32 re-scores, in place.  Returns the re-scored candidate ids (k,) int32."""
        n = expect_return.numel()
        assert expect_return.is_contiguous() and expect_return.dtype == torch.float32
        top = torch.empty((k,), dtype=torch.int32, device=self.device)
        args = self._args(mode, PREC_FP32, horizon, n, 0, n, lmbda, discount, rtg, slot)
        ins = [self._f32(t) for t in (states, actions, rewards, eps)]
        check(self.lib.m3pc_rescore_topk(self._h, C.byref(args), _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]),
                                         _ptr(ins[3]), _ptr(expect_return), k, _ptr(top), _stream(self.device)))
        return top

    def topk_window(self, expect_return: torch.Tensor, kmax: int, kmin: int, window: float, top: Optional[torch.Tensor] = None,
                    stats: Optional[torch.Tensor] = None, top_scores: Optional[torch.Tensor] = None, host_stats=None, seq: float = 0.0):
        """The kmax + 1 best candidates (ids, best first) and stats = [n, margin to the best one outside the n, max, raw
        count over the whole vector] where n = clamp(#{E >= max E - window}, kmin, kmax).  Both stay on the device.
        top_scores: optional (kmax + 1,) buffer receiving the listed scores; host_stats: optional pinned float tensor (>= 5)
        the kernel also writes, followed by ``seq`` (see ``HostStats``)."""
        n = expect_return.numel()
        assert expect_return.is_contiguous() and expect_return.dtype == torch.float32
        if top is None:
            top = torch.empty((kmax + 1,), dtype=torch.int32, device=self.device)
        if stats is None:
            stats = torch.empty((4,), dtype=torch.float32, device=self.device)
        check(self.lib.m3pc_topk_window(self._h, _ptr(expect_return), n, kmax, kmin, float(window), _ptr(top), _ptr(stats),
                                        _ptr(top_scores), None if host_stats is None else C.c_void_p(host_stats.data_ptr()),
                                        float(seq), _stream(self.device)))
        return top, stats

    def topk_window_issue(self, expect_return: torch.Tensor, kmax: int, kmin: int, window: float):
        """``topk_window`` with its statistics also written to a pinned, host-mapped buffer followed by a sequence number.
        Returns (top ids on the device, ticket for ``topk_window_wait``); one ticket outstanding at a time."""
        if not hasattr(self, "_hs"):
            self._hs = HostStats()
        seq = self._hs.next_seq()
        top, stats = self.topk_window(expect_return, kmax, kmin, window, host_stats=self._hs.buf, seq=seq)
        return top, (seq, stats)

    def topk_window_wait(self, ticket):
        """The statistics of ``topk_window_issue`` on the host without a stream synchronisation.
        [n, margin_outside, max, raw count] as python floats."""
        seq, stats = ticket
        return self._hs.wait(seq, stats)

    def topk_window_host(self, expect_return: torch.Tensor, kmax: int, kmin: int, window: float):
        """``topk_window_issue`` + ``topk_window_wait``: (top ids on the device, statistics on the host)."""
        top, ticket = self.topk_window_issue(expect_return, kmax, kmin, window)
        return top, self.topk_window_wait(ticket)

    def rescore_listed(self, mode: int, states, actions, rewards, eps, expect_return: torch.Tensor, index: torch.Tensor,
                       horizon: int, rtg: float, lmbda: float, discount: float, slot: int = 0):
        """Replace the entries ``index`` (int32 cuda) of ``expect_return`` by their fp32 re-scores, in place."""
        n = expect_return.numel()
        assert index.dtype == torch.int32 and index.is_contiguous() and expect_return.is_contiguous()
        args = self._args(mode, PREC_FP32, horizon, n, 0, n, lmbda, discount, rtg, slot)
        ins = [self._f32(t) for t in (states, actions, rewards, eps)]
        check(self.lib.m3pc_rescore_listed(self._h, C.byref(args), _ptr(ins[0]), _ptr(ins[1]), _ptr(ins[2]), _ptr(ins[3]),
                                           _ptr(index), index.numel(), _ptr(expect_return), _stream(self.device)))

    def rescore_merge(self, scores: torch.Tensor, index: torch.Tensor, n: int, top_scores: torch.Tensor, top_rescored: torch.Tensor,
                      delta: float = 0.0, merged: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, host_stats=None,
                      seq: float = 0.0):
        """merged = scores - median(top_scores[:n] - top_rescored[:n]), entries index[:n] replaced by top_rescored[:n];
        stats = [shift, largest deviation from it, need = #{scores > best fp32 + shift - delta}, margin].  See m3pc_rescore_merge."""
        assert scores.is_contiguous() and scores.dtype == torch.float32 and index.dtype == torch.int32
        if merged is None:
            merged = torch.empty_like(scores)
        if stats is None:
            stats = torch.empty((4,), dtype=torch.float32, device=self.device)
        check(self.lib.m3pc_rescore_merge(self._h, _ptr(scores), scores.numel(), _ptr(index), int(n), _ptr(top_scores),
                                          _ptr(top_rescored), float(delta), _ptr(merged), _ptr(stats),
                                          None if host_stats is None else C.c_void_p(host_stats.data_ptr()), float(seq),
                                          _stream(self.device)))
        return merged, stats

    def topk_race_window(self, expect_return: torch.Tensor, expo: torch.Tensor, temperature: float, kmax: int, kmin: int, rmax: int,
                         lst: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None,
                         list_scores: Optional[torch.Tensor] = None, want_stats: bool = True):
        """``topk_window`` plus the rmax best candidates by race key temperature * E - log expo (the multinomial draw's race),
        in one list: lst[rmax + i] = i-th best by score (i <= kmax), lst[rmax - 1 - i] = i-th best racer.  Returns (lst, stats)."""
        n = expect_return.numel()
        assert expect_return.is_contiguous() and expect_return.dtype == torch.float32
        assert expo.is_contiguous() and expo.dtype == torch.float32 and expo.numel() == n
        if lst is None:
            lst = torch.empty((rmax + kmax + 1,), dtype=torch.int32, device=self.device)
        if stats is None and want_stats:
            stats = torch.empty((4,), dtype=torch.float32, device=self.device)
        assert lst.numel() >= rmax + min(kmax + 1, n)
        check(self.lib.m3pc_topk_race_window(self._h, _ptr(expect_return), _ptr(expo), float(temperature), n, int(kmax), int(kmin),
                                             int(rmax), _ptr(lst), _ptr(stats), _ptr(list_scores), None, 0.0, _stream(self.device)))
        return lst, stats

    def rescore_merge_race(self, scores: torch.Tensor, expo: torch.Tensor, temperature: float, lst: torch.Tensor, r: int, n: int,
                           list_scores: torch.Tensor, list_rescored: torch.Tensor, delta: float = 0.0,
                           merged: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, host_stats=None, seq: float = 0.0):
        """``rescore_merge`` over r race entries followed by n score entries (lst / list_scores / list_rescored start at the first
        race entry); stats (8 floats) = [shift, deviation, need, margin, -, need_race, best listed race key, its threshold]."""
        assert scores.is_contiguous() and scores.dtype == torch.float32 and lst.dtype == torch.int32
        assert lst.numel() >= r + n and list_scores.numel() >= r + n and list_rescored.numel() >= r + n
        if merged is None:
            merged = torch.empty_like(scores)
        if stats is None:
            stats = torch.empty((8,), dtype=torch.float32, device=self.device)
        assert stats.numel() >= 8
        check(self.lib.m3pc_rescore_merge_race(self._h, _ptr(scores), _ptr(expo), float(temperature), scores.numel(), _ptr(lst), int(r),
                                               int(n), _ptr(list_scores), _ptr(list_rescored), float(delta), _ptr(merged), _ptr(stats),
                                               None if host_stats is None else C.c_void_p(host_stats.data_ptr()), float(seq),
                                               _stream(self.device)))
        return merged, stats

    def merge_race_select(self, scores: torch.Tensor, expo: torch.Tensor, temperature: float, lst: torch.Tensor, r: int, n: int,
                          list_scores: torch.Tensor, list_rescored: torch.Tensor, a0: torch.Tensor, delta: float = 0.0,
                          merged: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, host_stats=None, seq: float = 0.0,
                          out=None):
        """``rescore_merge_race`` + ``select`` on the merged vector in one launch.  Returns (merged, stats, (p, eval_action, argmax,
        sample_idx, sample_action))."""
        nt = scores.numel()
        assert scores.is_contiguous() and scores.dtype == torch.float32 and lst.dtype == torch.int32
        assert lst.numel() >= r + n and list_scores.numel() >= r + n and list_rescored.numel() >= r + n
        assert a0.shape[0] == nt and a0.stride(-1) == 1 and expo.numel() == nt and expo.is_contiguous()
        if merged is None:
            merged = torch.empty_like(scores)
        if stats is None:
            stats = torch.empty((8,), dtype=torch.float32, device=self.device)
        p, ev, am, si, sa = out if out is not None else self.select_buffers(nt)
        check(self.lib.m3pc_merge_race_select(self._h, _ptr(scores), _ptr(expo), float(temperature), nt, _ptr(lst), int(r), int(n),
                                              _ptr(list_scores), _ptr(list_rescored), float(delta), _ptr(merged), _ptr(stats),
                                              None if host_stats is None else C.c_void_p(host_stats.data_ptr()), float(seq),
                                              _ptr(a0), a0.stride(0), _ptr(p), _ptr(ev), _ptr(am), _ptr(si), _ptr(sa),
                                              _stream(self.device)))
        return merged, stats, (p, ev, am, si, sa)

    def select_buffers(self, n: int):
        """Output tensors of ``select`` (p, eval_action, argmax, sample_idx, sample_action), allocated on the current stream."""
        dev = self.device
        return (torch.empty((n,), dtype=torch.float32, device=dev), torch.empty((self.A,), dtype=torch.float32, device=dev),
                torch.empty((1,), dtype=torch.int32, device=dev), torch.empty((1,), dtype=torch.int32, device=dev),
                torch.empty((1, self.A), dtype=torch.float32, device=dev))

    def select(self, expect_return: torch.Tensor, a0: torch.Tensor, temperature: float,
               expo: Optional[torch.Tensor] = None, out=None):
        """a0: (N, A) view (may be a strided slice sample_actions[:, 0]).  Returns (p, eval_action, argmax)
        and, when ``expo`` (N Exp(1) variates) is given, also (sample_idx, sample_action (1, A)).
        out: optional ``select_buffers(n)`` tuple to write into."""
        n = expect_return.numel()
        assert a0.shape[0] == n and a0.stride(-1) == 1
        p, ev, am, si, sa = out if out is not None else self.select_buffers(n)
        if expo is None:
            si = sa = None
        else:
            assert expo.numel() == n and expo.dtype == torch.float32 and expo.is_contiguous()
        check(self.lib.m3pc_select(self._h, _ptr(expect_return.contiguous()), _ptr(a0), a0.stride(0), n,
                                   float(temperature), _ptr(expo), _ptr(p), _ptr(ev), _ptr(am), _ptr(si), _ptr(sa),
                                   _stream(self.device)))
        if expo is None:
            return p, ev, am
        return p, ev, am, si, sa

    # -- profiling ---------------------------------------------------------------------------------
    def profile_enable(self, on):
        """False / True, or 2: also run the candidate halves one after the other (each bracket holds one launch alone)."""
        check(self.lib.m3pc_profile_enable(self._h, int(on)))

    def profile_read(self, precision: int = -1, reset: bool = True):
        n, ms, fl = C.c_longlong(), C.c_double(), C.c_double()
        check(self.lib.m3pc_profile_read(self._h, precision, C.byref(n), C.byref(ms), C.byref(fl), int(reset)))
        return n.value, ms.value, fl.value
