"""Actor-critic policies whose parameters live in ONE flat f32 device buffer (what the HIP
kernels, the fused clip+Adam and the RCCL gradient all-reduce operate on) while exposing the
reference's / torch's parameter names as views.

  MLPActorCritic   the reference's network, PPOV2.0/model.py:17-53 (state_dict keys feature.{0,1,3,4}.*,
                   actor.*, critic.*; orthogonal init gains sqrt(2) / 0.01 / 1.0, zero biases)
  LSTMActorCritic  the LSTM actor-critic BASELINE.json specifies: nn.LSTM(obs, H, L) semantics
                   (PPOV2.0/model.py:206-212) -> actor Linear(H, A) | critic Linear(H, 1)
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops


class _FlatPolicy:
    """flat parameter buffer + named views (no copies)."""

    def _alloc(self, layout, device):
        self.layout = layout                       # list of (name, shape)
        total = sum(int(np.prod(s)) for _, s in layout)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.views, self.grad_views = {}, {}
        o = 0
        for name, shape in layout:
            n = int(np.prod(shape))
            self.views[name] = self.flat[o:o + n].view(shape)
            self.grad_views[name] = self.grad[o:o + n].view(shape)
            o += n

    @property
    def device(self):
        return self.flat.device

    def num_params(self):
        return self.flat.numel()


def _orthogonal(shape, gain, gen):
    w = torch.empty(shape)
    torch.nn.init.orthogonal_(w, gain=gain, generator=gen)
    return w


class MLPActorCritic(_FlatPolicy):
    KEYS = ("feature.0.weight", "feature.0.bias", "feature.1.weight", "feature.1.bias",
            "feature.3.weight", "feature.3.bias", "feature.4.weight", "feature.4.bias",
            "actor.weight", "actor.bias", "critic.weight", "critic.bias")

    def __init__(self, input_size=6, output_size=5, h1=256, h2=128, device="cuda", seed=None):
        self.in_dim, self.n_act, self.h1, self.h2 = input_size, output_size, h1, h2
        A = output_size
        # flat order of csrc/mlp.hip (heads contiguous: actor rows then the critic row)
        self._alloc([("feature.0.weight", (h1, input_size)), ("feature.0.bias", (h1,)),
                     ("feature.1.weight", (h1,)), ("feature.1.bias", (h1,)),
                     ("feature.3.weight", (h2, h1)), ("feature.3.bias", (h2,)),
                     ("feature.4.weight", (h2,)), ("feature.4.bias", (h2,)),
                     ("head.weight", (A + 1, h2)), ("head.bias", (A + 1,))], device)
        assert self.flat.numel() == ops.mlp_param_count(input_size, h1, h2, A)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        sd = {  # model.py:29-40
            "feature.0.weight": _orthogonal((h1, input_size), math.sqrt(2), gen), "feature.0.bias": torch.zeros(h1),
            "feature.1.weight": torch.ones(h1), "feature.1.bias": torch.zeros(h1),
            "feature.3.weight": _orthogonal((h2, h1), math.sqrt(2), gen), "feature.3.bias": torch.zeros(h2),
            "feature.4.weight": torch.ones(h2), "feature.4.bias": torch.zeros(h2),
            "actor.weight": _orthogonal((A, h2), 0.01, gen), "actor.bias": torch.zeros(A),
            "critic.weight": _orthogonal((1, h2), 1.0, gen), "critic.bias": torch.zeros(1),
        }
        self.load_state_dict(sd)
        self._stash = None

    # -- reference-compatible (de)serialisation -------------------------------------------------
    def _named(self, src):
        A = self.n_act
        out = {k: src[k] for k in self.KEYS[:8]}
        out["actor.weight"], out["critic.weight"] = src["head.weight"][:A], src["head.weight"][A:]
        out["actor.bias"], out["critic.bias"] = src["head.bias"][:A], src["head.bias"][A:]
        return out

    def named_views(self):
        return self._named(self.views)

    def named_grads(self):
        return self._named(self.grad_views)

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.named_views().items()}

    def load_state_dict(self, sd):
        nv = self.named_views()
        for k in self.KEYS:
            nv[k].copy_(torch.as_tensor(sd[k], dtype=torch.float32).reshape(nv[k].shape))

    # -- compute ----------------------------------------------------------------------------------
    def heads(self, x, stash=None):
        """x [B, in] -> heads [B, A+1] (logits | value); keeps the stash for backward()."""
        heads, self._stash = ops.mlp_fwd(self.flat, x, self.in_dim, self.h1, self.h2, self.n_act, stash=stash)
        self._x = x
        return heads

    def backward(self, dheads):
        """d(loss)/d(heads) -> self.grad (flat, overwritten).  Consumes the stash."""
        ops.mlp_bwd(self.flat, self._x, self._stash, dheads, self.in_dim, self.h1, self.h2, self.n_act, grad=self.grad)
        self._stash = None
        return self.grad


class LSTMActorCritic(_FlatPolicy):
    def __init__(self, obs_dim=6, hidden=128, num_layers=1, n_act=5, device="cuda", seed=0):
        self.obs_dim, self.hidden, self.num_layers, self.n_act = obs_dim, hidden, num_layers, n_act
        H, A = hidden, n_act
        layout = []
        for l in range(num_layers):
            i = obs_dim if l == 0 else H
            layout += [(f"lstm.weight_ih_l{l}", (4 * H, i)), (f"lstm.weight_hh_l{l}", (4 * H, H)),
                       (f"lstm.bias_ih_l{l}", (4 * H,)), (f"lstm.bias_hh_l{l}", (4 * H,))]
        layout += [("head.weight", (A + 1, H)), ("head.bias", (A + 1,))]
        self._alloc(layout, device)
        gen = torch.Generator().manual_seed(seed)
        k = 1.0 / math.sqrt(H)
        for name, shape in layout[:-2]:            # nn.LSTM default init U(-1/sqrt(H), 1/sqrt(H))
            self.views[name].copy_((torch.rand(shape, generator=gen) * 2 - 1) * k)
        self.views["head.weight"][:A].copy_(_orthogonal((A, H), 0.01, gen))       # heads as model.py:34-40
        self.views["head.weight"][A:].copy_(_orthogonal((1, H), 1.0, gen))
        self.views["head.bias"].zero_()
        self._saved = None

    def _named(self, src):
        A = self.n_act
        out = {k: v for k, v in src.items() if k.startswith("lstm.")}
        out["actor.weight"], out["critic.weight"] = src["head.weight"][:A], src["head.weight"][A:]
        out["actor.bias"], out["critic.bias"] = src["head.bias"][:A], src["head.bias"][A:]
        return out

    def named_views(self):
        return self._named(self.views)

    def named_grads(self):
        return self._named(self.grad_views)

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.named_views().items()}

    def load_state_dict(self, sd):
        nv = self.named_views()
        for k, v in nv.items():
            v.copy_(torch.as_tensor(sd[k], dtype=torch.float32).reshape(v.shape))

    def zero_state(self, n):
        z = torch.zeros(self.num_layers, n, self.hidden, dtype=torch.float32, device=self.device)
        return z, z.clone()

    def heads(self, obs, keep, h0, c0, work=None, want_heads=True):
        """obs [N,T,I], keep [N,T] or None, h0,c0 [L,N,H] -> heads [N*T, A+1] (logits | value).  Saves what
        backward() needs.  `work` may hold preallocated 'stash{l}', 'y{l}', 'heads' tensors.  The top layer's
        sequence kernel applies the actor / critic rows itself (uav_lstm_fwd heads output)."""
        N, T, _ = obs.shape
        v = self.views
        x = obs
        saved = []
        work = work or {}
        heads = None
        if self._interleaved_forward_ok(N, work, T):
            # stacked h = 256 layers: the steppers of the rollout (weights split once, the layer above reading the piece planes
            # of the layer below) step all layers time step by time step -- uav_lstm_fwd layer by layer converts the layer
            # below's whole output to piece planes first (split_x_kernel) and re-splits the weights per call; same kernels,
            # BIT-identical stash / y (tests/test_gpu_lstm.py::test_lstm_stepper_equals_sequence_forward)
            self.begin_steps(h0, c0)
            keep_t = None if keep is None else keep.t().contiguous()
            kt = (lambda t: None) if keep_t is None else (lambda t: keep_t[t])
            if self.num_layers == 2 and getattr(self, "use_step_pairs", True):
                # layer 1's step t + 1 beside layer 2's step t, one launch (uav_lstm_stepper_step_pair): T + 1 launches for 2 T steps
                s0, s1 = self._steppers
                y0, y1, st0, st1 = work["y0"], work["y1"], work["stash0"], work["stash1"]
                s0.step(obs, 0, y0, st0, keep=kt(0))
                for t in range(T - 1):
                    ops.lstm_stepper_step_pair((s0, obs, t + 1, y0, st0, None, kt(t + 1)), (s1, y0, t, y1, st1, s0, kt(t)))
                s1.step(y0, T - 1, y1, st1, below=s0, keep=kt(T - 1))
            else:
                for t in range(T):
                    self.step_layers_at(obs, t, work, keep=kt(t))
            for l in range(self.num_layers):
                saved.append((x, work[f"stash{l}"], work[f"y{l}"], h0[l]))
                x = work[f"y{l}"]
            self._saved = (saved, keep, x)
            if not want_heads:
                return x.view(N * T, self.hidden)
            heads = work.get("heads")
            if heads is None or heads.shape[0] != N:
                heads = torch.empty(N, T, self.n_act + 1, dtype=torch.float32, device=obs.device)
            ops.gemm_rows(x.view(N * T, self.hidden), v["head.weight"], v["head.bias"], heads.view(N * T, self.n_act + 1))
            return heads.view(N * T, self.n_act + 1)
        for l in range(self.num_layers):
            top = want_heads and l == self.num_layers - 1
            if top:
                heads = work.get("heads")
                if heads is None or heads.shape[0] != N:
                    heads = torch.empty(N, T, self.n_act + 1, dtype=torch.float32, device=obs.device)
            y, hn, cn, stash = ops.lstm_fwd(x, keep, h0[l], c0[l], v[f"lstm.weight_ih_l{l}"], v[f"lstm.weight_hh_l{l}"],
                                            v[f"lstm.bias_ih_l{l}"], v[f"lstm.bias_hh_l{l}"],
                                            stash=work.get(f"stash{l}"), y=work.get(f"y{l}"),
                                            w_head=v["head.weight"] if top else None,
                                            b_head=v["head.bias"] if top else None, heads=heads if top else None)
            saved.append((x, stash, y, h0[l]))
            x = y
        self._saved = (saved, keep, x)
        if not want_heads:
            return x.view(N * T, self.hidden)          # trunk output (uav_ppo_loss_from_y applies the heads)
        return heads.view(N * T, self.n_act + 1)

    def adopt_forward(self, obs, keep, h0, stash, y):
        """Epoch 0 of a PPO update runs with the rollout's parameters: the rollout already wrote this forward pass's
        stash and y (the fused rollout kernel; for stacked / h = 256 policies the stepper, one list entry per layer),
        so register them instead of recomputing."""
        stashes = stash if isinstance(stash, (list, tuple)) else [stash]
        ys = y if isinstance(y, (list, tuple)) else [y]
        saved, x = [], obs
        for l in range(len(ys)):
            saved.append((x, stashes[l], ys[l], h0[l]))
            x = ys[l]
        self._saved = (saved, keep, x)
        N, T, H = x.shape
        return x.view(N * T, H)

    def _interleaved_forward_ok(self, N, work, T=None):
        """The update's forward pass can run on the rollout's steppers: a stack of h = 256 layers on the fp16-split step path,
        steppers already built for this many envs, and the [N, T] arrays they fill present WITH this call's T (a caller whose
        obs has another horizon than the preallocated work arrays takes the uav_lstm_fwd layer loop, which allocates).
        NOTE: that path runs on the ROLLOUT's steppers and overwrites their hn / cn and piece planes; a rollout copies its
        final state out (trainer._collect_stepwise_lstm) before any update can run, and begin_steps() re-seeds them."""
        if not getattr(self, "use_stepper_forward", True) or self.num_layers < 2 or self.hidden != 256:
            return False
        sp = getattr(self, "_steppers", None)
        if sp is None or sp[0].N != N or ops.lstm_bwd_caps(self.device, self.obs_dim, 256) == 0:
            return False
        return all(work.get(f"y{l}") is not None and work[f"y{l}"].shape[0] == N and work.get(f"stash{l}") is not None
                   and work[f"stash{l}"].shape[0] == N
                   and (T is None or (work[f"y{l}"].shape[1] == T and work[f"stash{l}"].shape[1] == T))
                   for l in range(self.num_layers))

    def steppers(self, N, device):
        """One ops.LstmStepper per layer (uav_lstm_stepper_*: h = 256, fp16-split arithmetic), created once."""
        if getattr(self, "_steppers", None) is None or self._steppers[0].N != N:
            self._steppers = [ops.LstmStepper(N, self.obs_dim if l == 0 else self.hidden, self.hidden, device)
                              for l in range(self.num_layers)]
        return self._steppers

    def begin_steps(self, h, c):
        """Start a step-wise rollout from state h, c [L, N, H]: weights split once, state moved into the steppers."""
        v = self.views
        for l, sp in enumerate(self.steppers(h.shape[1], h.device)):
            sp.begin(v[f"lstm.weight_ih_l{l}"], v[f"lstm.weight_hh_l{l}"], v[f"lstm.bias_ih_l{l}"], v[f"lstm.bias_hh_l{l}"],
                     h[l], c[l])

    def step_at(self, obs_seq, t, work, heads_seq, keep=None):
        """Time step t of all layers on the [N, T, ...] arrays of the update (obs_seq [N, T, I] row t already written;
        work['y{l}'], work['stash{l}'] filled at t); keep [N]: this step's restart mask; heads_seq [N, T, A+1] row t =
        actor / critic rows of the top layer."""
        x = self.step_layers_at(obs_seq, t, work, keep=keep)
        v = self.views
        return ops.gemm_rows(x[:, t], v["head.weight"], v["head.bias"], heads_seq[:, t])

    def step_layers_at(self, obs_seq, t, work, keep=None):
        """The recurrent layers of step_at only; returns the top layer's [N, T, H] output array (row t filled)."""
        x = obs_seq
        for l, sp in enumerate(self._steppers):
            sp.step(x, t, work[f"y{l}"], work[f"stash{l}"], below=self._steppers[l - 1] if l > 0 and self.hidden == 256 else None,
                    keep=keep)
            x = work[f"y{l}"]
        return x

    def step(self, obs, h, c, keep=None, work=None):
        """One time step for N envs (step-wise rollout of configurations the fused rollout kernel does
        not cover: stacked layers, h = 256).  obs [N, I]; h, c [L, N, H] updated in place; keep [N] or None.
        Returns heads [N, A+1]."""
        N = obs.shape[0]
        x = obs.view(N, 1, -1)
        k = None if keep is None else keep.view(N, 1)
        work = work if work is not None else {}
        v = self.views
        for l in range(self.num_layers):
            st = work.get(f"step_stash{l}")
            if st is None:
                st = work[f"step_stash{l}"] = torch.empty(N, 1, 6 * self.hidden, dtype=torch.float32, device=obs.device)
            y, hn, cn, _ = ops.lstm_fwd(x, k, h[l], c[l], v[f"lstm.weight_ih_l{l}"], v[f"lstm.weight_hh_l{l}"],
                                        v[f"lstm.bias_ih_l{l}"], v[f"lstm.bias_hh_l{l}"], stash=st)
            h[l].copy_(hn)
            c[l].copy_(cn)
            x = y
        return ops.gemm(x.view(N, self.hidden), v["head.weight"], trans_b=True, bias=v["head.bias"])

    def backward(self, dheads, work=None, dhead_bias=None):
        """dheads [N*T, A+1] -> self.grad (flat, overwritten).  dhead_bias: the column sums of dheads
        when the loss kernel already produced them (uav_ppo_loss), else computed here."""
        saved, keep, y_last = self._saved
        N, T, H = y_last.shape
        v, g = self.views, self.grad_views
        work = work or {}
        if dhead_bias is not None:
            if dhead_bias.data_ptr() != g["head.bias"].data_ptr():      # the loss kernel may write the gradient view directly
                g["head.bias"].copy_(dhead_bias)
        else:
            ops.colsum(dheads, out=g["head.bias"])
        dy = None
        top_in = self.obs_dim if self.num_layers == 1 else self.hidden
        caps = ops.lstm_bwd_caps(dheads.device, top_in, self.hidden)
        fused_heads = bool(caps & 2)                 # dy = dheads . W_head formed on chip
        L = self.num_layers
        if L >= 2 and (caps & 4) and getattr(self, "use_stack_bwd", True):
            # h = 256 stack: all layers' BPTTs as ONE pipelined call (uav_lstm_bwd_stack: the layer below follows the one
            # above by one step on its own stream, the HBM-bound gate-gradient kernel of one layer under the latency-bound
            # recurrent product of the other), then the weight gradients layer by layer
            specs = []
            for l in reversed(range(L)):
                x, stash, y, h0 = saved[l]
                key = f"dgates{l}"
                dg_bytes = ops.lstm_dgates_bytes(N, T, H, dheads.device)          # opaque at h = 256 (fp16 piece chunks)
                if work.get(key) is None or work[key].numel() * 4 < dg_bytes:
                    work[key] = work["dgates"] if (l == L - 1 and work.get("dgates") is not None and work["dgates"].numel() * 4 >= dg_bytes) \
                        else ops.lstm_dgates(N, T, H, dheads.device)
                if l > 0 and (work.get(f"dx{l}") is None or work[f"dx{l}"].shape[0] != N):
                    work[f"dx{l}"] = torch.empty(N, T, H, dtype=torch.float32, device=dheads.device)
                specs.append({"stash": stash, "w_hh": v[f"lstm.weight_hh_l{l}"], "w_ih": v[f"lstm.weight_ih_l{l}"] if l > 0 else None,
                              "dgates": work[key], "dx": work[f"dx{l}"] if l > 0 else None})
            ops.lstm_bwd_stack(specs, keep, dheads=dheads.view(N, T, -1), w_head=v["head.weight"])
            for l in reversed(range(L)):
                x, stash, y, h0 = saved[l]
                top = (l == L - 1)
                ops.lstm_bwd(x, keep, stash, v[f"lstm.weight_ih_l{l}"], v[f"lstm.weight_hh_l{l}"], y, h0,
                             wgrad_dheads=dheads.view(N, T, -1) if top else None, dgates=work[f"dgates{l}"],
                             dw_ih=g[f"lstm.weight_ih_l{l}"], dw_hh=g[f"lstm.weight_hh_l{l}"], db=g[f"lstm.bias_ih_l{l}"],
                             dw_head=g["head.weight"] if top else None, want_dstate=False, bwd_done=True,
                             db_hh=g[f"lstm.bias_hh_l{l}"])
            self._saved = None
            return self.grad
        if not fused_heads:
            if work.get("dy") is None or work["dy"].shape[0] != N * T:
                work["dy"] = torch.empty(N * T, H, dtype=torch.float32, device=dheads.device)
            dy = ops.gemm(dheads, v["head.weight"], out=work["dy"]).view(N, T, H)
        for l in reversed(range(self.num_layers)):
            x, stash, y, h0 = saved[l]
            top = (l == self.num_layers - 1)
            use_dheads = top and fused_heads
            r = ops.lstm_bwd(x, keep, stash, v[f"lstm.weight_ih_l{l}"], v[f"lstm.weight_hh_l{l}"], y, h0,
                             dy=None if use_dheads else dy,
                             dheads=dheads.view(N, T, -1) if use_dheads else None,
                             w_head=v["head.weight"] if use_dheads else None,
                             wgrad_dheads=dheads.view(N, T, -1) if top else None,
                             need_dx=(l > 0), dgates=work.get("dgates"), dw_ih=g[f"lstm.weight_ih_l{l}"],
                             dw_hh=g[f"lstm.weight_hh_l{l}"], db=g[f"lstm.bias_ih_l{l}"],
                             dw_head=g["head.weight"] if top else None, want_dstate=False, db_hh=g[f"lstm.bias_hh_l{l}"])
            dy = r["dx"]
        self._saved = None
        return self.grad
