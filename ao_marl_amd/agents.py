"""Multi-agent layout and batched SAC actor / critic FORWARD.

The reference runs one SAC agent per OS process and talks to it through torch RPC, one state slice
out and one (action, mean) back per agent per step (train_rpc.py:706-732, 925-953).  Here every
agent's MLP is a slice of stacked weight tensors and one `torch.bmm` per layer evaluates all
agents for all environments on the GPU the environments live on (PyTorch-ROCm is used for exactly
these matmuls, nothing else on the hot path).

Restated reference logic (host side, cited per function):
  create_agents_dictionary_original   train_rpc.py:265-296
  get_state_shape_worker              train_rpc.py:310-334
  modes_chosen_window_n_zernike       helper_rpc/helper_states.py:202-283 (incl. its TT quirk)
  modes_chosen_original               helper_rpc/helper_states.py:29-67
  GaussianPolicy.forward / sample     algorithms_rpc/model_rpc.py:121-158
  QNetwork.forward                    algorithms_rpc/model_rpc.py:56-69
"""
import math
from collections import OrderedDict

import numpy as np
import torch

LOG_SIG_MIN = -20.0


class AgentLayout(object):
    """Which Btt modes each agent drives and which state entries it sees."""

    def __init__(self, nmodes, n_zernike_start_end, n_agents_modal, include_tip_tilt=True,
                 window_n_zernike=-1, include_tip_tilt_windowed=False, n_filtered=0,
                 state_keys=("dm_history_2", "dm_history_1", "dm_before_linear", "dm_residual"),
                 state_block=None):
        lo, hi = n_zernike_start_end
        assert lo > -1 and hi > -1 and hi > lo                       # train_rpc.py:280-281
        total_controlled = hi - lo
        assert total_controlled % n_agents_modal == 0                # train_rpc.py:272-279
        local = total_controlled // n_agents_modal
        self.nmodes = nmodes
        self.start = lo
        self.local_controlled_modes = local
        self.agents = OrderedDict()                                  # worker_id -> [lo, hi)
        wid = 1
        for m in range(lo, hi, local):
            self.agents[wid] = (m, m + local)
            wid += 1
        self.include_tip_tilt = include_tip_tilt
        if include_tip_tilt:
            self.agents[wid] = (nmodes - 2, nmodes)
            total_controlled += 2
        self.total_controlled_modes = total_controlled
        self.n_agents = len(self.agents)
        self.window = window_n_zernike
        # --- state blocks: with a window (or tt_treated_as_mode) every block is the full modal
        #     vector, otherwise the controlled subset (ao_env.py:482-505)
        self.block = state_block if state_block is not None else \
            (nmodes if window_n_zernike > -1 else total_controlled)
        self.state_keys = tuple(state_keys)
        self.indices_of_state = OrderedDict()
        o = 0
        for k in self.state_keys:
            self.indices_of_state[k] = (o, o + self.block)
            o += self.block
        self.state_dim = o
        if window_n_zernike > -1:
            self.modes_chosen = self._windowed(window_n_zernike, include_tip_tilt_windowed,
                                               n_filtered)
        else:
            self.modes_chosen = self._original()
        # --- action layout: concatenated agent outputs, modal agents first, TT last
        #     (train_rpc.py:667-675, select_correct_modes_for_array)
        self.action_slices = OrderedDict()
        for w, (a, b) in self.agents.items():
            if include_tip_tilt and (a, b) == (nmodes - 2, nmodes):
                self.action_slices[w] = (total_controlled - 2, total_controlled)
            else:
                self.action_slices[w] = (a - lo, b - lo)
        self.action_dim = total_controlled
        # the Btt mode each action component drives (rlSupervisor.py:677-691)
        self.action_modes = np.r_[np.arange(lo, hi), [nmodes - 2, nmodes - 1]] \
            if include_tip_tilt else np.arange(lo, hi)

    def _original(self):
        out = OrderedDict()
        for w, (a, b) in self.agents.items():
            if self.include_tip_tilt and (a, b) == (self.nmodes - 2, self.nmodes):
                bot, top = self.total_controlled_modes - 2 - self.start, \
                    self.total_controlled_modes - self.start
            else:
                bot, top = a - self.start, b - self.start
            parts = []
            for k, (s0, s1) in self.indices_of_state.items():
                if k in "wfs":                       # substring test, as in the reference
                    parts.append(np.arange(s0, s1))
                else:
                    parts.append(np.arange(s0 + bot, s0 + top))
            out[w] = np.concatenate(parts)
        return out

    def _windowed(self, w_n, tt_windowed, n_filtered):
        out = OrderedDict()
        n_ag = len(self.agents)
        for w, (a, b) in self.agents.items():
            parts = []
            if self.include_tip_tilt and w == n_ag:
                for k, (s0, s1) in self.indices_of_state.items():
                    if k in "wfs":
                        parts.append(np.arange(s0, s1))
                    else:
                        idx = np.arange(s1 - 2, s1)
                        if tt_windowed:
                            # reference quirk (helper_states.py:239-243): ABSOLUTE indices
                            # 0 .. 2w-1, not offset by the block start
                            idx = np.concatenate([idx, np.arange(0, int(2 * w_n))])
                        parts.append(idx)
                out[w] = np.concatenate(parts)
                break
            for k, (s0, s1) in self.indices_of_state.items():
                if k in "wfs":
                    parts.append(np.arange(s0, s1))
                    continue
                end_point = s1 - (n_filtered - 2)
                length = end_point - s0
                if a - w_n < 0:
                    diff = a - w_n
                    ini, end = 0, b + w_n - diff
                elif b + w_n > length:
                    diff = b + w_n - length
                    ini, end = a - int(w_n) - diff, length
                else:
                    ini, end = a - w_n, b + w_n
                parts.append(np.arange(s0 + ini, s0 + end))
            out[w] = np.concatenate(parts)
        return out

    def state_shapes(self):
        return [len(v) for v in self.modes_chosen.values()]

    def action_shapes(self):
        return [b - a for (a, b) in self.agents.values()]


def _xavier_uniform(gen, fan_out, fan_in, gain=1.0):
    a = gain * math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(fan_out, fan_in, generator=gen) * 2 - 1) * a


class BatchedGaussianPolicy(object):
    """All agents' GaussianPolicy MLPs (Linear-ReLU x num_layers, mean / log_std heads) as
    stacked, zero-padded tensors; forward for [nenv] environments x [A] agents with torch.bmm.

    Weight layout: W1 [A, in_max, H], Wh [L-1][A, H, H], Wm / Ws [A, H, act_max] (+ biases).
    `gather_idx` [A, in_max] indexes the environment state (padding -> a zero column appended to
    the state), `scatter_idx` [action_dim] places agent outputs into the global action vector.
    """

    def __init__(self, layout, hidden=256, num_layers=2, log_sig_max=2.0, action_scale=1.0,
                 action_bias=0.0, last_layer_zero=True, seed=1234, device="cuda:0"):
        self.layout, self.H, self.L = layout, hidden, num_layers
        self.log_sig_max, self.scale, self.bias = log_sig_max, action_scale, action_bias
        self.device = torch.device(device)
        ins, acts = layout.state_shapes(), layout.action_shapes()
        A, in_max, act_max = layout.n_agents, max(ins), max(acts)
        self.A, self.in_max, self.act_max = A, in_max, act_max
        g = torch.Generator().manual_seed(seed)
        W1 = torch.zeros(A, in_max, hidden)
        Wh = [torch.zeros(A, hidden, hidden) for _ in range(num_layers - 1)]
        Wm, Ws = torch.zeros(A, hidden, act_max), torch.zeros(A, hidden, act_max)
        gather = torch.full((A, in_max), layout.state_dim, dtype=torch.long)
        for i, (w, idx) in enumerate(layout.modes_chosen.items()):
            W1[i, :ins[i], :] = _xavier_uniform(g, hidden, ins[i]).T   # model_rpc.py:10-14
            for h in Wh:
                h[i] = _xavier_uniform(g, hidden, hidden).T
            if not last_layer_zero:                                    # model_rpc.py:103-106
                Wm[i, :, :acts[i]] = _xavier_uniform(g, acts[i], hidden).T
                Ws[i, :, :acts[i]] = _xavier_uniform(g, acts[i], hidden).T
            gather[i, :ins[i]] = torch.as_tensor(idx, dtype=torch.long)
        dev = self.device
        self.W1, self.Wh, self.Wm, self.Ws = W1.to(dev), [h.to(dev) for h in Wh], Wm.to(dev), \
            Ws.to(dev)
        self.b1 = torch.zeros(A, 1, hidden, device=dev)
        self.bh = [torch.zeros(A, 1, hidden, device=dev) for _ in range(num_layers - 1)]
        self.bm = torch.zeros(A, 1, act_max, device=dev)
        self.bs = torch.zeros(A, 1, act_max, device=dev)
        self.gather = gather.to(dev)
        # scatter: global action index -> (agent, local index)
        ag = torch.zeros(layout.action_dim, dtype=torch.long)
        lo = torch.zeros(layout.action_dim, dtype=torch.long)
        for i, (w, (a, b)) in enumerate(layout.action_slices.items()):
            ag[a:b] = i
            lo[a:b] = torch.arange(b - a)
        self.sc_agent, self.sc_local = ag.to(dev), lo.to(dev)
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        self.gather_i32 = self.gather.to(torch.int32).contiguous()
        self.sc_agent_i32, self.sc_local_i32 = self.sc_agent.to(torch.int32), self.sc_local.to(torch.int32)
        self.seed, self._draws = int(seed), 0
        self._native = None          # stacked nn.Linear-layout copies for the HIP batched GEMM
        self.use_native = self.device.type == "cuda"
        self.native_forward, self._desc = True, None     # select_action as one library call
        self.layer_by_layer = False                      # ... and (False) as one kernel

    def _refresh_native(self):
        """[A, out, in] (K-contiguous) copies + merged mean|log_std head for
        libaomarl.linear_batched (fused bias + ReLU epilogue)."""
        t = lambda w: w.transpose(1, 2).contiguous()   # noqa: E731
        self._native = dict(
                W1=t(self.W1), b1=self.b1[:, 0].contiguous(),
                Wh=[t(w) for w in self.Wh], bh=[b[:, 0].contiguous() for b in self.bh],
                Whead=torch.cat([t(self.Wm), t(self.Ws)], dim=1).contiguous(),
                bhead=torch.cat([self.bm[:, 0], self.bs[:, 0]], dim=1).contiguous())

    def load_agent(self, i, state_dict):
        """Load one reference actor checkpoint (`model_state_dict`, train_rpc.py:1155-1161) into
        slot i: linear1 / hidden.N / mean_linear / log_std_linear weights [out, in]."""
        ins, acts = self.layout.state_shapes()[i], self.layout.action_shapes()[i]
        with torch.no_grad():
            self.W1[i].zero_()
            self.W1[i, :ins] = state_dict["linear1.weight"].T.to(self.device)
            self.b1[i, 0] = state_dict["linear1.bias"].to(self.device)
            for j in range(self.L - 1):
                self.Wh[j][i] = state_dict["hidden.%d.weight" % j].T.to(self.device)
                self.bh[j][i, 0] = state_dict["hidden.%d.bias" % j].to(self.device)
            self.Wm[i].zero_(); self.Ws[i].zero_()
            self.Wm[i, :, :acts] = state_dict["mean_linear.weight"].T.to(self.device)
            self.Ws[i, :, :acts] = state_dict["log_std_linear.weight"].T.to(self.device)
            self.bm[i, 0, :acts] = state_dict["mean_linear.bias"].to(self.device)
            self.bs[i, 0, :acts] = state_dict["log_std_linear.bias"].to(self.device)
        self._native = None

    def split_states(self, state):
        """[nenv, state_dim] -> [A, nenv, in_max] (TrainerRPC.divide_states_for_agents)."""
        if self.use_native and not state.requires_grad:
            from . import libaomarl as la
            return la.split_states(state.to(torch.float32), self.gather_i32)
        padded = torch.cat([state, state.new_zeros(state.shape[0], 1)], dim=1)
        return padded[:, self.gather].permute(1, 0, 2).contiguous()

    def _native_head(self, state):
        """[A, nenv, 2 * act_max] = mean | log_std (unclamped), all layers on the HIP batched GEMM."""
        from . import libaomarl as la
        if self._native is None:
            self._refresh_native()
        n = self._native
        x = self.split_states(state.to(torch.float32))
        x = la.linear_batched(x, n["W1"], n["b1"], relu=True)
        for W, b in zip(n["Wh"], n["bh"]):
            x = la.linear_batched(x, W, b, relu=True)
        return la.linear_batched(x, n["Whead"], n["bhead"], relu=False)

    def _actor_desc(self, nenv):
        """aomarl_actor_desc + scratch for `nenv` environments (rebuilt when the weights or the batch
        size change)."""
        from . import libaomarl as la
        import ctypes as C
        if self._native is None:
            self._refresh_native()
        if self._desc is not None and self._desc[0].nenv == nenv and self._desc[4] is self._native and \
                bool(self._desc[0].flags) == self.layer_by_layer:
            return self._desc[0]
        n, A, H = self._native, self.A, self.H
        d = la.ActorDesc()
        d.n_agents, d.nenv, d.state_dim, d.in_max, d.act_max = A, nenv, self.layout.state_dim, \
            self.in_max, self.act_max
        d.hidden, d.n_hidden, d.action_dim = H, self.L, self.layout.action_dim
        z = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=self.device)  # noqa: E731
        scratch = dict(x=z(A, nenv, self.in_max), h0=z(A, nenv, H), h1=z(A, nenv, H),
                       head=z(A, nenv, 2 * self.act_max))
        nh = max(self.L - 1, 1)
        Wh = (C.c_void_p * nh)(*[w.data_ptr() for w in n["Wh"]])
        bh = (C.c_void_p * nh)(*[b.data_ptr() for b in n["bh"]])
        d.gather, d.W1, d.b1 = self.gather_i32.data_ptr(), n["W1"].data_ptr(), n["b1"].data_ptr()
        d.Wh, d.bh = Wh, bh
        d.Whead, d.bhead = n["Whead"].data_ptr(), n["bhead"].data_ptr()
        d.sc_agent, d.sc_local = self.sc_agent_i32.data_ptr(), self.sc_local_i32.data_ptr()
        d.log_sig_min, d.log_sig_max, d.scale, d.bias = LOG_SIG_MIN, self.log_sig_max, self.scale, self.bias
        for k, t in scratch.items():
            setattr(d, k, t.data_ptr())
        d.flags = la.ACTOR_LAYER_BY_LAYER if self.layer_by_layer else 0
        tiled = None
        if not self.layer_by_layer and H % 16 == 0:
            tiled = dict(W1=la.tile_weights(n["W1"]), Wh=[la.tile_weights(w) for w in n["Wh"]],
                         Whead=la.tile_weights(n["Whead"]))
            tiled["ptrs"] = (C.c_void_p * nh)(*[w.data_ptr() for w in tiled["Wh"]])
            d.W1_tiled, d.Wh_tiled, d.Whead_tiled = tiled["W1"].data_ptr(), tiled["ptrs"], \
                tiled["Whead"].data_ptr()
        self._desc = (d, scratch, Wh, bh, n, tiled)        # keeps every pointer alive
        return d

    out_ring, _ring, _ring_i = 0, None, 0       # set out_ring = 6 for graph-replayed environment steps

    def reset_output_ring(self):
        self._ring_i = 0

    def _select_action_one_call(self, state, eval_mode, eps, out=None):
        """TrainerRPC.choose_action for all agents in ONE library call (aomarl_actor_forward): the
        same kernels as _native_head + policy_sample, issued from C."""
        from . import libaomarl as la
        import ctypes as C
        nenv = state.shape[0]
        d = self._actor_desc(nenv)
        self._draws += 1
        if self.out_ring:
            # fixed output buffers, cycled: an action tensor stays valid for out_ring - 1 further calls (graph
            # replay of the environment step needs stable addresses, see VecAoEnv.OUT_RING)
            if self._ring is None or self._ring[0][0].shape[0] != nenv:
                self._ring = [(torch.empty(nenv, self.layout.action_dim, dtype=torch.float32, device=self.device),
                               torch.empty(nenv, self.layout.action_dim, dtype=torch.float32, device=self.device))
                              for _ in range(self.out_ring)]
                self._ring_i = 0
            a, m = self._ring[self._ring_i]
            self._ring_i = (self._ring_i + 1) % self.out_ring
        else:
            a = torch.empty(nenv, self.layout.action_dim, dtype=torch.float32, device=self.device)
            m = torch.empty_like(a)
        if out is not None:                 # the caller's buffer for what is returned first (a trajectory's row)
            if out.shape != a.shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != a.device:
                raise ValueError("select_action: out must be a contiguous float32 [nenv, action_dim] tensor on the policy's device")
            if eval_mode:
                m = out
            else:
                a = out
        if eps is not None:
            eps = eps.to(torch.float32).contiguous()
        la.check(la.load().aomarl_actor_forward(
                C.byref(d), state.data_ptr(), eps.data_ptr() if eps is not None else None,
                self.seed & 0xFFFFFFFF, self._draws & 0xFFFFFFFF, a.data_ptr(), m.data_ptr(),
                la.raw_stream(self.device)))
        return (m if eval_mode else a), m

    def forward(self, state):
        if self.use_native:
            head = self._native_head(state)
            mean = head[:, :, :self.act_max]
            log_std = head[:, :, self.act_max:].clamp(LOG_SIG_MIN, self.log_sig_max)
            return mean, log_std
        x = self.split_states(state.to(torch.float32))
        x = torch.relu(torch.baddbmm(self.b1, x, self.W1))
        for W, b in zip(self.Wh, self.bh):
            x = torch.relu(torch.baddbmm(b, x, W))
        mean = torch.baddbmm(self.bm, x, self.Wm)
        log_std = torch.baddbmm(self.bs, x, self.Ws).clamp(LOG_SIG_MIN, self.log_sig_max)
        return mean, log_std

    def _assemble(self, per_agent):
        # [A, nenv, act_max] -> [nenv, action_dim]
        return per_agent[self.sc_agent, :, self.sc_local].T.contiguous()

    def select_action(self, state, eval_mode=False, eps=None, out=None):
        """See _select_action: the one-call native path goes straight to the library (no autograd context to
        enter -- nothing there is a torch operation -- and a host-bound step notices the 3 us).  out: where the
        returned action goes (no copy on the one-call path)."""
        if self.use_native and self.native_forward and state.dtype == torch.float32 and \
                state.dim() == 2 and state.shape[1] == self.layout.state_dim:
            return self._select_action_one_call(state if state.is_contiguous() else state.contiguous(), eval_mode, eps, out)
        a, m = self._select_action(state, eval_mode, eps)
        if out is not None:
            out.copy_(a)
            a = out
        return a, m

    @torch.no_grad()
    def _select_action(self, state, eval_mode=False, eps=None):
        """(action, mean), both [nenv, action_dim] in [-1, 1]*scale+bias.  A normal sample is
        always drawn, like the reference does even in eval mode (model_rpc.py:137-144).
        On the GPU the whole tail (clamp, exp, sample, tanh, scale, scatter into the global action
        vector) is one kernel with its own counter-based normals (Philox keyed by this policy's
        seed and draw count); `eps` [nenv, action_dim] overrides the draws."""
        if self.use_native and self.native_forward and state.dtype == torch.float32 and \
                state.dim() == 2 and state.shape[1] == self.layout.state_dim:
            return self._select_action_one_call(state.contiguous(), eval_mode, eps)
        if self.use_native:
            from . import libaomarl as la
            head = self._native_head(state)
            self._draws += 1
            a, m = la.policy_sample(head, self.act_max, self.sc_agent_i32, self.sc_local_i32,
                                    LOG_SIG_MIN, self.log_sig_max, self.scale, self.bias, self.seed,
                                    self._draws, eps=eps)
            return (m if eval_mode else a), m
        mean, log_std = self.forward(state)
        if eps is None:
            e = torch.randn(mean.shape, generator=self.gen, device=self.device)
        else:       # [nenv, action_dim] -> per-agent layout
            e = torch.zeros_like(mean)
            e[self.sc_agent, :, self.sc_local] = eps.T
        x_t = mean + log_std.exp() * e
        action = torch.tanh(x_t) * self.scale + self.bias
        mu = torch.tanh(mean) * self.scale + self.bias
        a, m = self._assemble(action), self._assemble(mu)
        return (m if eval_mode else a), m


class BatchedQNetwork(object):
    """Twin-Q critic forward for all agents (model_rpc.py:22-69), same stacking scheme.

    The reference passes `hidden_size_critic` as a one-element LIST (GlobalConfig.py:36), which
    takes QNetwork's list branch: len(hidden_dim) - 1 = 0 extra hidden layers, i.e.
    Linear(in+act, 256) - ReLU - Linear(256, 1) whatever `num_layers_critic` says.  That is the
    default here (`extra_hidden=0`)."""

    def __init__(self, layout, hidden=256, extra_hidden=0, seed=4321, device="cuda:0"):
        self.layout, self.device = layout, torch.device(device)
        ins, acts = layout.state_shapes(), layout.action_shapes()
        A = layout.n_agents
        self.in_max, self.act_max = max(ins), max(acts)
        g = torch.Generator().manual_seed(seed)
        dev = self.device

        def stack(fan_in_list, fan_out, pad_in):
            W = torch.zeros(A, pad_in, fan_out)
            for i, fi in enumerate(fan_in_list):
                W[i, :fi] = _xavier_uniform(g, fan_out, fi).T
            return W.to(dev)

        self.q = []
        for _ in range(2):
            # input = [state (padded to in_max) | action (padded to act_max)]
            Win = torch.zeros(A, self.in_max + self.act_max, hidden)
            for i in range(A):
                w = _xavier_uniform(g, hidden, ins[i] + acts[i]).T
                Win[i, :ins[i]] = w[:ins[i]]
                Win[i, self.in_max:self.in_max + acts[i]] = w[ins[i]:]
            hid = [stack([hidden] * A, hidden, hidden) for _ in range(extra_hidden)]
            out = stack([hidden] * A, 1, hidden)
            self.q.append((Win.to(dev), hid, out))

    @torch.no_grad()
    def forward(self, split_state, split_action):
        """split_state [A, n, in_max], split_action [A, n, act_max] -> (q1, q2) each [A, n, 1]."""
        x = torch.cat([split_state, split_action], dim=2)
        res = []
        for Win, hid, out in self.q:
            h = torch.relu(torch.bmm(x, Win))
            for W in hid:
                h = torch.relu(torch.bmm(h, W))
            res.append(torch.bmm(h, out))
        return res[0], res[1]
