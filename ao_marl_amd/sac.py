"""On-device replay memory + batched multi-agent SAC update + the training episode loop.

The reference trains one SAC agent per OS process (torch RPC): the master pushes each agent's
(s, a, r, s', mask) into a per-agent master memory during the episode, and at the end of the
episode every agent runs `updates_per_episode_rpc` updates, trickling one master transition into
its own `ReplayMemory` per update (train_rpc.py:734-780, 1086-1133; replay_memory_rpc.py:5-29).
Here all agents are slices of stacked parameter tensors (same scheme as agents.py), the replay
memory is one ring buffer in HBM holding the *global* state once per transition (every agent's
state is an index gather of it), and one update step = one autograd pass over all agents: the
total loss is the sum of the per-agent losses, the parameter sets are disjoint, and Adam is
element-wise, so each agent receives exactly the update the reference computes for it.

Restated reference logic (cited per method):
  SAC.get_bellman_backup / update_critic / calculate_q_loss     train_rpc.py:985-1038
  SAC.update_actor / calculate_policy_loss                       train_rpc.py:1044-1064
  SAC.update_alpha, initialise_alpha                             train_rpc.py:1070-1084, 838-854
  SAC.update_parameters_sac                                      train_rpc.py:1086-1133
  GaussianPolicy.sample (log-prob with tanh correction)          algorithms_rpc/model_rpc.py:133-164
  QNetwork                                                       algorithms_rpc/model_rpc.py:17-69
  soft_update / hard_update                                      algorithms_rpc/utils.py:24-30
  TrainerRPC.episode / manage_memory / save_model                train_rpc.py:503-547, 734-757, 1140-1161
Pinned by tests/golden/host_sac_update.pt (tools/gen_golden_sac.py runs the reference's own update
methods on the reference's modules).
"""
import math
from collections import OrderedDict

import torch

from .agents import BatchedGaussianPolicy, LOG_SIG_MIN, _xavier_uniform

EPSILON = 1e-5                         # model_rpc.py:8

# defaults of config/parameters_sac.cfg + GlobalConfig.py:52-63
DEFAULT_SAC = dict(alpha=0.2, automatic_entropy_tuning=True, batch_size=256, gamma=0.1,
                   hidden_size_actor=256, hidden_size_critic=256, num_layers_actor=2,
                   lr=3e-4, target_update_interval=1, tau=0.005, memory_size=1000000,
                   gaussian_mu=0.0, gaussian_std=1.0, initialize_last_layer_0=True,
                   LOG_SIG_MAX=2.0, updates_per_episode_rpc=1000,
                   # NOT in the reference (default off = the reference's learner): every agent's rewards divided by their
                   # own standard deviation before they enter its replay memory ("auto": measured once, on the first
                   # training episode behind its closing transient).  Independent learners whose rewards differ by four
                   # orders of magnitude -- -3.5 per step for the tip-tilt agent, -1e-3 for the highest modes of the 40x40
                   # system -- all see targets of order one; what an episode reports (rewards, Strehl) is untouched
                   reward_scale=None)


class BatchedReplay(object):
    """All agents' ReplayMemory objects as one ring buffer of global transitions in HBM.

    One row = (state [state_dim], action [action_dim], reward [n_agents], next_state, mask): the
    per-agent tuples the reference stores (manage_memory, train_rpc.py:734-757) are index gathers
    of it.  Sampling draws an independent index set per agent (each reference agent samples its
    own memory); indices are drawn with replacement on the device (the reference uses
    `random.sample`, without replacement: indistinguishable for batch 256 of >= 1e4 rows)."""

    def __init__(self, state_dim, action_dim, n_agents, capacity, device, seed=0):
        self.capacity, self.device = int(capacity), torch.device(device)
        self._dims = (int(state_dim), int(action_dim), int(n_agents))
        self._buf = None                    # allocated on first use (see _alloc)
        self._mask_ones = True
        self.position, self.size = 0, 0
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def row_bytes(self):
        sd, ad, na = self._dims
        return 4 * (2 * sd + ad + na + 1)

    def _alloc(self):
        """The ring is allocated when the first transition arrives, not at construction: the
        reference's default of 1e6 rows is ~46 GB at the 40x40 windowed layout, far more than a
        rollout-only or short run ever touches."""
        if self._buf is None:
            sd, ad, na = self._dims
            nbytes = self.capacity * self.row_bytes()
            if self.device.type == "cuda":
                total = torch.cuda.get_device_properties(self.device).total_memory
                if nbytes > 0.25 * total:
                    import warnings
                    warnings.warn("replay ring of %d rows needs %.1f GB (%.0f %% of this GPU's memory); "
                                  "pass memory_size to BatchedSAC" %
                                  (self.capacity, nbytes / 1e9, 100.0 * nbytes / total))
            f32 = dict(dtype=torch.float32, device=self.device)
            # masks are 1 for every transition the trainer stores (episodes end by step count,
            # train_rpc.py:514, 745): the column starts as ones and is only written when that changes
            self._buf = dict(state=torch.zeros(self.capacity, sd, **f32),
                             next_state=torch.zeros(self.capacity, sd, **f32),
                             action=torch.zeros(self.capacity, ad, **f32),
                             reward=torch.zeros(self.capacity, na, **f32),
                             mask=torch.ones(self.capacity, 1, **f32))
        return self._buf

    state = property(lambda self: self._alloc()["state"])
    next_state = property(lambda self: self._alloc()["next_state"])
    action = property(lambda self: self._alloc()["action"])
    reward = property(lambda self: self._alloc()["reward"])
    mask = property(lambda self: self._alloc()["mask"])

    def __len__(self):
        return self.size

    def reset(self):
        self.position, self.size = 0, 0

    def push(self, state, action, reward, next_state, mask):
        """Append n transitions ([n, ...] tensors; mask scalar or [n]); oldest rows are overwritten."""
        n = state.shape[0]
        if n > self.capacity:
            state, action, reward, next_state = (t[-self.capacity:] for t in
                                                 (state, action, reward, next_state))
            n = self.capacity
        if not torch.is_tensor(mask):
            mask = None if float(mask) == 1.0 and self._mask_ones else \
                torch.full((n, 1), float(mask), dtype=torch.float32, device=self.device)
        elif mask.numel() == 1:
            mask = mask.reshape(1, 1).expand(n, 1)
        if self.position + n <= self.capacity:
            # the common case: one contiguous block of rows -> plain slice copies
            sl = slice(self.position, self.position + n)
            self.state[sl], self.action[sl] = state, action
            self.reward[sl], self.next_state[sl] = reward, next_state
            if mask is not None:
                self.mask[sl] = mask.reshape(-1, 1)[-n:].to(torch.float32)
                self._mask_ones = False
        else:
            idx = (self.position + torch.arange(n, device=self.device)) % self.capacity
            self.state[idx], self.action[idx] = state, action
            self.reward[idx], self.next_state[idx] = reward, next_state
            if mask is None:
                mask = torch.ones(n, 1, dtype=torch.float32, device=self.device)
            self.mask[idx] = mask.reshape(-1, 1)[-n:].to(torch.float32)
        self.position = (self.position + n) % self.capacity
        self.size = min(self.size + n, self.capacity)

    def rows(self, begin, count):
        """`count` rows starting at logical row `begin` (insertion order while not wrapped)."""
        begin %= self.capacity
        if begin + count <= self.capacity:                   # contiguous: views, no gather
            sl = slice(begin, begin + count)
            return (self.state[sl], self.action[sl], self.reward[sl], self.next_state[sl],
                    self.mask[sl])
        idx = (begin + torch.arange(count, device=self.device)) % self.capacity
        return (self.state[idx], self.action[idx], self.reward[idx], self.next_state[idx],
                self.mask[idx])

    def sample_indices(self, n_agents, batch_size):
        return torch.randint(0, self.size, (n_agents, batch_size), generator=self.gen,
                             device=self.device)


class TrajectoryReplay(object):
    """One episode's transitions WITHOUT a copy per step: the environment writes every state and reward, the policy
    every action, straight into trajectory buffers S [T + 1][nenv][state_dim], A [T][nenv][action_dim],
    R [T][nenv][n_agents]; the delayed-MDP tuple stored at step t >= n (environment/delayed_mdp.py:5-58,
    manage_memory train_rpc.py:734-757: state and action of step t - n, next state and reward of step t, n = delay +
    not modification_online) is then four VIEWS of those buffers n rows of the batch apart.  Offers what
    BatchedSAC.update_parameters reads of a master memory: len(), rows(begin, count), reset().
    Why: on the stepping stream every extra launch joins the control / agent chain of the frame in flight and waits
    tens of microseconds for wave slots beside the frame kernel (0.49 -> 0.65 ms per step with the four slice copies
    and the return accumulation in line, tools/episode_probe.py)."""

    def __init__(self, state_dim, action_dim, n_agents, max_steps, nenv, lag, device):
        f32 = dict(dtype=torch.float32, device=device)
        self.T, self.nenv, self.lag = int(max_steps), int(nenv), int(lag)
        self.S = torch.empty(self.T + 1, nenv, state_dim, **f32)
        self.A = torch.empty(self.T, nenv, action_dim, **f32)
        self.R = torch.empty(self.T, nenv, n_agents, **f32)
        self._ones = torch.ones(self.T * nenv, 1, **f32)
        self.t = 0                          # steps taken

    def __len__(self):
        return max(0, self.t - self.lag) * self.nenv

    def reset(self):
        self.t = 0

    def rows(self, begin, count):
        n, k = self.nenv, self.lag * self.nenv
        flat = lambda x: x.reshape(-1, x.shape[-1])      # noqa: E731
        S, A, R = flat(self.S), flat(self.A), flat(self.R)
        if begin < 0 or begin + count > len(self):
            raise IndexError("rows [%d, %d) of %d" % (begin, begin + count, len(self)))
        return (S[begin:begin + count], A[begin:begin + count], R[begin + k:begin + k + count],
                S[begin + k:begin + k + count], self._ones[begin:begin + count])


class _StackedLinearHip(torch.autograd.Function):
    """act(x W + b) for all agents at once on the library's batched GEMM (aomarl_gemm_batched):
    x [A, B, in], W [A, in, out], b [A, 1, out].  Backward: dx = dy W^T, dW = x^T dy, db = sum dy,
    the same three products autograd derives for torch.baddbmm."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        from . import libaomarl as L
        x, W = x.contiguous(), W.contiguous()
        y = L.gemm_batched(x, W, False, True, bias=b.reshape(b.shape[0], -1), relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, W, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import libaomarl as L
        x, W, y = ctx.saved_tensors
        dy = dy * (y > 0) if ctx.relu else dy.contiguous()
        dx = L.gemm_batched(dy, W, False, False) if ctx.needs_input_grad[0] else None
        dW = L.gemm_batched(x, dy, True, True) if ctx.needs_input_grad[1] else None
        db = dy.sum(dim=1, keepdim=True) if ctx.needs_input_grad[2] else None
        return dx, dW, db, None


def stacked_linear(x, W, b, relu):
    """One layer of every agent's network.  On the GPU: the HIP batched GEMM (rocBLAS / hipBLASLt
    pick a 256x256 macro-tile for these 14-matrix batches and run them on 14 CUs); on the CPU (the
    golden tests of the update rule) plain torch."""
    if x.is_cuda:
        return _StackedLinearHip.apply(x, W, b, relu)
    y = torch.baddbmm(b, x, W)
    return torch.relu(y) if relu else y


def flat_layout(A, I, Na, H, Hc, L):
    """Offsets (floats) of the stacked tensors inside the flat policy / critic buffers -- the
    layout aomarl_sac_layout (include/aomarl.h) prescribes: every tensor starts on a multiple of 4
    floats.  Returns (policy_off [2L + 2], policy_len, critic_off [4], critic_len)."""
    def lay(sizes):
        off, o = [], 0
        for n in sizes:
            off.append(o)
            o = (o + n + 3) // 4 * 4
        return off, o
    po, plen = lay([A * I * H, A * H] + [A * H * H, A * H] * (L - 1) + [A * H * 2 * Na, A * 2 * Na])
    co, clen = lay([A * (I + Na) * 2 * Hc, A * 2 * Hc, A * 2 * Hc, A * 2])
    return po, plen, co, clen


class NativeUpdater(object):
    """Handle on the library's SAC update (aomarl_sac_create / aomarl_sac_update) for one batch
    size, working in place on a BatchedSAC's flat parameter buffers.  Adam moments and the
    gradients of the last update are tensors of this object (policy_grad, critic_grad, la_grad)."""

    def __init__(self, sac, batch):
        import ctypes as C
        import numpy as np
        from . import libaomarl as L
        self.L, self.sac, self.batch = L, sac, int(batch)
        lib = L.load()
        p, lay = sac.policy, sac.layout
        A, dev = sac.A, sac.device
        sg = p.gather.cpu().numpy().astype(np.int32)
        sg[sg >= lay.state_dim] = -1
        ag = np.full((A, sac.act_max), -1, dtype=np.int32)
        for i, (w, (lo, hi)) in enumerate(lay.action_slices.items()):
            ag[i, :hi - lo] = np.arange(lo, hi)
        nact = np.asarray(sac.acts, dtype=np.int32)
        te = sac.target_entropy.reshape(-1).cpu().numpy().astype(np.float32)
        self._keep = (sg, ag, nact, te)

        def z(n):
            return torch.zeros(n, dtype=torch.float32, device=dev)
        np_, nc = sac._pflat.numel(), sac._cflat.numel()
        self.policy_m, self.policy_v, self.policy_grad = z(np_), z(np_), z(np_)
        self.critic_m, self.critic_v, self.critic_grad = z(nc), z(nc), z(nc)
        self.la_m, self.la_v, self.la_grad = z(A), z(A), z(A)
        d = L.SacDesc()
        d.n_agents, d.batch, d.in_max, d.act_max = A, self.batch, sac.in_max, sac.act_max
        d.hidden, d.hidden_critic, d.n_hidden = p.H, sac.Hc, p.L
        d.state_dim, d.action_dim = lay.state_dim, lay.action_dim
        d.state_gather, d.action_gather = sg.ctypes.data, ag.ctypes.data
        d.n_act, d.target_entropy = nact.ctypes.data, te.ctypes.data
        d.gamma, d.tau, d.lr = sac.gamma, sac.tau, sac.lr
        d.beta1, d.beta2, d.adam_eps = 0.9, 0.999, 1e-8              # torch.optim.Adam defaults
        d.log_sig_min, d.log_sig_max = LOG_SIG_MIN, p.log_sig_max
        d.action_scale, d.action_bias = p.scale, p.bias
        d.policy, d.policy_m, d.policy_v, d.policy_grad = [t.data_ptr() for t in (
                sac._pflat, self.policy_m, self.policy_v, self.policy_grad)]
        d.critic, d.critic_m, d.critic_v, d.critic_grad = [t.data_ptr() for t in (
                sac._cflat, self.critic_m, self.critic_v, self.critic_grad)]
        d.critic_target = sac._ctflat.data_ptr()
        d.log_alpha, d.log_alpha_m, d.log_alpha_v, d.log_alpha_grad = [t.data_ptr() for t in (
                sac.log_alpha, self.la_m, self.la_v, self.la_grad)]
        d.alpha = sac.alpha.data_ptr()
        po = (C.c_longlong * (2 * p.L + 2))()
        co = (C.c_longlong * 4)()
        pl, cl = C.c_longlong(), C.c_longlong()
        L.check(lib.aomarl_sac_layout(C.byref(d), po, co, C.byref(pl), C.byref(cl)))
        if (list(po), pl.value, list(co), cl.value) != (sac._poff, np_, sac._coff, nc):
            raise L.AomarlError("flat parameter layout of the binding differs from the library's")
        h = C.c_void_p()
        L.check(lib.aomarl_sac_create(C.byref(d), C.byref(h)))
        self.handle, self.lib = h, lib
        self.losses = torch.zeros(5, A, dtype=torch.float32, device=dev)

    def update(self, idx=None, eps_next=None, eps_pi=None):
        """One update of every agent on rows of sac.memory (idx [A, B] int64, default: drawn in the
        kernel); returns the [5, A] loss log (q1, q2, policy, alpha, alpha value)."""
        sac, m = self.sac, self.sac.memory
        step = sac.total_update + 1
        flags = (1 if step % sac.target_update_interval == 0 else 0) | \
                (2 if sac.automatic_entropy_tuning else 0)
        for t in (idx, eps_next, eps_pi):
            assert t is None or (t.is_contiguous() and t.is_cuda)
        assert idx is None or (idx.dtype == torch.int64 and tuple(idx.shape) == (sac.A, self.batch))

        def ptr(t):
            return t.data_ptr() if t is not None else None
        self.L.check(self.lib.aomarl_sac_update(
                self.handle, m.state.data_ptr(), m.next_state.data_ptr(), m.action.data_ptr(),
                m.reward.data_ptr(), m.mask.data_ptr(), len(m), ptr(idx), ptr(eps_next), ptr(eps_pi),
                sac.seed & 0xFFFFFFFF, step & 0xFFFFFFFF, step, flags, self.losses.data_ptr(),
                self.L._stream_of(m.state)))
        return self.losses

    def __del__(self):
        try:
            self.lib.aomarl_sac_destroy(self.handle)
        except Exception:
            pass


class BatchedSAC(object):
    """Soft actor-critic for all agents of an AgentLayout at once.

    Parameters live in two flat buffers (policy, critic; plus the target critic) laid out as
    aomarl_sac_layout prescribes; the per-tensor attributes (policy.W1 ... , critic[k]["Win"] ...)
    are views of them.  On the GPU `update_from_memory` runs the library's hand-written update
    (NativeUpdater); `update` is the torch-autograd statement of the same rule (the CPU golden
    tests against the reference's update run on it, and the GPU tests check the two against each
    other)."""

    def __init__(self, layout, config=None, seed=1234, device="cuda:0", memory_size=None, native=None):
        cfg = dict(DEFAULT_SAC)
        cfg.update(config or {})
        self.cfg, self.layout, self.device = cfg, layout, torch.device(device)
        self.gamma, self.tau, self.lr = cfg["gamma"], cfg["tau"], cfg["lr"]
        self.target_update_interval = cfg["target_update_interval"]
        self.automatic_entropy_tuning = bool(cfg["automatic_entropy_tuning"])
        A = layout.n_agents
        self.A = A
        self.policy = BatchedGaussianPolicy(
                layout, hidden=cfg["hidden_size_actor"], num_layers=cfg["num_layers_actor"],
                log_sig_max=cfg["LOG_SIG_MAX"], action_scale=cfg["gaussian_std"],
                action_bias=cfg["gaussian_mu"], last_layer_zero=cfg["initialize_last_layer_0"],
                seed=seed, device=device)
        p = self.policy
        self.in_max, self.act_max = p.in_max, p.act_max
        ins, acts = layout.state_shapes(), layout.action_shapes()
        self.ins, self.acts = ins, acts
        dev = self.device
        am = torch.zeros(A, 1, self.act_max, device=dev)
        for i, na in enumerate(acts):
            am[i, 0, :na] = 1.0
        self.act_mask = am
        # ---- critic: Linear(in+act, H) - ReLU - Linear(H, 1), twice (the reference passes
        #      hidden_size_critic as a one-element list: no extra hidden layers, GlobalConfig.py:36)
        H = cfg["hidden_size_critic"]
        g = torch.Generator().manual_seed(seed + 7919)
        self.critic = []
        for _ in range(2):
            Win = torch.zeros(A, self.in_max + self.act_max, H)
            Wout = torch.zeros(A, H, 1)
            for i in range(A):
                w = _xavier_uniform(g, H, ins[i] + acts[i]).T
                Win[i, :ins[i]] = w[:ins[i]]
                Win[i, self.in_max:self.in_max + acts[i]] = w[ins[i]:]
                Wout[i] = _xavier_uniform(g, 1, H).T
            self.critic.append(dict(Win=Win.to(dev), bin=torch.zeros(A, 1, H, device=dev),
                                    Wout=Wout.to(dev), bout=torch.zeros(A, 1, 1, device=dev)))
        self.Hc, self.seed = H, int(seed)
        self._flatten_parameters()
        # ---- entropy temperature (initialise_alpha)
        self.target_entropy = -torch.tensor([float(a) for a in acts], device=dev).reshape(A, 1, 1)
        self.log_alpha = torch.zeros(A, 1, 1, device=dev)
        self.alpha = torch.full((A, 1, 1), float(cfg["alpha"]), device=dev)
        self.native = (dev.type == "cuda") if native is None else bool(native)
        self._updaters = {}
        # ---- optimisers (Adam is element-wise: one optimiser over stacked tensors == one per agent)
        for t in self._policy_params() + self._critic_params() + [self.log_alpha]:
            t.requires_grad_(True)
        self.policy_optim = torch.optim.Adam(self._policy_params(), lr=self.lr)
        self.critic_optim = torch.optim.Adam(self._critic_params(), lr=self.lr)
        self.alpha_optim = torch.optim.Adam([self.log_alpha], lr=self.lr)
        self.memory = BatchedReplay(layout.state_dim, layout.action_dim, A,
                                    memory_size if memory_size is not None else cfg["memory_size"],
                                    device, seed=seed)
        self.gen = torch.Generator(device=dev).manual_seed(seed + 1)
        self.total_update = 0
        self.last_losses = None
        self._started_torch = False
        self.keep_grads, self.grad_log = False, None    # tests: gradients of the torch update

    # ------------------------------------------------------------------ parameters
    def _flatten_parameters(self):
        """Move the freshly initialised tensors into the flat buffers and replace them by views."""
        p, A, dev = self.policy, self.A, self.device
        I, Na, H, Hc, L = self.in_max, self.act_max, p.H, self.Hc, p.L
        po, plen, co, clen = flat_layout(A, I, Na, H, Hc, L)
        self._poff, self._coff = po, co
        pf = torch.zeros(plen, dtype=torch.float32, device=dev)

        def view(flat, off, *shape):
            n = 1
            for k in shape:
                n *= k
            return flat[off:off + n].view(*shape)

        def adopt(dst, src):
            dst.copy_(src)
            return dst
        p.W1 = adopt(view(pf, po[0], A, I, H), p.W1)
        p.b1 = adopt(view(pf, po[1], A, 1, H), p.b1)
        p.Wh = [adopt(view(pf, po[2 * l], A, H, H), p.Wh[l - 1]) for l in range(1, L)]
        p.bh = [adopt(view(pf, po[2 * l + 1], A, 1, H), p.bh[l - 1]) for l in range(1, L)]
        whead, bhead = view(pf, po[2 * L], A, H, 2 * Na), view(pf, po[2 * L + 1], A, 1, 2 * Na)
        p.Wm, p.Ws = adopt(whead[:, :, :Na], p.Wm), adopt(whead[:, :, Na:], p.Ws)
        p.bm, p.bs = adopt(bhead[:, :, :Na], p.bm), adopt(bhead[:, :, Na:], p.bs)
        p._native = None
        cf = torch.zeros(clen, dtype=torch.float32, device=dev)

        def critic_views(flat):
            win, bin_ = view(flat, co[0], A, I + Na, 2 * Hc), view(flat, co[1], A, 1, 2 * Hc)
            wout, bout = view(flat, co[2], A, 2, Hc), view(flat, co[3], A, 2)
            return [dict(Win=win[:, :, k * Hc:(k + 1) * Hc], bin=bin_[:, :, k * Hc:(k + 1) * Hc],
                         Wout=wout[:, k, :].unsqueeze(2), bout=bout[:, k].reshape(A, 1, 1))
                    for k in range(2)]
        new = critic_views(cf)
        for qn, q in zip(new, self.critic):
            for k in qn:
                qn[k].copy_(q[k])
        self.critic = new
        ctf = cf.clone()                                                    # hard_update
        self.critic_target = critic_views(ctf)
        self._pflat, self._cflat, self._ctflat = pf, cf, ctf

    def updater(self, batch_size):
        """The native update for this batch size (created on first use)."""
        if not self.native:
            raise RuntimeError("this BatchedSAC was built with native=False")
        if self._started_torch:
            raise RuntimeError("the torch-autograd update already ran on this object: its Adam "
                               "state and the library's are separate")
        u = self._updaters.get(int(batch_size))
        if u is None:
            u = self._updaters[int(batch_size)] = NativeUpdater(self, batch_size)
            if len(self._updaters) > 1:
                raise RuntimeError("one batch size per BatchedSAC on the native path (the Adam "
                                   "moments belong to the updater)")
        return u

    def update_from_memory(self, batch_size=None, idx=None, eps_next=None, eps_pi=None):
        """One update of every agent on a batch drawn from self.memory: the library's update on the
        GPU, the torch statement of it otherwise."""
        batch_size = batch_size or self.cfg["batch_size"]
        if self.native:
            u = self.updater(batch_size)
            lo = u.update(idx, eps_next, eps_pi)
            self.total_update += 1
            self.policy._native = None            # inference copies of the weights are stale
            self.last_losses = dict(q1=lo[0], q2=lo[1], policy=lo[2], alpha=lo[3], alpha_value=lo[4])
            return self.last_losses
        return self.update(*self.batch_from_memory(batch_size, idx), eps_next=eps_next, eps_pi=eps_pi)

    # ------------------------------------------------------------------ learners on several ranks
    def _learner_state(self):
        """Every tensor that defines this learner: flat parameter buffers, temperature, Adam moments
        (the library's when the native update ran, torch.optim's otherwise)."""
        ts = [self._pflat, self._cflat, self._ctflat, self.log_alpha.data]
        for u in self._updaters.values():
            ts += [u.policy_m, u.policy_v, u.critic_m, u.critic_v, u.la_m, u.la_v]
        for opt in (self.policy_optim, self.critic_optim, self.alpha_optim):
            for st in opt.state.values():
                ts += [st[k] for k in ("exp_avg", "exp_avg_sq") if k in st]
        return ts

    def sync_learners(self, init=False):
        """One learner out of the ranks' learners.  The reference has ONE central learner per agent fed by
        every worker (train_rpc.py:759-781: the episode's replay goes to the agent process).  Here every
        rank updates on the transitions of its own environments; `init=True` broadcasts rank 0's weights
        so that all start equal, and after each episode's updates the parameters, the temperature and the
        Adam moments are AVERAGED over the ranks (one all-reduce of the flat buffers per episode, ~40 MB
        for the production layout: periodic parameter averaging, the cheap form of a data-parallel
        learner; a per-update gradient all-reduce would put 0.5 ms of ring time into a 0.45 ms update).
        No-op without a process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return False
        world = dist.get_world_size()
        with torch.no_grad():
            for t in self._learner_state():
                if init:
                    dist.broadcast(t, src=0)
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                    t.div_(world)
            if self.automatic_entropy_tuning:
                self.alpha.copy_(self.log_alpha.data.exp())
        self.policy._native = None                # inference copies of the weights are stale
        return True

    def _policy_params(self):
        p = self.policy
        return [p.W1, p.b1] + list(p.Wh) + list(p.bh) + [p.Wm, p.bm, p.Ws, p.bs]

    def _critic_params(self):
        return [q[k] for q in self.critic for k in ("Win", "bin", "Wout", "bout")]

    def load_reference_agent(self, i, policy_sd=None, critic_sd=None, critic_target_sd=None):
        """Load reference state_dicts (train_rpc.py:1155-1161: actor 'model_state_dict', critic
        `critic.state_dict()`) into agent slot i."""
        if policy_sd is not None:
            self.policy.load_agent(i, policy_sd)
        ni, na = self.ins[i], self.acts[i]

        def put(qs, sd):
            with torch.no_grad():
                for k, q in enumerate(qs):
                    w = sd["Q%d_input.weight" % (k + 1)].to(self.device)       # [H, ni + na]
                    q["Win"][i].zero_()
                    q["Win"][i, :ni] = w[:, :ni].T
                    q["Win"][i, self.in_max:self.in_max + na] = w[:, ni:].T
                    q["bin"][i, 0] = sd["Q%d_input.bias" % (k + 1)].to(self.device)
                    q["Wout"][i] = sd["Q%d_output.weight" % (k + 1)].to(self.device).T
                    q["bout"][i, 0] = sd["Q%d_output.bias" % (k + 1)].to(self.device)
        if critic_sd is not None:
            put(self.critic, critic_sd)
            put(self.critic_target, critic_target_sd if critic_target_sd is not None else critic_sd)

    def load_model(self, i, actor_path, critic_path=None, map_location="cpu"):
        """Load agent slot i from checkpoint files, accepting the three actor layouts the reference's
        loader accepts (src/error_budget/sac/sac.py:187-240): a dict with `alpha` / `log_alpha` /
        `target_entropy` / `model_state_dict`; the trainer's {`worker_id`, `models_controlled`,
        `model_state_dict`} (train_rpc.py:1155-1161); or the bare policy state_dict.  The critic
        file is `critic.state_dict()` (the target critic starts as its copy)."""
        # all three layouts are plain containers of tensors / numbers: no pickled code is accepted
        ck = torch.load(actor_path, map_location=map_location, weights_only=True)
        sd = None
        if critic_path is not None:
            sd = torch.load(critic_path, map_location=map_location, weights_only=True)
        has_alpha = isinstance(ck, dict) and "alpha" in ck
        if has_alpha and self._updaters:
            # checked before anything is written: a rejected load leaves the object as it was
            raise RuntimeError("load checkpoints before the first update (the native updater "
                               "holds a copy of the target entropies)")
        if has_alpha:
            self.policy.load_agent(i, ck["model_state_dict"])
            with torch.no_grad():
                self.alpha[i] = float(torch.as_tensor(ck["alpha"]).detach().reshape(-1)[0])
                self.log_alpha[i] = float(torch.as_tensor(ck["log_alpha"]).detach().reshape(-1)[0])
                # the reference rebuilds it as -prod(-target_entropy): the stored scalar itself
                self.target_entropy[i] = float(torch.as_tensor(ck["target_entropy"]).reshape(-1)[0])
        elif isinstance(ck, dict) and "worker_id" in ck:
            self.policy.load_agent(i, ck["model_state_dict"])
        else:
            self.policy.load_agent(i, ck)
        if sd is not None:
            self.load_reference_agent(i, critic_sd=sd)

    def export_agent(self, i, target=False):
        """(actor state_dict, critic state_dict) of agent i in the reference's layout
        (target=True: the target critic instead of the critic)."""
        p, ni, na = self.policy, self.ins[i], self.acts[i]
        actor = OrderedDict()
        actor["linear1.weight"] = p.W1[i, :ni].detach().T.contiguous().cpu()
        actor["linear1.bias"] = p.b1[i, 0].detach().cpu().clone()
        for j, (W, b) in enumerate(zip(p.Wh, p.bh)):
            actor["hidden.%d.weight" % j] = W[i].detach().T.contiguous().cpu()
            actor["hidden.%d.bias" % j] = b[i, 0].detach().cpu().clone()
        actor["mean_linear.weight"] = p.Wm[i, :, :na].detach().T.contiguous().cpu()
        actor["mean_linear.bias"] = p.bm[i, 0, :na].detach().cpu().clone()
        actor["log_std_linear.weight"] = p.Ws[i, :, :na].detach().T.contiguous().cpu()
        actor["log_std_linear.bias"] = p.bs[i, 0, :na].detach().cpu().clone()
        critic = OrderedDict()
        for k, q in enumerate(self.critic_target if target else self.critic):
            w = torch.cat([q["Win"][i, :ni], q["Win"][i, self.in_max:self.in_max + na]], dim=0)
            critic["Q%d_input.weight" % (k + 1)] = w.detach().T.contiguous().cpu()
            critic["Q%d_input.bias" % (k + 1)] = q["bin"][i, 0].detach().cpu().clone()
            critic["Q%d_output.weight" % (k + 1)] = q["Wout"][i].detach().T.contiguous().cpu()
            critic["Q%d_output.bias" % (k + 1)] = q["bout"][i, 0].detach().cpu().clone()
        return actor, critic

    def save_model(self, path_prefix, episode, experiment_name="aomarl"):
        """One actor + one critic file per agent, the reference's container
        (SAC.save_model, train_rpc.py:1140-1161)."""
        paths = []
        for i, (wid, modes) in enumerate(self.layout.agents.items()):
            actor, critic = self.export_agent(i)
            ap = "%s_worker_%d_sac_actor_%s_episode_%d" % (path_prefix, wid, experiment_name, episode)
            cp = "%s_worker_%d_sac_critic_%s_episode_%d" % (path_prefix, wid, experiment_name, episode)
            torch.save({"worker_id": wid, "models_controlled": list(modes),
                        "model_state_dict": actor}, ap)
            torch.save(critic, cp)
            paths.append((ap, cp))
        return paths

    # ------------------------------------------------------------------ networks (autograd path)
    def _policy_forward(self, x):
        p = self.policy
        h = stacked_linear(x, p.W1, p.b1, True)
        for W, b in zip(p.Wh, p.bh):
            h = stacked_linear(h, W, b, True)
        mean = stacked_linear(h, p.Wm, p.bm, False)
        log_std = stacked_linear(h, p.Ws, p.bs, False).clamp(LOG_SIG_MIN, p.log_sig_max)
        return mean, log_std

    def sample(self, x, eps=None):
        """GaussianPolicy.sample for split states x [A, B, in_max] -> (action, log_prob, mean);
        padded action columns are zero and carry no log-probability.  `eps`: the standard-normal
        draws ([A, B, act_max]; default: from this object's generator)."""
        p = self.policy
        mean, log_std = self._policy_forward(x)
        std = log_std.exp()
        if eps is None:
            eps = torch.randn(mean.shape, generator=self.gen, device=self.device)
        x_t = mean + std * eps
        y_t = torch.tanh(x_t)
        action = (y_t * p.scale + p.bias) * self.act_mask
        # Normal(mean, std).log_prob(x_t) - log(scale (1 - y^2) + eps)          model_rpc.py:153-158
        log_prob = -0.5 * eps * eps - log_std - 0.5 * math.log(2.0 * math.pi)
        log_prob = log_prob - torch.log(p.scale * (1.0 - y_t.pow(2).clamp(min=0, max=1)) + EPSILON)
        log_prob = (log_prob * self.act_mask).sum(dim=2, keepdim=True)
        mean_a = (torch.tanh(mean) * p.scale + p.bias) * self.act_mask
        return action, log_prob, mean_a

    @staticmethod
    def _q(qs, x, a):
        xa = torch.cat([x, a], dim=2)
        out = []
        for q in qs:
            h = stacked_linear(xa, q["Win"], q["bin"], True)
            out.append(stacked_linear(h, q["Wout"], q["bout"], False))
        return out[0], out[1]

    # ------------------------------------------------------------------ one update
    def update(self, x, a, r, x2, mask, eps_next=None, eps_pi=None):
        """update_critic -> update_actor -> update_alpha -> soft_update for every agent.
        x, x2 [A, B, in_max]; a [A, B, act_max]; r, mask [A, B, 1].  Returns per-agent losses."""
        if self._updaters:
            raise RuntimeError("the native update already ran on this object: its Adam state and "
                               "torch's are separate")
        self._started_torch = True
        # ---- critic                                                     train_rpc.py:985-1038
        with torch.no_grad():
            a2, logp2, _ = self.sample(x2, eps_next)
            q1t, q2t = self._q(self.critic_target, x2, a2)
            min_q = torch.min(q1t, q2t) - self.alpha * logp2
            target = r + mask * self.gamma * min_q
        q1, q2 = self._q(self.critic, x, a)
        q1_loss = ((q1 - target) ** 2).mean(dim=(1, 2))                 # F.mse_loss per agent
        q2_loss = ((q2 - target) ** 2).mean(dim=(1, 2))
        self.critic_optim.zero_grad(set_to_none=True)
        (q1_loss + q2_loss).sum().backward()
        if self.keep_grads:
            self.grad_log = dict(critic={k: [q[k].grad.clone() for q in self.critic]
                                         for k in ("Win", "bin", "Wout", "bout")})
        self.critic_optim.step()
        # ---- actor                                                      train_rpc.py:1044-1064
        pi, log_pi, _ = self.sample(x, eps_pi)
        q1p, q2p = self._q(self.critic, x, pi)
        policy_loss = (self.alpha * log_pi - torch.min(q1p, q2p)).mean(dim=(1, 2))
        self.policy_optim.zero_grad(set_to_none=True)
        policy_loss.sum().backward()
        if self.keep_grads:
            self.grad_log["policy"] = [t.grad.clone() for t in self._policy_params()]
        self.policy_optim.step()
        self.policy._native = None                # inference copies of the weights are stale
        # ---- temperature                                                train_rpc.py:1070-1084
        if self.automatic_entropy_tuning:
            alpha_loss = -(self.log_alpha * (log_pi + self.target_entropy).detach()).mean(dim=(1, 2))
            self.alpha_optim.zero_grad(set_to_none=True)
            alpha_loss.sum().backward()
            if self.keep_grads:
                self.grad_log["log_alpha"] = self.log_alpha.grad.clone()
            self.alpha_optim.step()
            self.alpha.copy_(self.log_alpha.detach().exp())
        else:
            alpha_loss = torch.zeros(self.A, device=self.device)
        # ---- target network                                             train_rpc.py:1128-1129
        self.total_update += 1
        if self.total_update % self.target_update_interval == 0:
            with torch.no_grad():
                for qt, q in zip(self.critic_target, self.critic):
                    for k in qt:
                        qt[k].mul_(1.0 - self.tau).add_(q[k].detach(), alpha=self.tau)
        self.last_losses = dict(q1=q1_loss.detach(), q2=q2_loss.detach(),
                                policy=policy_loss.detach(), alpha=alpha_loss.detach(),
                                alpha_value=self.alpha.reshape(-1).clone())
        return self.last_losses

    # ------------------------------------------------------------------ replay plumbing
    def split_rewards(self, reward):
        """[n, A] -> [A, n, 1]"""
        return reward.T.unsqueeze(2).contiguous()

    def batch_from_memory(self, batch_size, idx=None):
        """Independent index set per agent -> tensors in the layout `update` takes."""
        m, p = self.memory, self.policy
        if idx is None:
            idx = m.sample_indices(self.A, batch_size)                       # [A, B]
        flat = idx.reshape(-1)

        def gather_state(buf):
            s = buf[flat].reshape(self.A, batch_size, -1)
            s = torch.cat([s, s.new_zeros(self.A, batch_size, 1)], dim=2)    # zero pad column
            return torch.gather(s, 2, p.gather.unsqueeze(1).expand(-1, batch_size, -1))

        x, x2 = gather_state(m.state), gather_state(m.next_state)
        act = m.action[flat].reshape(self.A, batch_size, -1)
        a = torch.gather(act, 2, self._action_gather.unsqueeze(1).expand(-1, batch_size, -1)) \
            * self.act_mask
        r = m.reward[flat].reshape(self.A, batch_size, self.A)
        r = torch.gather(r, 2, torch.arange(self.A, device=self.device).reshape(self.A, 1, 1)
                         .expand(-1, batch_size, -1))
        mask = m.mask[flat].reshape(self.A, batch_size, 1)
        return x, a, r, x2, mask

    @property
    def _action_gather(self):
        if not hasattr(self, "_ag"):
            g = torch.zeros(self.A, self.act_max, dtype=torch.long, device=self.device)
            for i, (w, (lo, hi)) in enumerate(self.layout.action_slices.items()):
                g[i, :hi - lo] = torch.arange(lo, hi, device=self.device)
            self._ag = g
        return self._ag

    _rscale = None

    def _reward_scale(self, master):
        """cfg["reward_scale"]: None (the reference), a number or [n_agents] numbers, or "auto": 1 / std of every
        agent's rewards over the first training episode's transitions (the first tenth -- the loop closing behind the
        reset -- left out), measured once and kept."""
        rs = self.cfg.get("reward_scale")
        if rs is None:
            return None
        if self._rscale is None:
            if isinstance(rs, str):
                if rs != "auto":
                    raise ValueError("reward_scale: None, numbers or 'auto'")
                total = len(master)
                if total < 64:
                    return None
                r = master.rows(total // 10, total - total // 10)[2]
                self._rscale = (1.0 / r.std(dim=0).clamp(min=1e-12)).reshape(1, -1).to(torch.float32)
            else:
                self._rscale = torch.as_tensor(rs, dtype=torch.float32, device=self.device).reshape(1, -1)
        return self._rscale

    def update_parameters(self, master, batch_size=None, n_updates=None):
        """SAC.update_parameters_sac: n_updates updates; before each, the next slice of the
        episode's master memory moves into the agents' memory (1 transition per update in the
        reference, which steps one environment; len(master) / n_updates here)."""
        batch_size = batch_size or self.cfg["batch_size"]
        n_updates = n_updates or self.cfg["updates_per_episode_rpc"]
        total, moved = len(master), 0
        scale = self._reward_scale(master)
        per = -(-total // n_updates) if total else 0
        done = 0
        for _ in range(n_updates):
            if moved < total:
                n = min(per, total - moved)
                rows = master.rows(moved, n)
                if scale is not None:
                    rows = (rows[0], rows[1], rows[2] * scale) + tuple(rows[3:])
                self.memory.push(*rows)
                moved += n
            if len(self.memory) > batch_size:
                self.update_from_memory(batch_size)
                done += 1
        return done


def run_episode(env, sac, max_steps=None, train=True, eval_mode=False, linear_control=False,
                master=None, n_updates=None, batch_size=None, timing=None):
    """TrainerRPC.episode (train=True) / test_episode (train=False), batched over env.nenv
    environments (train_rpc.py:503-547, 549-603).  Returns a dict of device tensors:
    r_total [nenv], r_per_agent [nenv, A], sr_le [nenv], sr_se_mean [nenv] (test episodes: the training episode
    reads the Strehl once, at its end, like the reference's :546-549) (+ updates done).
    timing: a dict that receives `steps_s` / `updates_s` (one extra device synchronisation between the phases)."""
    import time as _time
    t_begin = _time.perf_counter()
    from .env import DelayedMDP
    cfg = env.config_rl
    max_steps = max_steps or cfg["max_steps_per_episode"]
    s = env.reset()
    mdp = DelayedMDP(cfg["delayed_assignment"], cfg["modification_online"])
    r_agents = torch.zeros(env.nenv, env.layout.n_agents, device=env.device)
    sr_se = torch.zeros(env.nenv, device=env.device)
    geo = getattr(env.supervisor, "geo", None) if not train else None
    geo_prev, geo_sq = None, None
    # A training episode on the GPU keeps its transitions in trajectory buffers the step writes into directly
    # (TrajectoryReplay: no launch per step beyond the actor and the environment); episodes with a caller's master
    # memory, on the CPU stand-in or with the integrator alone take the reference's bookkeeping step by step.
    traj = None
    if train and master is None and not linear_control and env.device.type == "cuda" and \
            not getattr(env.supervisor.sim, "graph_step", False):
        traj = TrajectoryReplay(env.layout.state_dim, env.layout.action_dim, env.layout.n_agents, max_steps, env.nenv,
                                mdp._n, env.device)
        traj.S[0].copy_(s)
        s = traj.S[0]
        master = traj
    elif train and master is None:
        master = BatchedReplay(env.layout.state_dim, env.layout.action_dim, env.layout.n_agents,
                               max_steps * env.nenv, env.device)
    for t in range(max_steps):
        if traj is not None:
            if eval_mode or not hasattr(env, "policy_step"):
                a, mu = sac.policy.select_action(s, eval_mode=eval_mode, out=traj.A[t])
                s_next, r, done, _ = env.step(a, out=(traj.S[t + 1], traj.R[t]))
            else:           # choose_action + env_step in one library call, every output straight into the trajectory
                a, s_next, r, done, _ = env.policy_step(sac.policy, s, out=(traj.S[t + 1], traj.R[t]), action_out=traj.A[t])
            traj.t = t + 1
            s = s_next
            continue
        if linear_control:
            a = None
        else:
            a, mu = sac.policy.select_action(s, eval_mode=eval_mode)
        s_next, r, done, _ = env.step(a, linear_control=linear_control)
        if train:
            if mdp.check_update_possibility():                     # manage_memory
                s0, a0, s2 = mdp.credit_assignment()
                master.push(s0, a0, r, s2, float(not done))
            mdp.save(s, a, s_next)                                 # manage_delayed_mdp
        r_agents += r
        if not train:                       # test_episode's per-step short-exposure Strehl (:583-584)
            sr_se += env.supervisor.get_strehl()[:, 0]
        if geo is not None:                 # test_episode: v2m . rtc.get_command(1) per step
            gm = env.supervisor.sim.volts2modes(env.supervisor.get_command(1))
            if geo_prev is not None:
                d = gm - geo_prev
                geo_sq = d * d if geo_sq is None else geo_sq + d * d
            geo_prev = gm
        s = s_next
    if traj is not None:
        r_agents = traj.R[:traj.t].sum(dim=0)
    out = dict(r_total=r_agents.sum(dim=1), r_per_agent=r_agents,
               sr_le=env.supervisor.get_strehl()[:, 1].clone(), sr_se_mean=sr_se / max_steps)
    if geo is not None and geo_sq is not None:
        # divide_rewards_for_agents_geometric (train_rpc.py:381-400): squared frame-to-frame
        # increments of the geometric command's modes, summed over the episode
        out["r_geo_per_agent"] = geo_sq @ env._reward_mat
        out["r_geo_total"] = out["r_geo_per_agent"].sum(dim=1)
        out["sr_le_geo"] = env.supervisor.get_strehl(1)[:, 1].clone()
    # the only collective of the path: every rank sees every environment's return, in global seed
    # order (E floats per rank per episode; no-op in a single-process run)
    from .dist import gather_episode_returns
    out["r_total_all"] = gather_episode_returns(out["r_total"])
    out["sr_le_all"] = gather_episode_returns(out["sr_le"])
    if timing is not None:
        torch.cuda.synchronize(env.device)
        timing["steps_s"] = _time.perf_counter() - t_begin
    if train:
        out["updates"] = sac.update_parameters(master, batch_size=batch_size, n_updates=n_updates)
        master.reset()
        if timing is not None:
            torch.cuda.synchronize(env.device)
            timing["updates_s"] = _time.perf_counter() - t_begin - timing["steps_s"]
    return out


def train_agent(env, sac, n_episodes, max_steps=None, test_every=50, n_updates=None, batch_size=None,
                on_episode=None, throughput=True):
    """TrainerRPC.train_agent (train_rpc.py:452-501), batched: training episodes, every
    `test_every` episodes one RL evaluation and one integrator evaluation on fresh seeds, and after
    EVERY episode (training or test) the simulation moves on to a seed block no environment of no
    rank has seen -- the reference's `self.seed += 1; env.set_sim_seed(self.seed)` (:486-487,
    :495-496) for a batch of environments spread over ranks.  Returns the per-episode log."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    log = []
    # several ranks = several collectors feeding ONE learner (BatchedSAC.sync_learners): equal weights
    # at the start, averaged learner state after every episode's updates
    sac.sync_learners(init=True)
    # throughput (default): the loop bench.py times -- VecAoEnv.throughput_mode: frame pipeline where eligible,
    # residual shortcut, and the next training episode's screens grown beside this one (aomarl_reset_prefetch_*: its
    # seeds are this episode's + one block per rank; an evaluation in between resets on other seeds, the prefetch is
    # dropped and that reset runs in the open).  throughput=False leaves the environment as the caller built it.
    if throughput and hasattr(env, "throughput_mode"):
        env.throughput_mode(reset_prefetch=world)
    for ep in range(int(n_episodes)):
        seed = env.supervisor.current_seed
        out = run_episode(env, sac, max_steps=max_steps, train=True, n_updates=n_updates,
                          batch_size=batch_size)
        sac.sync_learners()
        rec = dict(episode=ep, seed=seed, r_total=float(out["r_total_all"].mean()),
                   sr_le=float(out["sr_le_all"].mean()), updates=out.get("updates", 0),
                   r_total_rank=float(out["r_total"].mean()), sr_le_rank=float(out["sr_le"].mean()))
        if test_every and ep % int(test_every) == 0:
            env.next_seed_block(world)
            rec["test_seed"] = env.supervisor.current_seed
            rl = run_episode(env, sac, max_steps=max_steps, train=False, eval_mode=True)
            # the integrator baseline sees the same atmosphere as the RL evaluation
            # (test_episode("Integrator") follows test_episode("RL") without a new seed, :487-489)
            lin = run_episode(env, sac, max_steps=max_steps, train=False, linear_control=True)
            rec.update(test_r_rl=float(rl["r_total_all"].mean()), test_sr_le_rl=float(rl["sr_le_all"].mean()),
                       test_r_integrator=float(lin["r_total_all"].mean()),
                       test_sr_le_integrator=float(lin["sr_le_all"].mean()),
                       test_sr_se_rl=float(rl["sr_se_mean"].mean()), test_sr_se_integrator=float(lin["sr_se_mean"].mean()),
                       test_r_agents_rl=rl["r_per_agent"].mean(dim=0).tolist(),
                       test_r_agents_integrator=lin["r_per_agent"].mean(dim=0).tolist())
        env.next_seed_block(world)
        log.append(rec)
        if on_episode is not None:
            on_episode(rec)
    return log
