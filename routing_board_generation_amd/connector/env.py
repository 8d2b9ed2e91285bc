"""Connector environment -- API of [3P] jumanji.environments.routing.connector.env.Connector
(jumanji==0.2.2, pinned by the reference at requirements.txt:5), backed by `connector_kernel`.

Call sites in the reference that this class serves:
  routing_board_generation/rl_training/setup_train.py:158   Connector(generator=generator)
  demos/board_generator_demo.py:83-96                       _get_action_mask/_obs_from_grid/_get_extras
  package_evaluation/load_and_test_agents.ipynb:190-219     reset / step loop

Every method accepts a single env (key (2,), State without batch axis) or a batch (key (B, 2),
batched State): the batch form replaces `jax.vmap(env.reset)` / `jax.vmap(env.step)`.

Semantics not pinned by any reference golden (reward constants, discount, termination,
observation encoding, extras) are "defined by this project, modelled on jumanji 0.2.2";
see DESIGN.md.
"""
from typing import Optional, Tuple

import torch

from .. import ops, specs
from ..online_generators.uniform_generator import Generator, UniformRandomGenerator
from ..types import Agent, Observation, State, TimeStep


class DenseRewardFn:
    """[3P] connector/reward.py::DenseRewardFn: `connected_reward` on the step an agent connects,
    `timestep_reward` on every step it starts unconnected."""

    def __init__(self, timestep_reward: float = -0.03, connected_reward: float = 0.1) -> None:
        self.timestep_reward = timestep_reward
        self.connected_reward = connected_reward


def _state_bufs(state: State, batched: bool):
    def b(x):
        return (x if batched else x[None]).contiguous()

    return {"grid": b(state.grid), "step_count": b(state.step_count).reshape(-1),
            "start": b(state.agents.start), "target": b(state.agents.target),
            "position": b(state.agents.position), "key": b(state.key)}


def _make_state(st, batched: bool) -> State:
    B, N = st["start"].shape[0], st["start"].shape[1]
    ids = torch.arange(N, dtype=torch.int32, device=st["start"].device).expand(B, N)
    s = State(grid=st["grid"], step_count=st["step_count"],
              agents=Agent(id=ids, start=st["start"], target=st["target"],
                           position=st["position"]), key=st["key"])
    if batched:
        return s
    return State(grid=s.grid[0], step_count=s.step_count[0],
                 agents=Agent(*[x[0] for x in s.agents]), key=s.key[0])


def _make_timestep(ts, batched: bool) -> TimeStep:
    sel = (lambda x: x) if batched else (lambda x: x[0])
    obs = Observation(grid=sel(ts["obs_grid"]) if "obs_grid" in ts else None,
                      action_mask=sel(ts["action_mask"]), step_count=sel(ts["obs_step_count"]))
    extras = {"num_connections": sel(ts["num_connections"]),
              "ratio_connections": sel(ts["ratio_connections"]),
              "total_path_length": sel(ts["total_path_length"])}
    return TimeStep(step_type=sel(ts["step_type"]), reward=sel(ts["reward"]),
                    discount=sel(ts["discount"]), observation=obs, extras=extras)


def _state_from_leaves(v, ids: torch.Tensor) -> State:
    """v in ops.STATE_FIELDS order (batched)."""
    return State(grid=v[0], step_count=v[1],
                 agents=Agent(id=ids, start=v[2], target=v[3], position=v[4]), key=v[5])


def _timestep_from_leaves(v) -> TimeStep:
    """v in ops.TIMESTEP_FIELDS order (batched); the aggregated reward / discount stand in for
    the per-agent ones when the session produced those (MultiToSingleWrapper)."""
    return TimeStep(step_type=v[0], reward=v[1] if v[1] is not None else v[9],
                    discount=v[2] if v[2] is not None else v[10],
                    observation=Observation(grid=v[3], action_mask=v[4], step_count=v[5]),
                    extras={"num_connections": v[6], "ratio_connections": v[7],
                            "total_path_length": v[8]})


class Connector:
    def __init__(self, generator: Optional[Generator] = None, reward_fn: Optional[DenseRewardFn] = None,
                 time_limit: int = 50, viewer=None, donate: bool = False,
                 chunk_steps: Optional[int] = None) -> None:
        """donate=True (not in jumanji): batched step() advances the State in place and reuses two
        TimeStep buffer sets alternately instead of allocating outputs (ops.ConnectorSession).
        chunk_steps (not in jumanji): batched step() carves its fresh outputs from chunks of this
        many steps (default: up to 16, at most 1 GiB per field).  A tensor of a returned
        State / TimeStep that the caller keeps alive keeps its whole chunk alive (and out of
        reuse): a replay buffer that retains timesteps should .clone() them or pass
        chunk_steps=1."""
        self._generator = generator or UniformRandomGenerator(grid_size=10, num_agents=5)
        self._reward_fn = reward_fn or DenseRewardFn()
        self.time_limit = time_limit
        self.num_agents = self._generator.num_agents
        self.grid_size = self._generator.grid_size
        self._viewer = viewer
        self._auto_reset = False  # switched on by VmapAutoResetWrapper
        self._prefetch_delay = 0  # VmapAutoResetWrapper(prefetch=True): > 0, or None = the
        #                           generator's own schedule (ops.GeneratorSpec.refill)
        self._donate = bool(donate)
        self._chunk_steps = chunk_steps
        self._aggregate = False  # switched on by MultiToSingleWrapper
        self._sessions = {}  # (B, device, wrapper + env parameters) -> ops.ConnectorSession
        self._rollout_epochs = {}  # (B, device) -> ops.EpochStash of random_rollout
        # the generator as the env launches run it (include/rbg_prw.h::rbg_generator); None for a
        # user-defined Python generator, which takes the generic path (_generic_auto_reset)
        spec = getattr(self._generator, "native_spec", None)
        self._gen = spec() if callable(spec) else None
        self._hot = None  # (B, device, session) of the last batched call

    def _configure(self, **attrs) -> None:
        """Wrapper switches (VmapAutoResetWrapper / MultiToSingleWrapper)."""
        for k, v in attrs.items():
            setattr(self, "_" + k, v)
        self._hot = None

    @property
    def unwrapped(self) -> "Connector":
        return self

    @property
    def generator(self) -> Generator:
        return self._generator

    def _session(self, B: int, device: torch.device) -> ops.ConnectorSession:
        """The launch context of batched reset/step calls (ops.ConnectorSession), one per batch
        size / device / wrapper configuration."""
        hot = self._hot
        rf = self._reward_fn
        if (hot is not None and hot[0] == B and hot[1] == device and hot[3] == self.time_limit
                and hot[4] == rf.timestep_reward and hot[5] == rf.connected_reward):
            # (time_limit and the reward function are public attributes a caller may change
            # between calls: they are part of the key, not only of the session table's)
            return hot[2]
        fused_reset = self._auto_reset and self._gen is not None
        delay = self._prefetch_delay
        if delay is None:
            delay = self._gen.refill.get("every", 16) if self._gen is not None else 0
            r = self._gen.refill if self._gen is not None else {}
            if delay and not r.get("serial", True):
                # rolling refills (ops.EpochStash): an entry is requested `depth` episodes before
                # it is needed and becomes usable at most lead * K steps later, so with episodes
                # of at most time_limit steps the period K must not exceed depth * time_limit /
                # lead (measured at time limits 12 and 20: beyond it every reset misses and the
                # stash falls back to in-line refills, 1.6-2x slower); 15 % to spare for episodes
                # that end early.  Much shorter episodes (a trained policy) fall back at run time.
                room = int(r.get("depth", 1) * self.time_limit * 0.85 / max(2, r.get("lead", 2)))
                delay = max(1, min(delay, room))
            elif delay and r.get("depth", 1) == 1:
                # refills in line, one slot: a request is served by the next refill, at most K
                # steps later, and needed an episode later (time limit 12: 15.1 us per step with
                # K = 12, 15.7 with the walk's K = 20, where 40 % of the resets miss)
                delay = max(1, min(delay, self.time_limit))
        k = (B, device, fused_reset, delay, self._donate, self._aggregate,
             self.time_limit,
             self._reward_fn.timestep_reward, self._reward_fn.connected_reward)
        sess = self._sessions.get(k)
        if sess is None:
            while len(self._sessions) >= 4:  # a session pins its current output chunk: keep few
                self._sessions.pop(next(iter(self._sessions))).drain()
            sess = self._sessions[k] = ops.ConnectorSession(
                B, self.grid_size, self.num_agents, device, self.time_limit,
                self._reward_fn.timestep_reward, self._reward_fn.connected_reward,
                auto_reset=fused_reset, refill_every=delay, donate=self._donate,
                aggregate=self._aggregate, gen=self._gen, chunk_steps=self._chunk_steps)
        self._hot = (B, device, sess, self.time_limit, rf.timestep_reward, rf.connected_reward)
        return sess

    # ---------------------------------------------------------------------------------- specs
    def observation_spec(self) -> "specs.Spec[Observation]":
        """[3P] Connector.observation_spec: one env (no batch axis), per-agent views."""
        N, m = self.num_agents, self.grid_size
        return specs.Spec(
            Observation, "ObservationSpec",
            grid=specs.BoundedArray((N, m, m), torch.int32, 0, 3 * N, "grid"),
            action_mask=specs.BoundedArray((N, 5), torch.bool, False, True, "action_mask"),
            step_count=specs.BoundedArray((), torch.int32, 0, self.time_limit, "step_count"))

    def action_spec(self) -> "specs.MultiDiscreteArray":
        """One of NOOP / UP / RIGHT / DOWN / LEFT per agent."""
        return specs.MultiDiscreteArray([5] * self.num_agents, torch.int32, "action")

    def reward_spec(self) -> "specs.Array":
        shape = () if self._aggregate else (self.num_agents,)
        return specs.Array(shape, torch.float32, "reward")

    def discount_spec(self) -> "specs.BoundedArray":
        shape = () if self._aggregate else (self.num_agents,)
        return specs.BoundedArray(shape, torch.float32, 0.0, 1.0, "discount")

    # ---------------------------------------------------------------------------------- reset
    def reset(self, key: torch.Tensor) -> Tuple[State, TimeStep]:
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        if self._gen is not None:
            # fused: generator + State + observation in one launch
            if batched:
                sess = self._session(keys.shape[0], keys.device)
                st, ts = sess.reset(keys)
                return _state_from_leaves(st, sess.agent_ids), _timestep_from_leaves(ts)
            st, ts = ops.connector_reset(keys, self.grid_size, self.num_agents, gen=self._gen)
        else:
            st = _state_bufs(self._generator(keys), True)
            ts = ops.connector_observe(st)
        return _make_state(st, batched), self._single_agent_view(_make_timestep(ts, batched))

    # ---------------------------------------------------------------------------------- step
    def step(self, state: State, action: torch.Tensor) -> Tuple[State, TimeStep]:
        batched = state.grid.dim() == 3
        fused_reset = self._auto_reset and self._gen is not None
        if batched and (fused_reset or not self._auto_reset):
            # next-episode States come from the session's stash when it holds them (same results
            # either way)
            sess = self._session(state.grid.shape[0], state.grid.device)
            ag = state.agents
            if action.dtype is not torch.int32:
                action = action.to(torch.int32)
            so, ts = sess.step((state.grid, state.step_count, ag.start, ag.target, ag.position,
                                state.key), action.contiguous())
            return _state_from_leaves(so, sess.agent_ids), _timestep_from_leaves(ts)
        st = _state_bufs(state, batched)
        act = (action if batched else action[None]).to(torch.int32).contiguous()
        so, ts = ops.connector_step(st, act, self.time_limit, self._reward_fn.timestep_reward,
                                    self._reward_fn.connected_reward, auto_reset=fused_reset,
                                    gen=self._gen)
        if self._auto_reset and not fused_reset:
            so, ts = self._generic_auto_reset(so, ts)
        return _make_state(so, batched), self._single_agent_view(_make_timestep(ts, batched))

    def _single_agent_view(self, ts: TimeStep) -> TimeStep:
        """MultiToSingleWrapper on the paths without a fused aggregate (single envs, generators
        other than ParallelRandomWalk under auto-reset)."""
        if not self._aggregate:
            return ts
        return ts._replace(reward=ts.reward.sum(dim=-1), discount=ts.discount.max(dim=-1).values)

    # ---------------------------------------------------------------------------------- rollout
    def _rollout_epochs_for(self, state: State, record: bool = False):
        if state.grid.dim() != 3:
            raise ValueError("random_rollout expects a batched state")
        if self._auto_reset and self._gen is None:
            raise NotImplementedError(
                "fused rollouts need a generator the launch can run itself (Generator.native_spec); "
                "a user-defined Python generator steps through Connector.step")
        epochs = None
        # (measured per generator, tools/probe_rollout_gen.py: a stash pays in a many-step launch
        # only where running the generator inline stalls it -- GeneratorSpec.refill)
        want = self._gen is not None and self._gen.refill.get(
            "record_stash" if record else "rollout_stash", not record)
        if self._auto_reset and self._prefetch_delay != 0 and want:
            # two-slot next-episode stash, refilled after every launch: also an env that finishes
            # two episodes within the launch finds both successors ready
            B, dev = state.grid.shape[0], state.grid.device
            epochs = self._rollout_epochs.get((B, dev))
            if epochs is None:
                self._rollout_epochs.clear()
                epochs = self._rollout_epochs[(B, dev)] = ops.EpochStash(
                    B, self.grid_size, self.num_agents, dev, every=1, serial=True, depth=2,
                    gen=self._gen)
        return epochs

    def random_rollout(self, state: State, policy_keys: torch.Tensor, t0: int, n_steps: int,
                       want_timestep: bool = False, record: bool = False):
        """n_steps of `step` under the masked-uniform random policy (ops.random_actions) in ONE
        kernel launch: the rollout loop of benchmark_on_random_agent.py:59-102 with jumanji's
        RandomAgent.  Batched states only.  Returns (state, stats, timestep-of-the-last-step or
        None); stats are per-env int32 counters (include/rbg_prw.h::rbg_rollout_stats).
        record=True is the round-2 spelling of random_rollout_record (deprecated): it returns
        that call's (state, actions, trajectory) without the stats, as it did then."""
        if record:
            import warnings

            warnings.warn("random_rollout(record=True) is deprecated: call random_rollout_record, "
                          "which also returns the rollout stats", DeprecationWarning, stacklevel=2)
            so, _, actions, traj = self.random_rollout_record(state, policy_keys, t0, n_steps)
            return so, actions, traj
        epochs = self._rollout_epochs_for(state)
        so, stats, ts = ops.connector_rollout(
            _state_bufs(state, True), policy_keys.contiguous(), t0, n_steps, self.time_limit,
            self._reward_fn.timestep_reward, self._reward_fn.connected_reward,
            auto_reset=self._auto_reset, want_timestep=want_timestep, epochs=epochs,
            gen=self._gen)
        return _make_state(so, True), stats, (_make_timestep(ts, True) if ts is not None else None)

    def random_rollout_record(self, state: State, policy_keys: torch.Tensor, t0: int,
                              n_steps: int):
        """The same launch also RECORDING the trajectory the acting loop collects (what `lax.scan`
        stacks in run_epoch, benchmark_on_random_agent.py:71-86): returns (state, stats, actions
        [n_steps, B, N], TimeStep with a leading time axis) -- entry s is what the s-th
        `step(state, action)` of the loop returns."""
        epochs = self._rollout_epochs_for(state, record=True)
        B, dev = state.grid.shape[0], state.grid.device
        stats = {k: torch.empty((B,), dtype=torch.int32, device=dev) for k in ops.ROLLOUT_STATS}
        so, stats, traj = ops.connector_rollout_record(
            _state_bufs(state, True), policy_keys.contiguous(), t0, n_steps, self.time_limit,
            self._reward_fn.timestep_reward, self._reward_fn.connected_reward,
            auto_reset=self._auto_reset, epochs=epochs, gen=self._gen, stats=stats)
        return _make_state(so, True), stats, traj["action"], _make_timestep(traj, True)

    def _generic_auto_reset(self, so, ts):
        """VmapAutoResetWrapper for a user-defined Python generator (Generator.native_spec() is
        None): regenerate every env from split(state.key)[0] and select where the timestep is
        LAST -- what vmap(lax.cond) does in the reference for every generator.  The generators
        this package ships never come here."""
        from ..random import split

        done = ts["step_type"] == 2
        new_keys = split(so["key"], 2)[:, 0].contiguous()
        fresh = _state_bufs(self._generator(new_keys), True)
        fresh_ts = ops.connector_observe(fresh)

        def pick(new, old):
            d = done.reshape((-1,) + (1,) * (old.dim() - 1))
            return torch.where(d, new, old)

        out = {}
        for f in ("grid", "step_count", "start", "target", "position"):
            out[f] = pick(fresh[f], so[f])
        out["key"] = pick(fresh["key"].view(torch.int32), so["key"].view(torch.int32)).view(torch.uint32)
        for f in ("obs_grid", "action_mask", "obs_step_count"):
            ts[f] = pick(fresh_ts[f], ts[f])
        return out, ts

    # ------------------------------------------------- "private" methods the reference calls
    def _timestep_of(self, state: State) -> TimeStep:
        batched = state.grid.dim() == 3
        return _make_timestep(ops.connector_observe(_state_bufs(state, batched)), batched)

    def _get_action_mask(self, agents: Agent, grid: torch.Tensor) -> torch.Tensor:
        """[3P] Connector._get_action_mask(agent, grid).
        * ONE agent (`agent.id` without an agent axis, as inside the reference's
          `jax.vmap(env._get_action_mask, (0, None))(state.agents, state.grid)`,
          demos/board_generator_demo.py:85-87): (5,) bool, or (B, 5) for a batch of grids with one
          agent each;
        * all agents at once (`agents.id` of shape (N,) / (B, N)): (N, 5) / (B, N, 5) -- what that
          vmap returns, in one launch."""
        batched = grid.dim() == 3
        B = grid.shape[0] if batched else 1
        dev = grid.device
        if agents.id.dim() == grid.dim() - 2:  # a single agent per grid
            N = self.num_agents
            ids = agents.id.reshape(B).long()
            if not ids.is_cuda:  # (a range check on device ids would cost a host sync per call:
                if bool(((ids < 0) | (ids >= N)).any()):  # those are clamped instead)
                    raise ValueError("agent.id outside [0, num_agents)")
                ids = ids.to(dev)
            ids = ids.clamp(0, N - 1)
            rows = torch.arange(B, device=dev)

            def slot(x, fill):  # (B, 2) -> (B, N, 2) with the agent in its own slot
                full = torch.full((B, N, 2), fill, dtype=torch.int32, device=dev)
                full[rows, ids] = x.reshape(B, 2).to(torch.int32)
                return full

            # the other slots sit on their own target off the board's reach: they are never read
            every = Agent(id=torch.arange(N, dtype=torch.int32, device=dev).expand(B, N),
                          start=slot(agents.start, 0), target=slot(agents.target, 0),
                          position=slot(agents.position, 0))
            g3 = grid if batched else grid[None]
            mask = self._get_action_mask(every, g3)[rows, ids]
            return mask if batched else mask[0]
        state = State(grid=grid, step_count=torch.zeros((B,) if batched else (), dtype=torch.int32, device=dev),
                      agents=agents,
                      key=torch.zeros((B, 2) if batched else (2,), dtype=torch.uint32, device=dev))
        return self._timestep_of(state).observation.action_mask

    def _obs_from_grid(self, grid: torch.Tensor) -> torch.Tensor:
        """(m, m) -> (N, m, m): every agent's view with itself as wire 0."""
        batched = grid.dim() == 3
        B = grid.shape[0] if batched else 1
        dev, N = grid.device, self.num_agents
        z = torch.zeros((B, N, 2) if batched else (N, 2), dtype=torch.int32, device=dev)
        agents = Agent(id=torch.arange(N, dtype=torch.int32, device=dev), start=z, target=z - 1, position=z)
        state = State(grid=grid, step_count=torch.zeros((B,) if batched else (), dtype=torch.int32, device=dev),
                      agents=agents,
                      key=torch.zeros((B, 2) if batched else (2,), dtype=torch.uint32, device=dev))
        return self._timestep_of(state).observation.grid

    def _get_extras(self, state: State):
        return self._timestep_of(state).extras
