"""ParallelRandomWalkBoard -- drop-in for the reference class of the same name
(routing_board_generation/board_generation_methods/jax_implementation/board_generation/
parallel_random_walk.py:49-447).

`generate_board` runs the fused gfx950 walk (`prw_generate_kernel`).  The "private" methods that
the reference's unit tests call one by one (test_parallel_random_walk_board.py) are here too,
with the reference's signatures, on the HIP kernels of csrc/rbg_pieces.h: they take the
reference's grids (PATH 1 + 3i, POSITION 2 + 3i, TARGET 3 + 3i) and `Agent` pytrees -- one agent
(`agent.id` a scalar) or all N stacked, which is what `jax.vmap(method, in_axes=(None, 0))`
gives in the reference.  Pure index arithmetic (`_convert_*`, `_action_from_*`) is host code.
"""
from typing import Any, Tuple

import torch

from .. import ops
from ..types import Agent

UP, RIGHT, DOWN, LEFT, NOOP = 1, 2, 3, 4, 0  # [3P] jumanji connector/constants.py


class ParallelRandomWalkBoard:
    def __init__(self, rows: int, cols: int, num_agents: int):
        if rows != cols:
            # the reference's _adjacent_cells mixes `rows` and `cols` (parallel_random_walk.py
            # :276-285) and every caller passes a square grid
            raise ValueError("ParallelRandomWalkBoard supports square grids only")
        self.rows = rows
        self.cols = cols
        self.num_agents = num_agents

    def generate_board(self, key: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """key (2,) -> heads (2, N), targets (2, N), solved_grid (m, m);
        key (B, 2) -> the same with a leading B (what `jax.vmap(generate_board)` returns)."""
        batched = key.dim() == 2
        keys = key if batched else key[None]
        out = ops.prw_generate(keys.contiguous(), self.rows, self.num_agents, split_first=False,
                               want_solved=True, want_train=False)
        heads = out["start"].transpose(-1, -2)
        targets = out["target"].transpose(-1, -2)
        grid = out["solved_grid"]
        if batched:
            return heads, targets, grid
        return heads[0], targets[0], grid[0]

    def generate_board_with_stats(self, keys: torch.Tensor):
        """Batched variant that also returns the while_loop trip count per board."""
        out = ops.prw_generate(keys.contiguous(), self.rows, self.num_agents, split_first=False,
                               want_solved=True, want_train=False, want_iters=True)
        return (out["start"].transpose(-1, -2), out["target"].transpose(-1, -2),
                out["solved_grid"], out["iters"])

    # ------------------------------------------------------------------ helpers of the pieces
    def _dev(self, *xs) -> torch.device:
        for x in xs:
            if isinstance(x, torch.Tensor) and x.is_cuda:
                return x.device
        return torch.device("cuda", torch.cuda.current_device())

    def _grid3(self, grid: torch.Tensor, dev) -> torch.Tensor:
        """(m, m) of any dtype -> contiguous int32 [1, m, m] on the device."""
        g = torch.as_tensor(grid).to(device=dev, dtype=torch.int32)
        if g.shape != (self.rows, self.cols):
            raise ValueError(f"grid must have shape {(self.rows, self.cols)}, got {tuple(g.shape)}")
        return g.contiguous()[None]

    def _stacked(self, agent: Agent, dev) -> Tuple[Agent, Any]:
        """One agent or N stacked agents -> (all-N Agent with a leading batch axis of 1, ids or
        None).  A single agent is placed in the slot of its id; the other slots are inert."""
        N = self.num_agents
        i32 = dict(dtype=torch.int32, device=dev)
        if torch.as_tensor(agent.id).dim() == 0:
            k = int(agent.id)
            if not 0 <= k < N:
                raise ValueError("agent.id outside [0, num_agents)")
            full = {}
            for name, fill in (("start", 0), ("target", -1), ("position", 0)):
                t = torch.full((1, N, 2), fill, **i32)
                t[0, k] = torch.as_tensor(getattr(agent, name)).to(**i32)
                full[name] = t
            return Agent(id=torch.arange(N, **i32)[None], **full), k
        if tuple(torch.as_tensor(agent.id).shape) != (N,):
            raise ValueError(f"agents must be stacked over the {N} agents")
        return Agent(id=torch.arange(N, **i32)[None],
                     start=torch.as_tensor(agent.start).to(**i32).reshape(1, N, 2).contiguous(),
                     target=torch.as_tensor(agent.target).to(**i32).reshape(1, N, 2).contiguous(),
                     position=torch.as_tensor(agent.position).to(**i32).reshape(1, N, 2)
                     .contiguous()), None

    def _cells(self, cell, dev) -> Tuple[torch.Tensor, bool]:
        c = torch.as_tensor(cell).to(device=dev, dtype=torch.int32)
        return c.reshape(1, -1).contiguous(), c.dim() == 0

    # ------------------------------------------------------------------------- the pieces
    def _return_blank_board(self) -> torch.Tensor:
        return torch.zeros((self.rows, self.cols), dtype=torch.int32, device=self._dev())

    def _initialise_agents(self, key: torch.Tensor, grid: torch.Tensor) -> Tuple[torch.Tensor, Agent]:
        """(:147-186) key (2,) and an (empty) grid -> (grid with the heads, agents)."""
        dev = self._dev(key, grid)
        g, start = ops.prw_initialise_agents(key.reshape(1, 2).contiguous(), self.rows,
                                             self.num_agents, self._grid3(grid, dev))
        start = start[0]
        agents = Agent(id=torch.arange(self.num_agents, dtype=torch.int32, device=dev), start=start,
                       target=torch.full_like(start, -1), position=start.clone())
        return g[0], agents

    def _place_agent_heads_on_grid(self, grid: torch.Tensor, agent: Agent) -> torch.Tensor:
        """(:188-190) grid.at[agent.start].set(2 + 3 id); stacked agents give one grid each."""
        g = torch.as_tensor(grid)
        ids = torch.as_tensor(agent.id, device=g.device)
        start = torch.as_tensor(agent.start, device=g.device).long()
        if ids.dim() == 0:
            out = g.clone()
            out[start[0], start[1]] = (2 + 3 * ids).to(g.dtype)
            return out
        out = g.expand(ids.shape[0], *g.shape).clone()
        out[torch.arange(ids.shape[0], device=g.device), start[:, 0], start[:, 1]] = \
            (2 + 3 * ids).to(g.dtype)
        return out

    def _step_agents(self, key: torch.Tensor, grid: torch.Tensor, agents: Agent) -> Tuple[Agent, torch.Tensor]:
        """(:101-145) all agents draw an action and step at once; of colliding agents the one
        with the lower id stays where it was."""
        dev = self._dev(key, grid)
        full, _ = self._stacked(agents, dev)
        _, g, pos, _ = ops.prw_step(key.reshape(1, 2).contiguous(), self._grid3(grid, dev),
                                    full.target, full.position)
        return agents._replace(position=pos[0]), g[0]

    def _step(self, stepping_tuple: Tuple[torch.Tensor, torch.Tensor, Agent]
              ) -> Tuple[torch.Tensor, torch.Tensor, Agent]:
        """(:92-99) one iteration of the walk: (key, grid, agents) -> (next_key, grid, agents)."""
        key, grid, agents = stepping_tuple
        dev = self._dev(key, grid)
        full, _ = self._stacked(agents, dev)
        nk, g, pos, _ = ops.prw_step(key.reshape(1, 2).contiguous(), self._grid3(grid, dev),
                                     full.target, full.position)
        return nk[0], g[0], agents._replace(position=pos[0])

    def _continue_stepping(self, stepping_tuple: Tuple[torch.Tensor, torch.Tensor, Agent]) -> torch.Tensor:
        """(:192-198) True while some agent has an available cell."""
        _, grid, agents = stepping_tuple
        return ~self._is_any_step_possible(grid, agents).all()

    def _is_any_step_possible(self, grid: torch.Tensor, agent: Agent) -> torch.Tensor:
        """(:200-203) despite its name: True when the agent has NO available cell."""
        cell = self._convert_tuple_to_flat_position(torch.as_tensor(agent.position).movedim(-1, 0))
        return (self._available_cells(grid, cell) == -1).all(dim=-1)

    def _select_action(self, key: torch.Tensor, grid: torch.Tensor, agent: Agent) -> torch.Tensor:
        """(:205-230) one key per agent: key (2,) with one agent, (N, 2) with N stacked."""
        dev = self._dev(key, grid)
        full, k = self._stacked(agent, dev)
        N = self.num_agents
        keys = torch.zeros((1, N, 2), dtype=torch.int64, device=dev)
        if k is None:
            keys[0] = key.to(dev).to(torch.int64).reshape(N, 2)
        else:
            keys[0, k] = key.to(dev).to(torch.int64).reshape(2)
        act = ops.prw_select_action(keys.to(torch.uint32), self._grid3(grid, dev), full.position)
        return act[0] if k is None else act[0, k]

    def _convert_flat_position_to_tuple(self, position) -> torch.Tensor:
        p = torch.as_tensor(position)
        return torch.stack([torch.div(p, self.cols, rounding_mode="floor"),
                            torch.remainder(p, self.cols)]).to(torch.int32)

    def _convert_tuple_to_flat_position(self, position) -> torch.Tensor:
        p = torch.as_tensor(position)
        return (p[0] * self.cols + p[1]).to(torch.int32)

    def _action_from_positions(self, position_1, position_2) -> torch.Tensor:
        return self._action_from_tuple(self._convert_flat_position_to_tuple(position_2) -
                                       self._convert_flat_position_to_tuple(position_1))

    def _action_from_tuple(self, action_tuple) -> torch.Tensor:
        """(:247-260) (-1, 0) UP, (1, 0) DOWN, (0, -1) LEFT, (0, 1) RIGHT; anything else NOOP."""
        t = torch.as_tensor(action_tuple)
        dr, dc = t[0], t[1]
        act = (UP * ((dr == -1) & (dc == 0)).int() + DOWN * ((dr == 1) & (dc == 0)).int() +
               LEFT * ((dr == 0) & (dc == -1)).int() + RIGHT * ((dr == 0) & (dc == 1)).int())
        return act.to(torch.int32)

    def _adjacent_cells(self, cell) -> torch.Tensor:
        """(:262-291) [up, down, left, right] flat neighbours, -1 where there is none."""
        cells, scalar = self._cells(cell, self._dev(cell))
        adj = ops.prw_cell_queries(None, cells, self.rows, want=("adjacent",))["adjacent"][0]
        return adj[0] if scalar else adj

    def _available_cells(self, grid: torch.Tensor, cell) -> torch.Tensor:
        """(:293-319) the neighbours a wire standing on `cell` may step into, -1 elsewhere."""
        dev = self._dev(grid, cell)
        cells, scalar = self._cells(cell, dev)
        out = ops.prw_cell_queries(self._grid3(grid, dev), cells, self.rows,
                                   want=("available",))["available"][0]
        return out[0] if scalar else out

    def _is_cell_free(self, grid: torch.Tensor, cell) -> Tuple[torch.Tensor, torch.Tensor]:
        """(:321-338) (grid, grid[cell] == 0), False for cell -1."""
        dev = self._dev(grid, cell)
        cells, scalar = self._cells(cell, dev)
        free = ops.prw_cell_queries(self._grid3(grid, dev), cells, self.rows,
                                    want=("is_free",))["is_free"][0]
        return grid, (free[0] if scalar else free)

    def _is_cell_doubling_back(self, grid_wire_id: Tuple[torch.Tensor, Any], cell):
        """(:340-374) ((grid, wire_id), True when `cell` touches wire_id's cells at most once)."""
        grid, wire_id = grid_wire_id
        dev = self._dev(grid, cell)
        cells, scalar = self._cells(cell, dev)
        wires = torch.as_tensor(wire_id).to(device=dev, dtype=torch.int32).expand(cells.shape[1])
        ok = ops.prw_cell_queries(self._grid3(grid, dev), cells, self.rows,
                                  wire_id=wires.reshape(1, -1).contiguous(),
                                  want=("not_doubling_back",))["not_doubling_back"][0]
        return (grid, wire_id), (ok[0] if scalar else ok)

    def _step_agent(self, agent: Agent, grid: torch.Tensor, action) -> Tuple[Agent, torch.Tensor]:
        """(:376-399) the agent moves on its own copy of the grid if the move is valid.  N stacked
        agents and actions give N agents and N grids (vmap in_axes (0, None, 0))."""
        dev = self._dev(grid, agent.position)
        full, k = self._stacked(agent, dev)
        N = self.num_agents
        acts = torch.zeros((1, N), dtype=torch.int32, device=dev)
        a = torch.as_tensor(action).to(device=dev, dtype=torch.int32)
        if k is None:
            acts[0] = a.reshape(N)
        else:
            acts[0, k] = a.reshape(())
        pos, grids, _ = ops.prw_step_agent(self._grid3(grid, dev), full.target, full.position, acts)
        if k is None:
            return agent._replace(position=pos[0]), grids[0]
        return agent._replace(position=pos[0, k]), grids[0, k]

    def _is_valid_position(self, grid: torch.Tensor, agent: Agent, position) -> torch.Tensor:
        """(:401-429) in bounds, empty or the agent's own target, and the agent not connected."""
        g = torch.as_tensor(grid)
        row, col = int(position[0]), int(position[1])
        size = g.shape[0]
        if not (0 <= row < size and 0 <= col < size):
            return torch.tensor(False, device=g.device)
        v = g[row, col]
        open_cell = (v == 0) | (v == 3 + 3 * torch.as_tensor(agent.id, device=g.device))
        return open_cell & ~agent.connected.to(g.device)

    def update_solved_board_with_head_target_encodings(self, solved_grid: torch.Tensor, heads,
                                                       targets) -> torch.Tensor:
        """(:431-447) heads (rows, cols) get 2 + 3i, then targets 3 + 3i (the target wins)."""
        out = solved_grid.clone()
        ids = torch.arange(self.num_agents, device=out.device)
        out[torch.as_tensor(heads[0]).long(), torch.as_tensor(heads[1]).long()] = \
            (2 + 3 * ids).to(out.dtype)
        out[torch.as_tensor(targets[0]).long(), torch.as_tensor(targets[1]).long()] = \
            (3 + 3 * ids).to(out.dtype)
        return out
