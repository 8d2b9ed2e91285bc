// rbg_pieces.h -- the walk of ParallelRandomWalkBoard taken apart: the methods the reference's
// own unit tests call one by one (JAXI/.../test_parallel_random_walk_board.py: _step,
// _initialise_agents, _select_action, _adjacent_cells, _available_cells, _is_cell_free,
// _is_cell_doubling_back, _step_agent, _continue_stepping), on grids in the REFERENCE's encoding
// (PATH 1 + 3i, POSITION 2 + 3i, TARGET 3 + 3i) and with its treatment of arbitrary arguments:
// python floor division of flat cells, `-1` as "no cell", jnp's gather rule for out-of-range
// indices (a negative index wraps once, then the index is clamped).
//
// None of this is on the hot path -- generate_board runs the fused walk of rbg_walk.h, which keeps
// one value per wire and never materialises these intermediate results -- so the functions below
// are written for clarity, one thread per query, templated on where the grid lives (an int32
// grid in HBM, or the padded LDS board of a stepping kernel).
// Reference: JAXI/board_generation/parallel_random_walk.py:205-429.
#pragma once
#include "rbg_device.h"

namespace rbg {

__device__ __forceinline__ int floor_div(int a, int b) {  // python's //, b > 0
  const int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int floor_mod(int a, int b) { return a - floor_div(a, b) * b; }

// int32 grid [m][m] in HBM
struct HbmGrid {
  const int32_t *p;
  int m;
  __device__ __forceinline__ int at(int r, int c) const { return p[r * m + c]; }
};
// padded LDS board (rbg_device.h::Geo); (r, c) must be inside the grid
struct LdsGrid {
  const uint32_t *board;
  int S;
  __device__ __forceinline__ int at(int r, int c) const { return (int)board[(r + 2) * S + 4 + c]; }
};

// grid[divmod(cell, m)] the way jnp indexes: a negative index wraps once, then clamps
template <class G>
__device__ __forceinline__ int ref_gather(const G &grid, int m, int cell) {
  int r = floor_div(cell, m), c = floor_mod(cell, m);
  r += r < 0 ? m : 0;
  r = r < 0 ? 0 : r >= m ? m - 1 : r;
  c = c < 0 ? 0 : c >= m ? m - 1 : c;  // (floor_mod is already in [0, m))
  return grid.at(r, c);
}

// _adjacent_cells (:262-291): [up, down, left, right]; -1 when off the grid or wrapping a row
__device__ __forceinline__ void ref_adjacent_cells(int cell, int m, int out[4]) {
  const int cr = floor_div(cell, m), cc = floor_mod(cell, m);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cell + (k == 0 ? -m : k == 1 ? m : k == 2 ? -1 : 1);
    const bool inside = 0 <= c && c < m * m;
    const bool aligned = floor_div(c, m) == cr || floor_mod(c, m) == cc;
    out[k] = inside && aligned ? c : -1;
  }
}

// _is_cell_free (:321-338)
template <class G>
__device__ __forceinline__ bool ref_is_cell_free(const G &grid, int m, int cell) {
  return cell != -1 && ref_gather(grid, m, cell) == 0;
}

// _is_cell_doubling_back (:340-374): True when `cell` touches at most one cell of wire `wire`
template <class G>
__device__ __forceinline__ bool ref_not_doubling_back(const G &grid, int m, int wire, int cell) {
  int adj[4], touching = 0;
  ref_adjacent_cells(cell, m, adj);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (adj[k] == -1) continue;
    const int v = ref_gather(grid, m, adj[k]);
    touching += (int)(v == 3 * wire + 1 || v == 3 * wire + 2 || v == 3 * wire + 3);
  }
  return touching <= 1;
}

// _available_cells (:293-319); the wire is read off the grid at `cell`: (value - 1) // 3
template <class G>
__device__ __forceinline__ void ref_available_cells(const G &grid, int m, int cell, int out[4]) {
  int adj[4];
  ref_adjacent_cells(cell, m, adj);
  const int wire = floor_div(ref_gather(grid, m, cell) - 1, 3);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    out[k] = ref_is_cell_free(grid, m, adj[k]) && ref_not_doubling_back(grid, m, wire, adj[k])
                 ? adj[k]
                 : -1;
}

// _action_from_positions / _action_from_tuple (:238-260): NOOP, UP, RIGHT, DOWN, LEFT = 0 .. 4;
// a difference that is no unit step gives 0
__device__ __forceinline__ int ref_action_from_positions(int p1, int p2, int m) {
  const int dr = floor_div(p2, m) - floor_div(p1, m), dc = floor_mod(p2, m) - floor_mod(p1, m);
  return dr == -1 && dc == 0 ? 1 : dr == 1 && dc == 0 ? 3 : dr == 0 && dc == -1 ? 4
         : dr == 0 && dc == 1 ? 2 : 0;
}

// _select_action (:205-230): jax.random.choice(key, available, p = available != -1) -- cumsum,
// r = cum[-1] * (1 - uniform(key)), first cum >= r (index 0 when nothing is available: the "cell"
// is then -1 and the action whatever _action_from_positions makes of it, normally NOOP)
template <class G>
__device__ __forceinline__ int ref_select_action(Key2 key, const G &grid, int m, int cell) {
  int avail[4];
  ref_available_cells(grid, m, cell, avail);
  const float u = bits_to_uniform(random_bits_word(key, 0u, 1u));
  const float c1 = avail[0] != -1 ? 1.0f : 0.0f, c2 = c1 + (avail[1] != -1 ? 1.0f : 0.0f);
  const float c3 = c2 + (avail[2] != -1 ? 1.0f : 0.0f), c4 = c3 + (avail[3] != -1 ? 1.0f : 0.0f);
  const float r = __fmul_rn(c4, __fsub_rn(1.0f, u));
  const int idx = c1 >= r ? 0 : c2 >= r ? 1 : c3 >= r ? 2 : 3;
  return ref_action_from_positions(cell, avail[idx], m);
}

}  // namespace rbg
