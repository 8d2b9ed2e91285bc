#!/usr/bin/env python3
"""The reference's generator comparison under a random agent, on the native env.

`routing_board_generation/benchmarking/benchmarks/benchmark_on_random_agent.py:59-102`
(`BenchmarkOnRandomAgent.from_simulation`) runs `train(cfg)` with `agent=random` once per board
generator and keeps, per epoch, what `BenchmarkData` holds
(`benchmarking/utils/benchmark_data_model.py:24-31`): episode_length, episode_return,
num_connections, ratio_connections, total_path_length, time.  Here every epoch is ONE fused
rollout launch per generator (`env.random_rollout`: the masked-uniform random policy + `n_steps`
auto-resetting steps), and the per-epoch figures come from its integer counters:

    python examples/benchmark_on_random_agent.py [--grid-size 10 --num-agents 5 --epochs 5]

Prints one JSON object per generator with the `BenchmarkData` fields as per-epoch lists plus the
summary the reference's bar charts show (`average_*` / `std_*`).  The random policy is this
project's (include/rbg_prw.h::rbg_random_actions), not jumanji's RandomAgent, and the env
semantics are "parity unpinned" (DESIGN.md section 2): figures are comparable between
generators, not with the reference's saved plots.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd.interface.board_generator_interface import make_online_generator  # noqa: E402

GENERATION_TYPES = ("online_parallel_rw", "online_uniform", "online_seq_rw", "online_random_seed",
                    "offline_parallel_rw", "offline_seed_extension")


def mean_std(xs):
    t = torch.tensor(xs, dtype=torch.float64)
    return float(t.mean()), float(t.std(unbiased=False))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid-size", type=int, default=10)
    ap.add_argument("--num-agents", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--n-steps", type=int, default=100, help="env steps per epoch and env")
    ap.add_argument("--generators", default=",".join(GENERATION_TYPES))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    m, N, B = args.grid_size, args.num_agents, args.batch
    keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
    policy_keys = rbg.split(rbg.PRNGKey(1, device=dev), B)
    for name in args.generators.split(","):
        kw = {}
        if name.startswith("offline"):  # agent_training/configs/env/connector.yaml seed_extension.*
            kw = dict(randomness=0.0, two_sided=True, extension_iterations=1, extension_steps=1e23,
                      number_of_boards=10000)
        env = rbg.VmapAutoResetWrapper(rbg.Connector(make_online_generator(name, m, N, **kw)))
        state, _ = env.reset(keys)
        data = {k: [] for k in ("episode_length", "episode_return", "num_connections",
                                "ratio_connections", "time", "total_path_length")}
        rf = env.unwrapped._reward_fn
        for epoch in range(args.epochs):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            state, stats, _ = env.random_rollout(state, policy_keys, epoch * args.n_steps,
                                                 args.n_steps)
            s = {k: float(v.sum().item()) for k, v in stats.items()}  # (synchronises)
            data["time"].append(time.perf_counter() - t0)
            eps = max(s["episodes"], 1.0)
            data["episode_length"].append(B * args.n_steps / eps)
            data["episode_return"].append((rf.connected_reward * s["connections_made"] +
                                           rf.timestep_reward * s["unconnected_agent_steps"]) / eps)
            data["num_connections"].append(s["final_connections"] / eps)
            data["ratio_connections"].append(s["final_connections"] / eps / N)
            data["total_path_length"].append(s["final_path_length"] / eps)
        summary = {}
        for field, label in (("episode_return", "reward_per_wire"),
                             ("total_path_length", "total_wire_length"),
                             ("ratio_connections", "proportion_of_wires_connected"),
                             ("episode_length", "steps_till_board_terminates")):
            summary["average_" + label], summary["std_" + label] = mean_std(data[field])
        print(json.dumps({"generator_type": name, "rows": m, "columns": m, "number_of_wires": N,
                          "envs": B, "steps_per_epoch": args.n_steps, **data, **summary}),
              flush=True)


if __name__ == "__main__":
    main()
