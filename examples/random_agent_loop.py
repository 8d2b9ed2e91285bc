#!/usr/bin/env python3
"""The reference's acting loop on the native env, one process per GPU.

What `agent_training/training_script.py:70-116` does with jumanji's RandomAgent --
setup_training_state (key fan-out, `pmap(env.reset)`), `n_steps` of `env.step` per epoch, metrics
averaged over devices -- with the drop-in classes of routing_board_generation_amd:

    python examples/random_agent_loop.py                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/random_agent_loop.py                          # 8 GPUs, RCCL only for the metrics
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops, sharding  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid-size", type=int, default=16)
    ap.add_argument("--num-agents", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--n-steps", type=int, default=50)
    ap.add_argument("--one-launch-per-epoch", action="store_true",
                    help="collect the epoch's trajectory with ONE kernel launch "
                         "(env.random_rollout_record(...)) instead of n-steps env.step calls")
    args = ap.parse_args()

    rank, local_rank, world = sharding.init_distributed()
    dev = torch.device("cuda", local_rank)

    # setup_train._make_raw_env + setup_env (setup_train.py:148-167)
    generator = rbg.ParallelRandomWalkGenerator(grid_size=args.grid_size,
                                                num_agents=args.num_agents)
    if args.one_launch_per_epoch:  # per-agent rewards: the recording rollout returns those
        env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=generator))
    else:
        env = rbg.VmapAutoResetWrapper(
            rbg.MultiToSingleWrapper(rbg.Connector(generator=generator, donate=True)))

    # setup_training_state (setup_train.py:390-403): split the key, reset this rank's shard
    key = rbg.PRNGKey(0, device=dev)
    state, timestep, acting_key = sharding.reset_sharded(
        env, key, args.batch_per_gpu * world, rank, world)
    policy_keys = rbg.split(acting_key, args.batch_per_gpu)

    t = 0
    for epoch in range(args.epochs):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        episodes = torch.zeros((), dtype=torch.int64, device=dev)
        ep_return = torch.zeros((), dtype=torch.float64, device=dev)
        connections = torch.zeros((), dtype=torch.int64, device=dev)
        if args.one_launch_per_epoch:
            # the same n_steps of RandomAgent + env.step, recorded by one launch: every field of
            # `traj` has a leading time axis [n_steps, B, ...] (what lax.scan stacks in run_epoch)
            state, _, actions, traj = env.random_rollout_record(state, policy_keys, t, args.n_steps)
            last = traj.last()
            episodes += last.sum()
            ep_return += traj.reward.sum()
            connections += (traj.extras["num_connections"] * last).sum()
            t += args.n_steps
        for _ in range(0 if args.one_launch_per_epoch else args.n_steps):
            # RandomAgent: uniform over the legal actions of every agent
            action = ops.random_actions(policy_keys, t, timestep.observation.action_mask)
            state, timestep = env.step(state, action)
            last = timestep.last()
            episodes += last.sum()
            ep_return += timestep.reward.sum()
            connections += (timestep.extras["num_connections"] * last).sum()
            t += 1
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        m = sharding.all_reduce_metrics({
            "env_steps": float(args.batch_per_gpu * args.n_steps), "episodes": float(episodes),
            "num_connections": float(connections), "elapsed_s": elapsed}, device=dev)
        if rank == 0:
            print(f"epoch {epoch}: {m['env_steps'] / m['elapsed_s'] / 1e6:7.1f} M env-steps/s on "
                  f"{world} GPU(s), {int(m['episodes'])} episodes, "
                  f"{m['num_connections'] / max(m['episodes'], 1):.2f} connections per episode, "
                  f"mean step reward {float(ep_return) / (args.batch_per_gpu * args.n_steps):+.4f}")


if __name__ == "__main__":
    main()
