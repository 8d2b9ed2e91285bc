"""Developer probe: the fused (and the recording) random rollout for every generation_type,
50 steps per launch, two-slot stash refilled behind every launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402
from routing_board_generation_amd.interface.board_generator_interface import make_online_generator  # noqa: E402

B, m, N, T = 4096, 16, 10, 50
dev = torch.device("cuda", 0)
keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else [
    "online_parallel_rw", "online_uniform", "offline_parallel_rw", "online_seq_rw", "online_random_seed"]
for name in names:
    kw = dict(randomness=0.0, two_sided=True, number_of_boards=10000) if name.startswith("offline") else {}
    gen = make_online_generator(name, m, N, **kw).native_spec()
    for use_stash in (True, False):
        for record in (False, True):
            ep = ops.EpochStash(B, m, N, dev, every=1, serial=True, depth=2, gen=gen) if use_stash else None
            st, _ = ops.connector_reset(keys, m, N, gen=gen)
            if ep is not None:
                ep.before_launch()
                st, _ = ops.connector_reset_prefetch(keys, m, N, ep.stash, ep.tick, ep.ready_delay(), gen=gen)
                ep.after_launch()
            traj = ops.alloc_trajectory(T, B, m, N, dev) if record else None
            stats = {k: torch.empty((B,), dtype=torch.int32, device=dev) for k in ops.ROLLOUT_STATS}

            def launch(i):
                if record:
                    ops.connector_rollout_record(st, keys, i * T, T, state_out=st, trajectory=traj,
                                                 epochs=ep, gen=gen)
                else:
                    ops.connector_rollout(st, keys, i * T, T, state_out=st, stats=stats, epochs=ep, gen=gen)

            for i in range(3):
                launch(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(3, 13):
                launch(i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            print(json.dumps({"generator": name, "stash": use_stash, "record": record,
                              "us_per_launch": round(us, 1),
                              "env_steps_per_s_M": round(B * T / us, 1)}), flush=True)
