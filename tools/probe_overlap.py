"""Developer probe: how much do a long, latency-bound generator launch (SeedExtension or
SequentialRandomWalk, 4096 boards) and the HBM-bound step launches slow each other down when
they share the GPU?  Step loop (auto-reset off, policy fused) on one stream, `n` generator
launches kept in flight on side streams."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import routing_board_generation_amd as rbg  # noqa: E402
from routing_board_generation_amd import ops  # noqa: E402

B, m, N = 4096, 16, 10
dev = torch.device("cuda", 0)
keys = rbg.split(rbg.PRNGKey(0, device=dev), B)
gen_name = sys.argv[1] if len(sys.argv) > 1 else "seed"


def gen_launch(k):
    if gen_name == "seed":
        return ops.seed_extension_generate(k, m, N, split_first=True)
    return ops.seq_rw_generate(k, m, N, split_first=True, want_solved=False)


st, ts = ops.alloc_state(B, m, N, dev), ops.alloc_timestep(B, m, N, dev)
step_gen = None
if len(sys.argv) > 2:  # the env's generator kind (which step kernel runs): e.g. online_random_seed
    from routing_board_generation_amd.interface.board_generator_interface import make_online_generator
    step_gen = make_online_generator(sys.argv[2], m, N).native_spec()
loop = ops.StepLoop(st, ts, keys, time_limit=1 << 30, refill_every=0, fuse_policy=True, auto_reset=False,
                    gen=step_gen)
loop.reset(keys)
for t in range(50):
    loop.step(t)
torch.cuda.synchronize()


def time_steps(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(n):
        loop.step(t)
    e1.record()
    return e0, e1


# alone
e0, e1 = time_steps(400)
torch.cuda.synchronize()
steps_alone = e0.elapsed_time(e1) / 400 * 1e3
g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gen_launch(keys)
torch.cuda.synchronize()
g0.record()
for _ in range(5):
    gen_launch(keys)
g1.record()
torch.cuda.synchronize()
gen_alone = g0.elapsed_time(g1) / 5 * 1e3
print(json.dumps({"generator": gen_name, "steps_alone_us": round(steps_alone, 2), "gen_alone_us": round(gen_alone, 1)}), flush=True)

for n_side in (1, 2, 3, 4):
    sides = [torch.cuda.Stream(device=dev) for _ in range(n_side)]
    side_keys = [rbg.split(rbg.PRNGKey(10 + i, device=dev), B) for i in range(n_side)]
    torch.cuda.synchronize()
    evs = []
    reps = 6
    for i, s in enumerate(sides):
        with torch.cuda.stream(s):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                gen_launch(side_keys[i])
            b.record()
            evs.append((a, b))
    # main stream: step while the generators run (~reps * gen time)
    n_steps = int(reps * gen_alone * 1.2 / steps_alone)
    e0, e1 = time_steps(n_steps)
    torch.cuda.synchronize()
    gen_us = [a.elapsed_time(b) / reps * 1e3 for a, b in evs]
    print(json.dumps({"generators_in_flight": n_side, "steps_us": round(e0.elapsed_time(e1) / n_steps * 1e3, 2),
                      "gen_us_each": [round(x, 1) for x in gen_us],
                      "boards_per_us_total": round(n_side * B / (sum(gen_us) / n_side), 2)}), flush=True)
