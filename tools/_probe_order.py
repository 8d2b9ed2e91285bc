import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import routing_board_generation_amd as rbg
from routing_board_generation_amd import ops
m, N = 16, 10
keys = rbg.split(rbg.PRNGKey(0), 4096)
full = ops.prw_generate(keys, m, N, split_first=True, want_solved=False, want_state=True, want_iters=True)
it = full["iters"].cpu()
order = torch.argsort(it)
def timeit(idx, reps=300):
    k = keys.view(torch.int32)[idx.to(keys.device)].contiguous().view(torch.uint32)
    out = ops.prw_generate(k, m, N, split_first=True, want_solved=False, want_state=True, want_iters=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.prw_generate(k, m, N, split_first=True, want_solved=False, want_state=True, want_iters=True, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
nat = torch.arange(4096)
smax = int(order[-1]); pair0 = smax & ~1
print("straggler board", smax, "T", int(it[smax]), "partner T", int(it[smax ^ 1]), "block", smax // 8)
print("boards with T>=50:", int((it >= 50).sum()), " T>=56:", int((it >= 56).sum()))
def swap(idx, a, b):
    idx = idx.clone(); idx[[a, a + 1, b, b + 1]] = idx[[b, b + 1, a, a + 1]]; return idx
desc = order.flip(0)
tests = {
 "natural": nat,
 "natural, straggler pair to front": swap(nat, 0, pair0),
 "natural, straggler pair to end": swap(nat, 4094, pair0),
 "sorted desc": desc,
 "sorted desc, top pair to end": swap(desc, 0, 4094),
 "sorted desc, top pair to block 256": swap(desc, 0, 2048),
 "random perm": torch.randperm(4096, generator=torch.Generator().manual_seed(1)),
 "T>=40 first (natural within)": torch.cat([nat[it >= 40], nat[it < 40]]),
 "pairs sorted (pair max desc)": None,
 "blocks interleaved long/short": None,
}
# pairs kept as in natural order, but pairs ordered by max T descending
pm = torch.maximum(it[0::2], it[1::2]); po = torch.argsort(pm, descending=True)
tests["pairs sorted (pair max desc)"] = torch.stack([2 * po, 2 * po + 1], 1).reshape(-1)
# sorted desc within pairs (long with long) but pair order random
pairs = desc.reshape(-1, 2); perm = torch.randperm(2048, generator=torch.Generator().manual_seed(2))
tests["blocks interleaved long/short"] = pairs[perm].reshape(-1)
for name, idx in tests.items():
    print(f"{name:40s} {timeit(idx):7.2f} us")
