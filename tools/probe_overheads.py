#!/usr/bin/env python3
"""Developer probe: host and synchronisation overheads around a short timed region (what a
20-step bench run pays besides the kernels)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import routing_board_generation_amd as rbg
from routing_board_generation_amd import ops
dev=torch.device('cuda',0)
keys=rbg.split(rbg.PRNGKey(0),4096)
out=ops.prw_generate(keys,16,10,split_first=True,want_solved=False,want_train=True,want_state=True,want_iters=True)
step=ops.prw_generate_launcher(keys,16,10,True,out)
for _ in range(2000): step()
torch.cuda.synchronize()
def t(f,n=200):
    t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e6
print('idle synchronize us', t(lambda: torch.cuda.synchronize(dev)))
e=torch.cuda.Event(enable_timing=True)
print('event record us', t(lambda: e.record()))
torch.cuda.synchronize()
print('event query us', t(lambda: e.query()))
for K in (1,5,20,100):
    res=[]
    for rep in range(5):
        torch.cuda.synchronize(dev)
        t0=time.perf_counter()
        for _ in range(K): step()
        t_issue=time.perf_counter()-t0
        torch.cuda.synchronize(dev)
        res.append(((time.perf_counter()-t0)*1e6, t_issue*1e6))
    print(K, 'total us, issue us', [tuple(round(x,1) for x in r) for r in res])
print('--- wall time of 20 launches: bare / with an event pair / events + poll')
for mode in ('bare', 'events', 'events+poll', 'bare'):
    res = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if mode != 'bare': e0.record()
        for _ in range(20): step()
        if mode != 'bare': e1.record()
        if mode == 'events+poll':
            while not e1.query(): pass
        torch.cuda.synchronize(dev); torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) * 1e6
        res.append((round(wall, 1), round(e0.elapsed_time(e1) * 1e3, 1) if mode != 'bare' else None))
    print(mode, res)
