import sys, torch
sys.path.insert(0,'/root/repo')
import routing_board_generation_amd as rbg
from routing_board_generation_amd import ops
def timed(fn, reps):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for (m,N,B) in ((16,10,4096),(16,10,16384),(16,10,65536),(32,20,4096),(8,5,4096)):
    keys=rbg.split(rbg.PRNGKey(0),B)
    o1=ops.prw_generate(keys,m,N,split_first=True,want_solved=False,want_state=True,want_iters=True)
    o2=ops.prw_generate_philox(keys,m,N)
    t1=timed(lambda: ops.prw_generate(keys,m,N,split_first=True,want_solved=False,want_state=True,want_iters=True,out=o1),100)
    t2=timed(lambda: ops.prw_generate_philox(keys,m,N,out=o2),100)
    print(f"{m}x{m}/{N} B={B}: threefry(jax) {t1:8.1f} us   philox {t2:8.1f} us   T max {int(o1['iters'].max())} / {int(o2['iters'].max())}", flush=True)
