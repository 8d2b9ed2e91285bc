"""The next-episode stash's protocol on the GPU (include/rbg_prw.h::rbg_stash, rbg_stash_list,
rbg_stash_roll; routing_board_generation_amd/session.py::EpochStash):

  * the rolling form (several refills in flight beside the stepping stream) at the benchmark's
    batch, every step against the oracle;
  * the list launch's claims: acquire order, no row listed twice, ready_at kept truthful for a row
    that was re-assigned while its producer ran;
  * launch ticks rebased long before int32 wraps, results unchanged across the rebase.

Parity status: the env semantics are "parity unpinned" (DESIGN.md section 2): the HIP path is
compared with this project's oracle (VmapAutoResetWrapper._auto_reset written out,
tests/test_gpu_env_generators.py).
"""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_dicts_equal, state_to_np, timestep_to_np, to_np
from test_gpu_env_generators import _keys, aggregate, make_pair, oracle_auto_reset_step, oracle_reset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rbg():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (there is no CPU fallback to hide behind)")
    import routing_board_generation_amd as pkg

    pkg._native.lib()
    return pkg


@pytest.mark.parametrize("name", ["online_seq_rw", "online_random_seed"])
def test_rolling_refills_at_the_benchmark_batch_match_oracle(rbg, oracle, name):
    """BASELINE configs[2] / [4] shape: 16x16 / 10 agents, B = 4096, time limit 50, the generator's
    DEFAULT schedule -- refills rolling beside the stepping stream (8 / 10 slots per env, several
    producers in flight, one rbg_stash_roll per epoch) -- through the reference's training env
    VmapAutoResetWrapper(MultiToSingleWrapper(Connector(g))): State and the full timestep against
    the oracle after EVERY step.  The host never waits for a refill here except through the
    protocol's own events (the per-step comparison synchronises the stepping stream only after
    the step it compares; the refills keep running on their streams)."""
    from routing_board_generation_amd import ops

    m, N, B, tl, steps = 16, 10, 4096, 50, 160
    gen, gen_fn = make_pair(rbg, oracle, name, m, N)
    env = rbg.VmapAutoResetWrapper(rbg.MultiToSingleWrapper(
        rbg.Connector(generator=gen, time_limit=tl)))
    keys, akeys = _keys(rbg, 91, B), _keys(rbg, 92, B)
    state, ts = env.reset(keys)
    (sess,) = env.unwrapped._sessions.values()
    ep = sess.epochs
    assert ep is not None and not ep.serial and ep.stash["depth"] >= 8  # the rolling default
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    assert_dicts_equal(state_to_np(state), est, "reset state.")
    n_last = 0
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert (to_np(action) == eaction).all(), f"policy diverged at step {t}"
        assert_dicts_equal(state_to_np(state), est, f"{name} step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), aggregate(ets, N), f"{name} step {t} timestep.")
        n_last += int((ets["step_type"] == 2).sum())
    sess.drain()
    assert not ep.fell_back and not ep.serial
    hits, misses = int(to_np(ep.stash["hits"]).sum()), int(to_np(ep.stash["misses"]).sum())
    assert n_last > 2 * B and hits + misses == n_last, (hits, misses, n_last)
    assert misses < 0.02 * n_last, (hits, misses)
    assert ep._rolled >= steps // ep.K - 1  # one library call per epoch boundary


def _stash(rbg, B, m, N, depth):
    from routing_board_generation_amd import ops

    stash = ops.alloc_stash(B, m, N, "cuda", depth, rolling=True)
    return stash, ops._stash_struct(stash, B, m, N)


def test_list_claims_keep_ready_at_truthful(rbg):
    """rbg_stash_list on a constructed stash: a row whose producer is still running (claimed) is
    never listed again; if its request is NEWER than the claim (the env re-assigned the row
    while the producer ran) its ready_at moves on to the next list's tick, because only that
    list's producer can generate it -- otherwise a step launch could pass `tick >= ready_at`
    with no event ordering against the producer that writes the row.  A row that is
    free is claimed with this list's ready tick; a row whose tag already answers its request is
    left alone."""
    import torch

    nat = rbg._native
    lib = nat.lib()
    B, m, N, D = 64, 16, 10, 3
    stash, struct = _stash(rbg, B, m, N, D)
    R = D * B
    rows = torch.arange(R, device="cuda")
    kind = rows % 5
    # kind 0: satisfied (tag == refill_key); 1: free, pending; 2: claimed by an older list whose
    # claim covers the request (ready_at <= claim); 3: claimed, request newer than the claim;
    # 4: claimed, newer, but already promised a later tick than the next list's
    r_np = np.arange(R, dtype=np.uint32)
    want_np = np.stack([r_np + 1000, r_np + 5], axis=1)                  # every row has a request
    tag_np = want_np.copy()
    tag_np[r_np % 5 != 0, 0] = 7
    want = torch.from_numpy(want_np).cuda()
    stash["refill_key"].copy_(want)
    stash["tag"].copy_(torch.from_numpy(tag_np).cuda())
    claim_old, claim_now, skip = 201, 301, 401
    inflight = torch.where(kind >= 2, claim_old, 0).to(torch.int32)
    ready = torch.full((R,), 150, dtype=torch.int32, device="cuda")
    ready[kind == 1] = claim_now
    ready[kind == 2] = claim_old
    ready[kind == 3] = claim_now
    ready[kind == 4] = 999
    stash["inflight"].copy_(inflight)
    stash["ready_at"].copy_(ready)
    stash["scratch"].zero_()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nat.check(lib.rbg_stash_list(B, m, N, C.byref(struct), claim_now, skip, stream), "list")
    torch.cuda.synchronize()
    scratch = to_np(stash["scratch"])
    listed = np.sort(scratch[1:1 + scratch[0]])
    k = to_np(kind)
    assert (listed == np.nonzero(k == 1)[0]).all(), "exactly the free rows with a pending request"
    infl, rdy = to_np(stash["inflight"]), to_np(stash["ready_at"])
    assert (infl[k == 1] == claim_now).all() and (infl[k >= 2] == claim_old).all()
    assert (infl[k == 0] == 0).all()
    assert (rdy[k == 3] == skip).all(), "re-assigned rows move on to the next list's tick"
    assert (rdy[k == 2] == claim_old).all() and (rdy[k == 4] == 999).all()
    assert (rdy[k == 0] == 150).all() and (rdy[k == 1] == claim_now).all()
    # the next list (nothing released in between) lists nothing twice
    stash["scratch"].zero_()
    nat.check(lib.rbg_stash_list(B, m, N, C.byref(struct), skip, skip + 100, stream), "list")
    torch.cuda.synchronize()
    assert int(stash["scratch"][0]) == 0
    assert (to_np(stash["ready_at"])[k == 3] == skip + 100).all()  # still in flight: moves on again
    # released rows (the producer wrote tag, then cleared the claim) are not listed either: the
    # producer publishes the tag before the release, the list reads the claim first
    stash["tag"].copy_(want)
    stash["inflight"].zero_()
    nat.check(lib.rbg_stash_list(B, m, N, C.byref(struct), skip + 100, skip + 200, stream), "list")
    torch.cuda.synchronize()
    assert int(stash["scratch"][0]) == 0


def test_listed_producer_generates_each_row_once_and_releases_it(rbg, oracle):
    """rbg_stash_list + rbg_stash_refill_listed with claims: every pending row is generated from
    its request (checked against the oracle's generator), tagged, and released; the miss counter
    snapshot of rbg_stash_roll arrives packed as (misses << 20 | stamp)."""
    import torch

    from routing_board_generation_amd import ops
    from routing_board_generation_amd.schedule import MissWatch

    nat = rbg._native
    lib = nat.lib()
    B, m, N, D = 40, 10, 4, 2
    stash, struct = _stash(rbg, B, m, N, D)
    R = D * B
    req = rbg.split(rbg.PRNGKey(5), R)
    stash["refill_key"].copy_(req)
    stash["misses"].copy_(torch.arange(B, dtype=torch.int32, device="cuda"))
    stream = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    evs = [torch.cuda.Event() for _ in range(3)]
    for ev in evs:
        ev.record()
    acc = torch.zeros((1,), dtype=torch.int64, device="cuda")
    word = torch.zeros((1,), dtype=torch.int64, device="cuda")
    host = torch.zeros((1,), dtype=torch.int64).pin_memory()
    call = nat.StashRollCall()
    call.B, call.m, call.N = B, m, N
    call.claim_ready, call.skip_ready, call.stamp = 11, 21, 77
    call.stash = C.pointer(struct)
    call.step_stream, call.gen_stream = stream.cuda_stream, side.cuda_stream
    call.launched, call.listed, call.done = (e.cuda_event for e in evs)
    call.miss_acc, call.miss_word, call.miss_host = acc.data_ptr(), word.data_ptr(), host.data_ptr()
    nat.check(lib.rbg_stash_roll(C.byref(call)), "rbg_stash_roll")
    torch.cuda.synchronize()
    assert int(stash["scratch"][0]) == R
    assert (to_np(stash["tag"]) == to_np(req)).all() and (to_np(stash["inflight"]) == 0).all()
    assert MissWatch().observe(int(host[0])) == (B * (B - 1) // 2, 77) and int(acc[0]) == 0
    # next[r] = generator(split(refill_key[r])[0]): ParallelRandomWalkGenerator on the oracle side
    child = np.stack([oracle.split(k)[0] for k in to_np(req)])
    est = oracle.generate_batch(child, m, N, "parallel_random_walk", threads=8, want_solved=False)[0]
    got = {f: to_np(stash["next"][f]) for f in ops.STATE_FIELDS}
    assert_dicts_equal(got, est, "stash row ")
    # a second boundary with nothing pending generates nothing
    stash["scratch"].zero_()
    call.stamp = 78
    nat.check(lib.rbg_stash_roll(C.byref(call)), "rbg_stash_roll")
    torch.cuda.synchronize()
    assert int(stash["scratch"][0]) == 0 and MissWatch().observe(int(host[0]))[1] == 78
    # host memory the device cannot address (pageable): the snapshot arrives by an 8-byte copy
    pageable = torch.zeros((1,), dtype=torch.int64)
    call.miss_host, call.stamp = pageable.data_ptr(), 79
    nat.check(lib.rbg_stash_roll(C.byref(call)), "rbg_stash_roll")
    torch.cuda.synchronize()
    assert MissWatch().observe(int(pageable[0])) == (B * (B - 1) // 2, 79)
    assert int(word[0]) == int(pageable[0])


@pytest.mark.parametrize("name,delay", [("online_parallel_rw", None), ("online_seq_rw", None),
                                        ("online_random_seed", 6)])
def test_ticks_are_rebased_before_int32_wraps(rbg, oracle, name, delay, monkeypatch):
    """Launch ticks are int32 in the ABI; EpochStash renumbers them (and every ready_at on the
    device) at 2^30.  With the threshold lowered so that a short run crosses it several times --
    in line (the walk) and with rolling refills in flight -- every step still equals the oracle,
    the stash keeps serving the resets, and the tick stays small."""
    from routing_board_generation_amd import ops, session

    m, N, B, tl, steps = 10, 4, 64, 7, 150
    monkeypatch.setattr(session.EpochStash, "REBASE_AT", 40)
    gen, gen_fn = make_pair(rbg, oracle, name, m, N)
    env = rbg.VmapAutoResetWrapper(rbg.Connector(generator=gen, time_limit=tl),
                                   prefetch_delay=delay)
    keys, akeys = _keys(rbg, 95, B), _keys(rbg, 96, B)
    state, ts = env.reset(keys)
    (sess,) = env.unwrapped._sessions.values()
    ep = sess.epochs
    est, ets = oracle_reset(oracle, gen_fn, to_np(keys))
    top, n_last = 0, 0
    for t in range(steps):
        action = ops.random_actions(akeys, t, ts.observation.action_mask)
        eaction = oracle.random_actions_batch(to_np(akeys), t, ets["action_mask"])
        state, ts = env.step(state, action)
        ets = oracle_auto_reset_step(oracle, gen_fn, est, eaction, tl)
        assert_dicts_equal(state_to_np(state), est, f"{name} step {t} state.")
        assert_dicts_equal(timestep_to_np(ts), ets, f"{name} step {t} timestep.")
        top = max(top, ep.tick)
        n_last += int((ets["step_type"] == 2).sum())
    sess.drain()
    assert top < steps and top <= 40 + 2 * ep.K * (1 if ep.serial else ep.roll.ring), (top, ep.K)
    assert int(to_np(ep.stash["ready_at"]).max()) <= top + ep.K * 8
    hits = int(to_np(ep.stash["hits"]).sum())
    assert hits > 0.5 * n_last, (hits, n_last)  # rebasing did not cost the stash its rows
