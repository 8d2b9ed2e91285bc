"""Host logic of the next-episode stash (routing_board_generation_amd/schedule.py), CPU only:
when refills are queued, which tick a request becomes usable at, how the adaptive period moves.
The device side (rows, tags, requests) is covered by the GPU tests against the oracle."""
import random

from routing_board_generation_amd.schedule import MissWatch, RefillSchedule


def drive(sched, launches, misses_at=None):
    """Run `launches` ticks; returns (ready tick announced by every launch, ticks followed by a
    refill).  misses_at: tick -> cumulative miss count the host sees at the refill behind it."""
    announced, refills = [], []
    seen = getattr(sched, "_test_seen", 0)
    for _ in range(launches):
        announced.append(sched.tick + sched.ready_delay())
        if sched.refill_due():
            refills.append(sched.tick)
            if misses_at is not None:
                now = misses_at(sched.tick)
                sched.observe_misses(now - seen)
                seen = sched._test_seen = now
            sched.refill_queued()
        sched.launched()
    return announced, refills


def test_first_refill_follows_the_reset_then_every_kth_launch():
    s = RefillSchedule(every=16)
    announced, refills = drive(s, 70)
    assert refills == [0, 16, 32, 48, 64]
    assert announced[0] == 1                      # what the reset requests is usable by the first step
    assert announced[1:17] == [17] * 16 and announced[17:33] == [33] * 16


def test_a_request_is_usable_right_after_the_refill_that_follows_it():
    rnd = random.Random(5)
    for _ in range(50):
        K = rnd.randint(1, 60)
        s = RefillSchedule(every=K, batch=4096, adaptive=True)
        seen = [0]

        def misses(tick):
            seen[0] += rnd.choice([0, 0, 0, 10, 100, 5000])
            return seen[0]

        announced, refills = drive(s, 400, misses)
        for t, ready in enumerate(announced):
            nxt = min(r for r in refills + [10 ** 9] if r >= t)
            if nxt < 10 ** 9:
                assert ready == nxt + 1, (K, t, ready, nxt)   # never before the refill, never later
        gaps = [b - a for a, b in zip(refills[1:], refills[2:])]
        assert all(max(K // 8, 1) <= g <= K for g in gaps), (K, gaps)


def test_period_halves_under_misses_and_recovers_when_calm():
    s = RefillSchedule(every=48, batch=4096, adaptive=True)
    level = [0]

    def stormy(tick):
        level[0] += 200  # > batch / 64 new misses at every refill
        return level[0]

    drive(s, 200, stormy)
    assert s.K == 6  # 48 -> 24 -> 12 -> 6 = every / 8, and no further
    drive(s, 6 * 4 + 1, lambda tick: level[0])       # four calm refills
    assert s.K == 12
    drive(s, 2000, lambda tick: level[0])
    assert s.K == 48
    # a non-adaptive schedule ignores the counter
    fixed = RefillSchedule(every=48, batch=4096)
    drive(fixed, 300, stormy)
    assert fixed.K == 48


def test_a_few_misses_do_not_move_the_period():
    s = RefillSchedule(every=32, batch=4096, adaptive=True)
    total = [0]

    def drizzle(tick):
        total[0] += 3  # below batch / 64 per refill: neither calm nor a storm
        return total[0]

    drive(s, 500, drizzle)
    assert s.K == 32


def test_miss_snapshots_are_counted_once_per_stamp():
    """A host that runs ahead of the device reads the same snapshot many times, then several
    refills' worth at once: increments are handed out per stamp, stale reads give nothing."""
    w = MissWatch()
    assert w.observe(0) is None                                 # nothing has arrived yet
    assert w.observe(MissWatch.pack(10, 1)) == (10, 1)
    assert w.observe(MissWatch.pack(10, 1)) is None             # the same snapshot again
    assert w.observe(MissWatch.pack(130, 7)) == (120, 6)        # six refills later: 20 per refill
    top = MissWatch.STAMP_MASK
    w.stamp, w.misses = top, 130
    assert w.observe(MissWatch.pack(131, top + 2)) == (1, 2)    # the stamp wraps


def test_a_lagging_counter_is_neither_a_storm_nor_calm():
    """12 misses per refill (below batch / 64 = 64) arriving as one snapshot per 40 refills: the
    period stays; the same lag with 100 per refill halves it; stale reads never count as calm."""
    s = RefillSchedule(every=32, batch=4096, adaptive=True)
    w = MissWatch()
    for snap in range(1, 6):
        for _ in range(39):
            assert w.observe(MissWatch.pack(12 * 40 * (snap - 1), 40 * (snap - 1))) is None
        s.observe_misses(*w.observe(MissWatch.pack(12 * 40 * snap, 40 * snap)))
    assert s.K == 32
    s.observe_misses(*w.observe(MissWatch.pack(12 * 200 + 100 * 40, 240)))
    assert s.K == 16
    for _ in range(100):                                         # nothing new arrives: K stays put
        assert w.observe(MissWatch.pack(12 * 200 + 100 * 40, 240)) is None
    assert s.K == 16
    s.observe_misses(*w.observe(MissWatch.pack(12 * 200 + 100 * 40, 244)))  # four calm refills
    assert s.K == 32


def test_rebase_renumbers_the_serial_schedule_without_moving_anything():
    """Ticks are int32 in the ABI (include/rbg_prw.h): a session that would outlive 2^31 launches
    subtracts an offset from its tick, its next refill's tick and every ready_at it handed out.
    A schedule that rebases every few hundred launches announces, relative to the current tick,
    exactly what its twin that never rebases does; near 2^30 the numbers stay in int32."""
    for K, adaptive in ((1, False), (16, False), (48, True)):
        a, b = RefillSchedule(K, 4096, adaptive), RefillSchedule(K, 4096, adaptive)
        shifted = 0
        for step in range(5000):
            if step % 777 == 776:
                off = b.tick - 1
                b.rebase(off)
                shifted += off
            assert (a.ready_delay(), a.refill_due()) == (b.ready_delay(), b.refill_due())
            assert a.tick == b.tick + shifted and b.tick <= 777 + 1
            for s in (a, b):
                if s.refill_due():
                    s.observe_misses(500 if step % 900 < 300 else 0)
                    s.refill_queued()
                s.launched()
    s = RefillSchedule(20)
    s.tick, s.refill_after = (1 << 30) + 7, (1 << 30) + 13
    s.rebase(s.tick - 1)
    assert (s.tick, s.refill_after, s.ready_delay()) == (1, 7, 7)
    try:
        s.rebase(5)  # beyond the current tick
        assert False
    except ValueError:
        pass


def test_rolling_epochs_across_a_rebase():
    """The rolling form's arithmetic (schedule.RollingEpochs): requests of epoch e are usable from
    the first tick of epoch e + lead, ring slots follow the epoch, and a rebase -- a multiple of
    K x ring launches -- changes neither the phase within the epoch, nor any ring slot, nor any
    tick difference.  ready_at values on the device move down by the same offset (modelled here
    by the list `handed`)."""
    from routing_board_generation_amd.schedule import RollingEpochs

    for K, lead in ((1, 2), (8, 2), (32, 3), (64, 6)):
        sa, sb = RefillSchedule(K), RefillSchedule(K)
        ra, rb = RollingEpochs(sa, lead), RollingEpochs(sb, lead)
        assert ra.ring == lead + 1
        handed, shifted = [], 0  # absolute ready_at values the rebasing twin gave to the device
        threshold = 5 * K * ra.ring + 3
        for step in range(40 * K * ra.ring):
            if sb.tick >= threshold:
                off = rb.rebase_offset()
                assert off > 0 and off % (K * rb.ring) == 0
                sb.rebase(off)
                handed = [max(0, h - off) for h in handed]
                shifted += off
            assert sa.tick == sb.tick + shifted and sb.tick < threshold + 2 * K * rb.ring
            assert ra.ready_delay() == rb.ready_delay() >= 1
            assert ra.at_boundary() == rb.at_boundary()
            handed.append(sb.tick + rb.ready_delay())
            if sb.tick > 0:
                # usable exactly at the first tick of epoch e + lead
                e = rb.epoch()
                assert sb.tick + rb.ready_delay() == 1 + K * (e + lead)
            if ra.at_boundary():
                slot_a, claim_a, skip_a, aw_a, prev_a = ra.boundary()
                slot_b, claim_b, skip_b, aw_b, prev_b = rb.boundary()
                assert (slot_a, aw_a, prev_a) == (slot_b, aw_b, prev_b)
                assert claim_a == claim_b + shifted and skip_b == claim_b + K
                assert claim_b == sb.tick + rb.ready_delay()  # what this epoch's requests carry
                # the refill awaited before the next launch is the one whose rows become usable
                # there: refill e' = e + 1 - lead serves ready_at = 1 + K (e' + lead) = next tick
                if aw_b is not None:
                    assert 1 + K * ((rb.epoch() + 1 - lead) + lead) == sb.tick + 1
            sa.launched()
            sb.launched()
        assert shifted > 0
        # nothing handed out lies in the future beyond lead epochs, nothing is negative
        assert all(0 <= h <= sb.tick + K * (lead + 1) for h in handed)
    # at the real threshold the numbers stay far inside int32
    s = RefillSchedule(64)
    r = RollingEpochs(s, 3)
    s.tick = (1 << 30) + 12345
    before = (s.tick - 1) % 64, ((s.tick - 1) // 64) % r.ring
    s.refill_after = s.tick + 5
    s.rebase(r.rebase_offset())
    assert s.tick <= 2 * 64 * r.ring and ((s.tick - 1) % 64, ((s.tick - 1) // 64) % r.ring) == before
