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
