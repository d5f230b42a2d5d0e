"""CPU tests of the multi-stream launch scheduler (centertrack_amd/schedule.py): hazards between
channel slices of shared buffers, stream assignment and the cross-stream waits it emits."""
from centertrack_amd import schedule


class FakeBuf(object):
    _next = 1000

    def __init__(self):
        FakeBuf._next += 64
        self._p = FakeBuf._next

    def data_ptr(self):
        return self._p


class FakeView(object):
    def __init__(self, buf, c0, C):
        self.buf, self.c0, self.C = buf, c0, C


class L(object):
    def __init__(self, name, reads, writes, us=10.0):
        self.name = name
        self.reads = [schedule.region(r) for r in reads]
        self.writes = [schedule.region(w) for w in writes]
        self.us = us
        self.stream, self.waits, self.record = 0, [], False


def test_slices_of_one_buffer_only_conflict_when_they_overlap():
    b = FakeBuf()
    lo, hi, whole = FakeView(b, 0, 64), FakeView(b, 64, 64), FakeView(b, 0, 128)
    assert not schedule._overlap(schedule.region(lo), schedule.region(hi))
    assert schedule._overlap(schedule.region(lo), schedule.region(whole))
    assert schedule._overlap(schedule.region(hi), schedule.region(whole))
    assert not schedule._overlap(schedule.region(lo), schedule.region(FakeView(FakeBuf(), 0, 64)))


def test_independent_branch_goes_to_a_side_stream_and_joins_with_one_wait():
    a, b, c, d, e = (FakeView(FakeBuf(), 0, 16) for _ in range(5))
    launches = [
        L('p0', [], [a], 20.0),          # producer
        L('main1', [a], [b], 30.0),      # critical chain a -> b -> c
        L('main2', [b], [c], 30.0),
        L('side1', [a], [d], 25.0),      # needs only a: can overlap main1/main2
        L('join', [c, d], [e], 10.0),
    ]
    span = schedule.schedule(launches, 2)
    assert [l.stream for l in launches] == [0, 0, 0, 1, 0]
    assert launches[3].waits == [0] and launches[0].record          # side stream waits for the producer
    assert launches[4].waits == [3] and launches[3].record          # join waits for the side branch only
    assert launches[1].waits == [] and launches[2].waits == []
    assert span == 20 + 30 + 30 + 10                                 # side work fully hidden
    schedule.serialize(launches)
    assert all(l.stream == 0 and not l.waits and not l.record for l in launches)


def test_war_and_waw_hazards_serialise():
    buf = FakeBuf()
    x = FakeView(buf, 0, 32)
    y = FakeView(FakeBuf(), 0, 32)
    launches = [
        L('w1', [], [x]),
        L('r1', [x], [y]),
        L('w2', [], [x]),                # overwrites x: must wait for r1 (WAR) and w1 (WAW)
    ]
    schedule.schedule(launches, 3)
    deps_ok = launches[2].stream == launches[1].stream or 1 in launches[2].waits
    assert deps_ok
    # a short dependent launch is not moved to another stream just to start a little earlier
    assert launches[1].stream == launches[0].stream
