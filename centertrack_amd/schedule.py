"""Branch-level concurrency of a frame's launch plan on several HIP streams.

At batch 1 most layers of DLA-34 run on <= 512 workgroups and spend a third of their time
ramping up and draining, while the network has independent branches: the IDAUp ``proj`` DCNs
(dla.py:539-545) depend only on earlier backbone levels, the residual ``project`` 1x1 convs and
2x2 max-pools (dla.py:206-228) only on a level's input.  This module derives the dependency
DAG of a plan from the buffers each launch reads / writes (channel-slice precise: RAW, WAR and
WAW hazards), list-schedules the launches onto S streams using the autotuner's measured
durations, and records for every launch which other-stream predecessors it has to wait for.
Captured in a HIP graph the waits become graph edges, so the independent branches overlap.
"""

BIG = 1 << 30


def region(obj):
    """(buffer id, first channel, end channel) of a View or of a whole torch tensor."""
    if obj is None:
        return None
    if hasattr(obj, 'buf'):
        return (obj.buf.data_ptr(), obj.c0, obj.c0 + obj.C)
    return (obj.data_ptr(), 0, BIG)


def _overlap(a, b):
    return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]


def _conflict(li, lj):
    """does launch j (later in program order) have to wait for launch i?"""
    for w in li.writes:
        for r in lj.reads:
            if _overlap(w, r):
                return True
        for w2 in lj.writes:
            if _overlap(w, w2):
                return True
    for r in li.reads:
        for w in lj.writes:
            if _overlap(r, w):
                return True
    return False


def schedule(launches, num_streams):
    """Sets ``stream`` and ``waits`` (indices of launches on OTHER streams to wait for) on every
    launch and ``record`` on those somebody waits for.  Returns the simulated makespan (us)."""
    n = len(launches)
    deps = [[i for i in range(j) if _conflict(launches[i], launches[j])] for j in range(n)]
    S = max(1, num_streams)
    avail = [0.0] * S
    finish = [0.0] * n
    for j, l in enumerate(launches):
        l.record = False
        ready, last_dep = 0.0, -1
        for i in deps[j]:
            if finish[i] >= ready:
                ready, last_dep = finish[i], i
        best_s, best_t = 0, None
        pref = launches[last_dep].stream if last_dep >= 0 else 0
        for s in ([pref] + [s for s in range(S) if s != pref]):
            t = max(ready, avail[s])
            # a side stream must buy a real head start (its fork / join costs a few us)
            if best_t is None or t < best_t - (3.0 if s != pref else 0.0):
                best_s, best_t = s, t
        l.stream = best_s
        finish[j] = best_t + max(l.us, 1.0)
        avail[best_s] = finish[j]
        # only the latest predecessor on each other stream needs an event (stream order covers the rest)
        latest = {}
        for i in deps[j]:
            s = launches[i].stream
            if s != best_s and (s not in latest or i > latest[s]):
                latest[s] = i
        l.waits = sorted(latest.values())
        for i in l.waits:
            launches[i].record = True
    return max(finish) if n else 0.0


def serialize(launches):
    for l in launches:
        l.stream, l.waits, l.record = 0, [], False
