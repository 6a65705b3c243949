"""Event simulation of the match kernel's work sharing on the measured per-scan pass counts, to compare
helper policies off the GPU.  Model constants are the measured ones: set-up 53 us, a pass alone 34 us, a shared pass by
number of helpers (timeline table of 4.5, + 3.5 us sum and optimiser step), a helper's window 11 us + 2 us to find a scan.
Usage: python tools/sim_helpers.py gpurun_out/ev/evals2048.npy [B]"""
import heapq
import sys

import numpy as np

SETUP, SOLO, ATTACH = 53.0, 34.0, 13.0
COLLECT = {1: 20.3, 2: 16.2, 3: 12.9, 4: 11.8, 7: 10.5, 15: 9.8}


def pass_time(h):
    if h == 0:
        return SOLO
    ks = sorted(COLLECT)
    return float(np.interp(h, ks, [COLLECT[k] for k in ks])) + 3.5


def simulate(n_pass, policy, cap=8, base=7, penalty=12, n_wg=256, clairvoyant=False):
    B = len(n_pass)
    left = n_pass.astype(float).copy()          # passes still to do
    done_p = np.zeros(B)                         # passes done
    helpers = np.zeros(B, int)                   # registered helpers
    pending = [[] for _ in range(B)]             # helpers still building their window: ready times
    finished = np.zeros(B, bool)
    t_end = np.zeros(B)
    ev = []                                      # (time, kind, scan)   kind 0: pass of scan ends
    nxt = min(B, n_wg)
    for b in range(nxt):
        heapq.heappush(ev, (SETUP + SOLO, 0, b))
    free_at = []                                 # idle workgroups looking for work: (time)
    busy = 0.0
    now = 0.0

    def assign(t):
        # a free workgroup at time t picks a scan (or None)
        unfinished = int((~finished[:nxt]).sum()) + (B - nxt)
        if unfinished == 0:
            return None
        room = min(cap, max(min(cap, base), n_wg // max(unfinished, 1) - 1))
        cand = [b for b in range(nxt) if not finished[b] and helpers[b] + len(pending[b]) < room]
        if not cand:
            return None
        if clairvoyant:
            key = lambda b: (-(left[b] / 1.0) + 3.0 * (helpers[b] + len(pending[b])))
        elif policy == "current":
            key = lambda b: -(min(done_p[b], 200) - penalty * (helpers[b] + len(pending[b])))
        elif policy == "fewest":
            key = lambda b: (helpers[b] + len(pending[b]), -done_p[b])
        elif policy == "most_passes":
            key = lambda b: (-done_p[b], helpers[b] + len(pending[b]))
        return min(cand, key=key)

    idle = []                                    # workgroups with nothing joinable (retry at next event)
    while ev:
        now, kind, b = heapq.heappop(ev)
        left[b] -= 1; done_p[b] += 1
        # helpers whose window is ready by now register
        ready = [x for x in pending[b] if x <= now]
        pending[b] = [x for x in pending[b] if x > now]
        helpers[b] += len(ready)
        freed = []
        if left[b] <= 0:
            finished[b] = True; t_end[b] = now
            freed = [now] * (1 + helpers[b] + len(pending[b]))
            helpers[b] = 0; pending[b] = []
            if nxt < B:                          # owner takes the next scan of the queue
                freed.pop()
                heapq.heappush(ev, (now + SETUP + SOLO, 0, nxt)); nxt += 1
        else:
            heapq.heappush(ev, (now + pass_time(helpers[b]), 0, b))
        for t in freed + idle:
            tgt = assign(now)
            if tgt is None:
                continue
            pending[tgt].append(now + ATTACH)
        # (workgroups that found nothing retry at the next event)
        unfinished = int((~finished[:nxt]).sum())
        idle = [now] * max(0, 0)                 # placeholder: idle workgroups are recomputed below
        n_attached = int(helpers[:nxt][~finished[:nxt]].sum()) + sum(len(pending[x]) for x in range(nxt) if not finished[x])
        n_idle = n_wg - unfinished - n_attached
        idle = [now] * max(0, n_idle) if unfinished else []
    return t_end.max(), t_end


def main():
    ev = np.load(sys.argv[1])
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n = ev[:B]
    print("B", B, "mean passes %.2f max %d" % (n.mean(), n.max()), "work per CU %.0f us" % ((SETUP + n.mean() * SOLO) * B / 256))
    for name, kw in (("current cap 8", dict(policy="current")), ("current cap 15", dict(policy="current", cap=15)),
                     ("current cap 4", dict(policy="current", cap=4, base=4)),
                     ("fewest helpers first", dict(policy="fewest")), ("most passes first", dict(policy="most_passes")),
                     ("penalty 4", dict(policy="current", penalty=4)), ("penalty 30", dict(policy="current", penalty=30)),
                     ("clairvoyant", dict(policy="current", clairvoyant=True)), ("clairvoyant cap 15", dict(policy="current", clairvoyant=True, cap=15))):
        mk, te = simulate(n, **kw)
        print("%-24s makespan %6.1f us   finish 10%% %5.0f  median %5.0f  90%% %5.0f" % (name, mk, *np.percentile(te, [10, 50, 90])))


if __name__ == "__main__":
    main()
