"""A model of a K1c launch: rounds as bdf_col_plan_build makes them, single-wave workgroups dealt round-robin over 32 shader engines
(w mod 32; the row stream's CU mask leaves the first eight with 28 SIMDs), two waves per SIMD, the older wave at 8 cycles per
instruction, the younger at 17.6 while the older runs (tools/valu_cost_probe.hip: 8.0 / 5.5 cycles per instruction at one / two waves
per SIMD; stamps: 430 / 900 cycles per observation step).  Compares wave orders by the modelled launch length."""
import os, sys, heapq
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

C_OBS, C_FIN, C_FOLD, C_PART, PART_EXTRA = 56.0, 1460.0, 180.0, 700.0, 3000.0

def rounds_of(counts, T=128):
    """-> list of (cost_model, true_cost_fn placeholder) : cost in instructions"""
    singles, pairs, R = [], [], []
    for n in counts:
        n = int(n)
        if n <= T: singles.append(n)
        elif n <= 2 * T: pairs.append(n)
        elif n <= 4 * T: R.append(C_OBS * ((n + 3) // 4) + 2 * C_FOLD + C_FIN)
        else:
            W = (n + 4 * T - 1) // (4 * T)
            share = (C_FIN + 2 * C_FOLD + 60.0 * ((W + 3) // 4)) / W
            for q in range(W):
                m = n * (q + 1) // W - n * q // W
                R.append(C_OBS * ((m + 3) // 4) + 2 * C_FOLD + C_PART + share + PART_EXTRA)
    pairs.sort(reverse=True); singles.sort(reverse=True)
    si = 0
    for pi in range(0, len(pairs), 2):
        lmax = pairs[pi] - pairs[pi] // 2
        if pi + 1 >= len(pairs):
            for q in range(2):
                if si < len(singles): lmax = max(lmax, singles[si]); si += 1
        R.append(C_OBS * lmax + C_FOLD + C_FIN)
    while si < len(singles):
        R.append(C_OBS * singles[si] + C_FIN); si += 4
    return np.array(R)

def order_current(cost, S2=896):
    rank = np.argsort(-cost, kind="stable")
    nw = len(rank)
    if nw <= S2: return rank
    two = min(nw, 2 * S2); P = two - S2; A = S2 - P
    return np.concatenate([rank[A:A + P], rank[:A], rank[two - 1:two - 1 - P:-1] if P else [], rank[two:]]).astype(int)

def se_sizes(reserve):
    return [28 if (reserve and se < 8) else 32 for se in range(32)]

def simulate(cost_in_order, reserve=True, old=8.0, young=17.6):
    """event simulation; returns launch length in cycles"""
    sizes = se_sizes(reserve)
    nw = len(cost_in_order)
    per_se = [[] for _ in range(32)]
    for w in range(nw): per_se[w % 32].append(cost_in_order[w])
    end = 0.0
    for se in range(32):
        n = sizes[se]; q = per_se[se]
        # SIMD state: list of (remaining_instr) for older/younger; process via time stepping per SIMD with a shared FIFO
        simds = [[] for _ in range(n)]          # each: list of remaining instr, index 0 = older
        nxt = 0
        for s in range(n):
            if nxt < len(q): simds[s].append(q[nxt]); nxt += 1
        for s in range(n):
            if nxt < len(q): simds[s].append(q[nxt]); nxt += 1
        t = [0.0] * n                              # local clocks
        # event loop: repeatedly advance the SIMD with the earliest next completion
        def next_done(s):
            w = simds[s]
            if not w: return None
            return t[s] + w[0] * old               # the older wave always runs at `old`
        heap = [(next_done(s), s) for s in range(n) if simds[s]]
        heapq.heapify(heap)
        while heap:
            td, s = heapq.heappop(heap)
            w = simds[s]
            dt = td - t[s]
            if len(w) > 1: w[1] = max(0.0, w[1] - dt / young)
            w.pop(0); t[s] = td
            end = max(end, td)
            if nxt < len(q):                       # a waiting workgroup of this engine takes the free slot (as the younger wave)
                w.append(q[nxt]); nxt += 1
            if w: heapq.heappush(heap, (next_done(s), s))
    return end

def order_se_aware(cost, reserve=True):
    """per engine: exact complements (heaviest with lightest) within the engine's share, engines with fewer SIMDs get lighter work"""
    sizes = se_sizes(reserve)
    rank = list(np.argsort(-cost, kind="stable"))
    nw = len(rank)
    per = [nw // 32 + (1 if se < nw % 32 else 0) for se in range(32)]        # waves each engine receives (w mod 32)
    # deal the ranked waves to engines: heavy ones snake over the engines weighted by SIMD count
    lists = [[] for _ in range(32)]
    load = [0.0] * 32
    for r in rank:
        # engine with room and the smallest load per SIMD
        best = min((se for se in range(32) if len(lists[se]) < per[se]), key=lambda se: (load[se] / sizes[se], se))
        lists[best].append(r); load[best] += cost[r]
    out = np.empty(nw, dtype=int)
    for se in range(32):
        L = lists[se]                     # descending cost
        n = sizes[se]
        m = len(L)
        seq = [None] * m
        # positions 0..n-1: older waves; n..2n-1: their partners; beyond: wait for slots (lightest last come first to free slots?)
        n_old = min(n, m)
        n_pair = min(n, max(0, m - n))
        rest = m - n_old - n_pair
        # the heaviest n_old are the older waves; partners: lightest of the remaining, lightest beside heaviest; the rest (medium) wait
        olds = L[:n_old]
        others = L[n_old:]
        partners = others[len(others) - n_pair:][::-1] if n_pair else []          # lightest first
        waiting = others[:len(others) - n_pair]
        # pairs that are a single older wave (no partner) should be the heaviest: put partner-less olds first? partner k sits beside old k
        # olds with partners: the n_pair LIGHTEST olds get... keep simple: old k (k-th heaviest) gets partner k (k-th lightest) for k < n_pair
        seq = olds + partners + waiting
        for k, r in enumerate(seq): out[32 * k + se] = r
    return out


# ---- a planner IN simulated time: workgroup after workgroup in dispatch order, the round that still ends by T* on the slot the model
# says the workgroup gets (an empty SIMD: the heaviest that fits alone; beside an older wave / on a freed slot: the heaviest that ends
# by T*), T* by bisection.  Result on MovieLens (users / movies, modelled launch, k cycles): T = 96: 73.7 / 75.5, T = 112: 76.4 / 77.2,
# T = 128: 76.8 / 78.4 against 77.9 / 76.9 for the order the library uses -- within 5 %: the older / younger issue rates leave a SIMD
# ~75 % busy whatever the order, and the library's complementary order is already there.  Not built into the library.
import bisect
OLD, YOUNG = 8.0, 17.6
def greedy_order(cost, nsimd_per_se=28, n_se=32, Tstar=None):
    """dispatch-order list scheduling in simulated time.  Returns order (list of round indices) or None if Tstar infeasible."""
    n = len(cost)
    srt = sorted(range(n), key=lambda r: cost[r])           # ascending costs
    vals = [cost[r] for r in srt]
    avail = list(range(n))                                   # indices into srt (ascending cost), maintained as sorted list of positions
    import bisect
    pos = list(range(n))                                     # remaining positions in ascending cost order
    def pop_largest_le(x):
        # largest cost <= x among remaining; None if none
        i = bisect.bisect_right([vals[p] for p in pos], x) - 1 if False else None
        return None
    # maintain remaining as a sorted list of (cost, idx)
    rem = sorted((cost[r], r) for r in range(n))
    keys = [c for c, _ in rem]
    def take_le(x):
        i = bisect.bisect_right(keys, x) - 1
        if i < 0: return None
        keys.pop(i); return rem.pop(i)[1]
    def take_max():
        keys.pop(); return rem.pop()[1]
    # per-SE state: each SIMD: [t, older_remaining, younger_remaining]
    order = []
    se_state = [[[0.0, None, None] for _ in range(nsimd_per_se)] for _ in range(n_se)]
    se_next_empty = [0] * n_se
    w = 0
    while rem:
        se = w % n_se
        st = se_state[se]
        k = se_next_empty[se]
        if k < nsimd_per_se:
            # empty SIMD: becomes older.  heaviest remaining that fits alone: 8 c <= Tstar
            r = take_le(Tstar / OLD)
            if r is None: return None
            st[k][1] = cost[r]; se_next_empty[se] += 1
            order.append(r); w += 1; continue
        if k < 2 * nsimd_per_se:
            s = st[k - nsimd_per_se]
            se_next_empty[se] += 1
            R = s[1]                                   # older's remaining (all started at t = 0)
            # finish time of a younger of cost c: c <= 0.4545 R -> 17.6 c ; else 8 R + 8 (c - R*OLD/YOUNG)
            cmax = (Tstar - OLD * R) / OLD + R * OLD / YOUNG if Tstar > OLD * R else Tstar / YOUNG
            r = take_le(cmax)
            if r is None:
                order.append(-1); w += 1; continue     # nothing fits: leave the slot (an empty workgroup)
            s[2] = cost[r]
            order.append(r); w += 1; continue
        # third generation: this SE's SIMD that frees a slot first
        best = None
        for i, s in enumerate(st):
            # advance: older finishes at t + 8*R ; then younger becomes older
            if s[1] is None: continue
            tfree = s[0] + OLD * s[1]
            if best is None or tfree < best[0]: best = (tfree, i)
        if best is None: return None
        tfree, i = best
        s = st[i]
        dt = tfree - s[0]
        y = s[2]
        if y is not None: y = max(0.0, y - dt / YOUNG)
        s[0] = tfree; s[1] = y; s[2] = None
        if s[1] is None or s[1] <= 0.0:
            # SIMD empty at tfree: new wave is older alone
            cmax = (Tstar - tfree) / OLD
            r = take_le(cmax)
            if r is None: return None
            s[1] = cost[r]; order.append(r); w += 1; continue
        R = s[1]
        rem_t = Tstar - tfree
        cmax = (rem_t - OLD * R) / OLD + R * OLD / YOUNG if rem_t > OLD * R else rem_t / YOUNG
        r = take_le(cmax)
        if r is None: return None
        s[2] = cost[r]; order.append(r); w += 1
    return order

def plan(cost, nsimd=28):
    lo, hi = cost.max() * OLD, cost.sum() * OLD
    best = None
    for _ in range(24):
        mid = 0.5 * (lo + hi)
        o = greedy_order(cost, nsimd, 32, mid)
        if o is None: lo = mid
        else: best, hi = (o, mid), mid
    return best


if __name__ == "__main__":
    import bdf_amd as B
    from bdf_amd import datasets
    rd, _ = datasets.movielens_relation_data(B, ntest=500_000, seed=1, alpha=1.5, class_cut=2.5)
    rel = rd.relations[0]
    ids = np.asarray(rel.data.ids)
    for mode, name in ((0, "users"), (1, "movies")):
        cnt = np.bincount(ids[:, mode] - 1, minlength=rel.data.dims[mode])
        cost = rounds_of(cnt)
        tot = cost.sum()
        for reserve in (True, False):
            nS = sum(se_sizes(reserve))
            ideal = tot * 5.5 / nS
            cur = simulate(cost[order_current(cost)], reserve)
            cf = simulate(cost[np.argsort(-cost, kind="stable")], reserve)
            sea = simulate(cost[order_se_aware(cost, reserve)], reserve)
            print(f"{name:6s} reserve={int(reserve)} rounds {len(cost)} total {tot/1e6:.2f} M instr; packed bound {ideal/1e3:.1f} k cycles; heaviest alone {cost.max()*8/1e3:.1f} k | "
                  f"costliest-first {cf/1e3:.1f} k | current (S2=896) {cur/1e3:.1f} k | S2=1024 {simulate(cost[order_current(cost, 1024)], reserve)/1e3:.1f} k | engine-aware {sea/1e3:.1f} k")


def two_class_plan(cost, P, ratio=8.0 / 17.6):
    """rounds -> 2 P waves (lists of rounds): P `older` waves of capacity 1 and P `younger` ones of capacity `ratio`, longest
    processing time first onto the wave that would finish earliest (its load plus the round, over its capacity)"""
    order = np.argsort(-cost, kind="stable")
    cap = np.concatenate([np.ones(P), np.full(P, ratio)])
    load = np.zeros(2 * P)
    waves = [[] for _ in range(2 * P)]
    import heapq
    heap = [(0.0, i) for i in range(2 * P)]
    heapq.heapify(heap)
    # uniform-machines LPT: place on the machine minimising (load + c) / cap  -- scan a few candidates from the heap
    for r in order:
        c = cost[r]
        best, bi = None, None
        # exact argmin needs a scan; 2P = 1792..2048 is small
        fin = (load + c) / cap
        bi = int(np.argmin(fin))
        waves[bi].append(int(r)); load[bi] += c
    return waves, load


def simulate_waves(wave_costs_in_order, reserve=True, old=8.0, young=17.6):
    return simulate(np.asarray(wave_costs_in_order, dtype=float), reserve, old, young)
