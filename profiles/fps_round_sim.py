"""How many picks does one exchange of the speculative FPS kernel decide?  A numpy model of csrc/fps.hip fps_spec_kernel's rounds (16 waves,
each publishing its best point, the best point of its other lanes and a bound; entries accepted in order while they beat the global bound
and are not changed by the entries before them): rounds, picks per round and why rounds end, for the blocked Morton layout the kernel uses,
for tiles interleaved across the waves, and for more published entries per wave.  CPU only.
usage: python profiles/fps_round_sim.py uniform|lidar [m] [greedy|lazy]
(round 4: uniform 480 rounds = 8.5 picks per round, LiDAR-shaped 742 = 5.5; interleaved tiles 467 / 595 -- but every wave then rebuilds
its entries every round, four waves per SIMD; 3 / 4 entries per wave 395 / 383 and 587 / 522 -- at the price of a 64-entry merge.)"""
import os, sys, importlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = importlib.import_module('3d_adapt_auto_driving_amd.synth')

def morton_order(p):
    x, z = p[:, 0], p[:, 2]
    gx = np.clip(((x - x.min()) / max(x.max() - x.min(), 1e-9) * 64).astype(int), 0, 63)
    gz = np.clip(((z - z.min()) / max(z.max() - z.min(), 1e-9) * 64).astype(int), 0, 63)
    def part(v):
        v = v.astype(np.uint32); r = np.zeros_like(v)
        for b in range(6): r |= ((v >> b) & 1) << (2 * b)
        return r
    code = part(gx) | (part(gz) << 1)
    return np.argsort(code, kind='stable')

def simulate(p, m, layout, PPT=16, top=2, rmax=16, stats=None, W=16, greedy=False):
    n = p.shape[0]
    order = morton_order(p)
    # slot index array [w][i][lane] -> position s in Morton order
    w_, i_, l_ = np.meshgrid(np.arange(W), np.arange(PPT), np.arange(64), indexing='ij')
    if layout == 'blocked':
        s = w_ * 64 * PPT + i_ * 64 + l_
    elif layout == 'interleaved':
        s = (i_ * W + w_) * 64 + l_
    elif layout == 'quad':       # tiles interleaved across groups of 4 waves
        g, wi = w_ // 4, w_ % 4
        s = g * (4 * 64 * PPT) + (i_ * 4 + wi) * 64 + l_
    idx = order[s]                      # [W][PPT][64] original index
    P = p[idx].astype(np.float32)       # [W][PPT][64][3]
    pt = np.full((W, PPT, 64), 1e10, np.float32)
    def upd(o):
        d = ((P - o) ** 2).sum(-1).astype(np.float32)
        np.minimum(pt, d, out=pt)
    upd(p[0].astype(np.float32))
    j, rounds = 1, 0
    hist = np.zeros(rmax + 1, int)
    why = {'bound': 0, 'blocked': 0, 'rmax': 0, 'end': 0}
    while j < m:
        lane_best = pt.max(1)                                    # [W][64]
        lane_arg = pt.argmax(1)
        srt = np.sort(pt, axis=1)
        lane_second = srt[:, -2, :]
        ordl = np.argsort(-lane_best, axis=1, kind='stable')     # lanes by best desc
        ents = []
        wB = np.zeros(W, np.float32)
        for w in range(W):
            ls = ordl[w, :top]
            third = lane_best[w, ordl[w, top]]
            wB[w] = max(third, lane_second[w, ls].max())
            for l in ls:
                ents.append((lane_best[w, l], idx[w, lane_arg[w, l], l], w))
        ents.sort(key=lambda e: (-e[0], e[1]))
        gB = wB.max()
        if greedy:
            # the exact sequential selection over the PUBLISHED entries with their values updated by the pivots accepted so far: the best
            # remaining entry is the true next pick as long as its updated value is strictly above the bound of everything unpublished
            E = np.array([p[e[1]] for e in ents], np.float32)
            val = np.array([e[0] for e in ents], np.float32)
            key = np.array([e[1] for e in ents])
            alive = np.ones(len(ents), bool)
            acc, reason = [], 'rmax'
            while True:
                if len(acc) >= min(rmax, m - j): reason = 'rmax' if len(acc) >= rmax else 'end'; break
                cand = np.where(alive)[0]
                if len(cand) == 0: reason = 'entries'; break
                b = cand[np.lexsort((key[cand], -val[cand]))[0]]
                if acc and not (val[b] > gB): reason = 'bound'; break
                acc.append(ents[b]); alive[b] = False
                d = ((E - E[b]) ** 2).sum(-1).astype(np.float32)
                val = np.minimum(val, d)
            why[reason] = why.get(reason, 0) + 1
            hist[len(acc)] += 1
            for a in acc: upd(p[a[1]].astype(np.float32))
            j += len(acc); rounds += 1
            continue
        acc = [ents[0]]
        reason = 'rmax'
        for e in ents[1:]:
            if len(acc) >= min(rmax, m - j): reason = 'rmax' if len(acc) >= rmax else 'end'; break
            if not (e[0] > gB): reason = 'bound'; break
            pe = p[e[1]].astype(np.float32)
            blk = False
            for a in ents:
                if a is e: break
                if ((p[a[1]].astype(np.float32) - pe) ** 2).sum() < e[0]: blk = True; break
            if blk: reason = 'blocked'; break
            acc.append(e)
        else:
            reason = 'rmax'
        why[reason] += 1
        hist[len(acc)] += 1
        for a in acc: upd(p[a[1]].astype(np.float32))
        j += len(acc); rounds += 1
    return rounds, hist, why

def simulate_lazy(p, m, mode, PPT=16, W=16, rmax=16):
    """Which waves have to rebuild their published entries after a round?  mode 'touched': every wave one of whose tile boxes passed a
    pivot's box test (the kernel until round 5); 'E': only a wave whose published points E1 / E2 changed; 'Etile': a wave one of whose
    tiles HOLDING E1 / E2 was updated (what fps_spec_kernel does since round 5: no per-update cost).  The other waves keep their entries
    and their (stale, still valid) bound.  -> rounds, rebuilds per round, box-touched waves per round."""
    order = morton_order(p)
    w_, i_, l_ = np.meshgrid(np.arange(W), np.arange(PPT), np.arange(64), indexing='ij')
    idx = order[w_ * 64 * PPT + i_ * 64 + l_]
    P = p[idx].astype(np.float32)
    lo, hi = P.min(2), P.max(2)                                 # tile boxes [W][PPT][3]
    pt = np.full((W, PPT, 64), 1e10, np.float32)
    np.minimum(pt, ((P - p[0].astype(np.float32)) ** 2).sum(-1).astype(np.float32), out=pt)
    cache = [None] * W
    bound = np.full(W, np.inf, np.float32)                      # v1 at the last rebuild (what the box tests prune against)
    j = 1; rounds = 0; rebuilds = 0; touched_tot = 0
    def build(w):
        lane_best = pt[w].max(0); lane_arg = pt[w].argmax(0)
        lane_second = np.sort(pt[w], axis=0)[-2]
        ordl = np.argsort(-lane_best, kind='stable')
        ls = ordl[:2]
        wB = max(lane_best[ordl[2]], lane_second[ls].max())
        return [(lane_best[l], idx[w, lane_arg[l], l], w, (lane_arg[l], l)) for l in ls], wB, lane_best[ordl[0]]
    need = np.ones(W, bool)
    while j < m:
        for w in range(W):
            if need[w]:
                cache[w] = build(w); rebuilds += 1; bound[w] = cache[w][2]
        ents = sorted((e for w in range(W) for e in cache[w][0]), key=lambda e: (-e[0], e[1]))
        gB = max(c[1] for c in cache)
        acc = [ents[0]]
        for e in ents[1:]:
            if len(acc) >= min(rmax, m - j) or not (e[0] > gB): break
            pe = p[e[1]].astype(np.float32)
            blk = False
            for a in ents:
                if a is e: break
                if ((p[a[1]].astype(np.float32) - pe) ** 2).sum() < e[0]: blk = True; break
            if blk: break
            acc.append(e)
        old = pt.copy()
        touched = np.zeros(W, bool); tt = np.zeros((W, PPT), bool)
        for a in acc:
            o = p[a[1]].astype(np.float32)
            d = np.maximum(np.maximum(lo - o, o - hi), 0)
            tmask = (d * d).sum(-1) * 0.99999 < bound[:, None]
            touched |= tmask.any(1); tt |= tmask
            dd = ((P - o) ** 2).sum(-1).astype(np.float32)
            np.minimum(pt, np.where(tmask[:, :, None], dd, pt), out=pt)
        touched_tot += touched.sum()
        if mode == 'E':
            need = np.array([any(pt[w][e[3]] < old[w][e[3]] for e in cache[w][0]) for w in range(W)])
        elif mode == 'Etile':
            need = np.array([any(tt[w, e[3][0]] for e in cache[w][0]) for w in range(W)])
        else:
            need = touched
        j += len(acc); rounds += 1
    return rounds, rebuilds / rounds, touched_tot / rounds


if __name__ == '__main__':
    kind = sys.argv[1]; m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    make = S.lidar_scenes if kind == 'lidar' else S.scenes
    pts = make(2, 16384, seed0=0)
    if len(sys.argv) > 3 and sys.argv[3] == 'lazy':        # which waves rebuild their entries (round 5)
        for mode in ('touched', 'E', 'Etile'):
            r, rb, tc = simulate_lazy(pts[0][:, :3], m, mode)
            print(kind, mode, 'rounds', r, 'rebuilds per round %.2f' % rb, 'box-touched waves per round %.2f' % tc, flush=True)
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == 'greedy':      # the merge as an exact greedy selection over the published entries (not built)
        for g in (False, True):
            r, h, why = simulate(pts[0][:, :3], m, 'blocked', top=2, greedy=g)
            print(kind, 'blocked', 'greedy' if g else 'prefix', 'rounds', r, 'picks/round %.2f' % ((m - 1) / r), why, 'hist', h.tolist(), flush=True)
        sys.exit(0)
    for layout in ('blocked', 'quad', 'interleaved'):
        for top in (2,):
            r, h, why = simulate(pts[0][:, :3], m, layout, top=top)
            print(kind, layout, 'top', top, 'rounds', r, 'picks/round %.2f' % ((m - 1) / r), why, 'hist', h.tolist(), flush=True)
