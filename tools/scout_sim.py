#!/usr/bin/env python3
"""CPU model of the scout kernels' hint and resolve steps on one synthetic page: how often a tile's hinted entry is
the true one under several ways of voting, and how many rounds the resolve step needs under several rules for
taking offers.  (No GPU; the chain semantics are reduced to headers of one or two bytes, everything else ends a
chain, which is what a guess is allowed in the kernels.)

usage: scout_sim.py [kind = dict12 | dict20 | dict32 | adv17] [pages = 3]"""
import os
import sys
from collections import Counter, defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parquet2_amd import synth

T = 2048
DEAD = -1


def build(stream, w):
    """nxt[o]: the header behind the run whose header would begin at o (DEAD: the chain ends or is not followed)."""
    n = len(stream)
    b = stream.astype(np.int64)
    b1 = np.concatenate([b[1:], [0]])
    b2 = np.concatenate([b[2:], [0, 0]])
    two = (b & 0x80) != 0
    ind = np.where(two, (b & 0x7F) | (b1 << 7), b)
    c = np.where(two, 2, 1)
    longer = two & ((b1 & 0x80) != 0)
    rb = (w + 7) // 8
    body = np.where(ind & 1, (ind >> 1) * w, rb)
    nxt = np.arange(n) + c + body
    nxt = np.where(longer | (nxt > n), DEAD, nxt)
    return nxt


def tile_exits(nxt):
    n = len(nxt)
    ex = [DEAD] * n
    nx = nxt.tolist()
    for o in range(n - 1, -1, -1):
        t = nx[o]
        if t == DEAD:
            ex[o] = DEAD
        elif t >= n:
            ex[o] = n  # the end of the stream: the chain ends in order
        elif t // T != o // T:
            ex[o] = t
        else:
            ex[o] = ex[t]
    return ex


def true_entries(nxt, n_tiles):
    n = len(nxt)
    entry = [None] * n_tiles
    o = 0
    headers = 0
    while o != DEAD and o < n:
        k = o // T
        if entry[k] is None:
            entry[k] = o
        o = int(nxt[o])
        headers += 1
    return entry, headers


def hints_by_count(ex, n_tiles, n):
    """What the table kernel does: the exit most offsets of the tile before leave at (two votes at least)."""
    hint = [None] * n_tiles
    ranks = []
    tops = [None] * n_tiles
    for k in range(n_tiles - 1):
        cnt = Counter(e for e in ex[k * T:min((k + 1) * T, n)] if e != DEAD and (k + 1) * T <= e < (k + 2) * T)
        tops[k + 1] = cnt.most_common(4)
        if cnt:
            e, c = cnt.most_common(1)[0]
            if c >= 2:
                hint[k + 1] = e
    return hint, tops


def hints_by_span(ex, nxt, n_tiles, n):
    """Votes weighted with the bytes of the tile each voter's own run covers."""
    hint = [None] * n_tiles
    for k in range(n_tiles - 1):
        score = defaultdict(int)
        hi = min((k + 1) * T, n)
        for o in range(k * T, hi):
            e = ex[o]
            if e != DEAD and (k + 1) * T <= e < (k + 2) * T:
                score[e] += min(int(nxt[o]), hi) - o
        if score:
            hint[k + 1] = max(score.items(), key=lambda kv: kv[1])[0]
    return hint


def hints_pushed(ex, tops, n_tiles, depth, keep=4):
    """The best `keep` exits of the tile before, pushed through this tile's table and added to this tile's own votes,
    `depth` times over."""
    best = list(tops)
    for _ in range(depth):
        new = [None] * n_tiles
        for k in range(1, n_tiles):  # entries into tile k are exits of tile k - 1
            score = defaultdict(int)
            for e, c in (tops[k] or []):
                score[e] += c
            # what arrives in tile k - 1 (best[k - 1]) and leaves it into tile k
            for x, c in (best[k - 1] or []):
                if x < len(ex):
                    e = ex[x]
                    if e != DEAD and k * T <= e < (k + 1) * T:
                        score[e] += c
            new[k] = sorted(score.items(), key=lambda kv: -kv[1])[:keep]
        best = new
    hint = [None] * n_tiles
    for k in range(1, n_tiles):
        if best[k] and best[k][0][1] >= 2:
            hint[k] = best[k][0][0]
    return hint


def resolve(ex, hint, n_tiles, n, rule):
    """The Jacobi rounds of the resolve kernel.  rule: 'any' (every offer is taken, a reached tile's first, else the
    nearest), 'keep' (an unreached offer is not taken by a tile whose own evaluation has not ended its chain)."""
    cand = [0] + [hint[k] for k in range(1, n_tiles)]
    leave = [None] * n_tiles
    dirty = [c is not None for c in cand]
    rounds = 0
    while True:
        rounds += 1
        for k in range(n_tiles):
            if dirty[k]:
                leave[k] = ex[cand[k]] if cand[k] < len(ex) else DEAD
                dirty[k] = False
        # reached tiles
        reached = [False] * n_tiles
        k = 0
        while True:
            reached[k] = True
            e = leave[k]
            if e is None or e == DEAD or e >= n:
                break
            m = e // T
            if m >= n_tiles:
                break
            k = m
            if leave[k] is None and cand[k] is None:
                break
            if reached[k]:
                break
        offers = {}
        for k in range(n_tiles):
            e = leave[k]
            if e is None or e == DEAD or e >= n:
                continue
            m = e // T
            if m >= n_tiles:
                continue
            key = (1 if reached[k] else 0, k)
            if m not in offers or offers[m][0] < key:
                offers[m] = (key, e)
        changed = False
        for m, (key, e) in offers.items():
            if cand[m] == e:
                continue
            if rule == 'keep' and key[0] == 0 and cand[m] is not None and leave[m] not in (None, DEAD):
                continue
            cand[m] = e
            dirty[m] = True
            if key[0]:
                changed = True
        if not changed:
            return rounds


def page(kind, seed):
    n = 1 << 20
    if kind.startswith("dict"):
        w = int(kind[4:])
        s, _ = synth.dict_page(synth.seed_for(2, seed), n, w)
    else:
        w = int(kind[3:])
        s, _ = synth.adversarial(synth.seed_for(4, seed), n, w)
    return s, w


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "dict12"
    pages = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    for p in range(pages):
        s, w = page(kind, p)
        n = len(s)
        n_tiles = (n + T - 1) // T
        nxt = build(s, w)
        ex = tile_exits(nxt)
        entry, headers = true_entries(nxt, n_tiles)
        entered = [k for k in range(1, n_tiles) if entry[k] is not None]
        hint, tops = hints_by_count(ex, n_tiles, n)
        variants = {"count": hint, "span": hints_by_span(ex, nxt, n_tiles, n)}
        for d in (1, 2, 3):
            variants["pushed%d" % d] = hints_pushed(ex, tops, n_tiles, d)
        in_top4 = sum(1 for k in entered if tops[k] and entry[k] in [e for e, _ in tops[k]])
        print("%s page %d: %d bytes, %d tiles (%d entered), %d headers; the true entry is among the 4 most voted in %.1f%%"
              % (kind, p, n, n_tiles, len(entered), headers, 100.0 * in_top4 / len(entered)))
        for name, h in variants.items():
            wrong = sum(1 for k in entered if h[k] != entry[k])
            none = sum(1 for k in entered if h[k] is None)
            line = "  hints by %-8s wrong %5.1f%% (none %4.1f%%)" % (name, 100.0 * wrong / len(entered), 100.0 * none / len(entered))
            for rule in ("any", "keep"):
                line += "  rounds[%s] %3d" % (rule, resolve(ex, h, n_tiles, n, rule))
            print(line)


if __name__ == "__main__":
    main()
