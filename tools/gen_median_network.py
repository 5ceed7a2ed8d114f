#!/usr/bin/env python3
"""Emit repet-python_amd/csrc/median_networks.inc: compare-exchange networks that place the lower
and upper median of N values at positions N/2-1 and N/2.

Each network is a sorting network for N wires -- Parberry's pairwise network or Batcher's odd-even merge sort,
whichever leaves fewer instructions -- pruned by a backward liveness pass to the comparators that can influence
those two outputs. Both have the same size as sorters, but the pairwise network moves its long-range comparators
to the front and therefore loses far more of them to the pruning (N = 100: 776 comparators / 1 426 instructions
against 889 / 1 652). A wave evaluates one network per 64 frequency bins
with the N values of each lane in registers (static indices only), which is how the gfx950 mask
kernels compute np.median (repet.py:872,1421,1427,1496,1535) without LDS or scratch traffic.

Lists shorter than N are centred with -1 / +inf pads by the kernel, so the medians stay at N/2-1, N/2.
"""
import os
import random
import sys

SIZES = (2, 4, 8, 10, 12, 16, 24, 32, 48, 64, 80, 100, 128)


def batcher(n):
    """Comparator list of Batcher's odd-even merge sort for any n (Knuth 5.2.2 Algorithm M)."""
    ces = []
    t = 1
    while (1 << t) < n:
        t += 1
    p = 1 << (t - 1)
    while p > 0:
        q = 1 << (t - 1)
        r = 0
        d = p
        while True:
            for i in range(n - d):
                if (i & p) == r:
                    ces.append((i, i + d))
            if q == p:
                break
            d = q - p
            q >>= 1
            r = p
        p >>= 1
    return ces


def pairwise(n):
    """Comparator list of Parberry's pairwise sorting network (1992) for any n: the network of the next power of
    two without the comparators that touch a wire >= n (all comparators put the minimum on the lower wire, so
    +inf on the dropped wires would never move)."""
    m = 1
    while m < n:
        m *= 2
    ces = []

    def sweep(a, d):                        # runs of `a` wires, every other run: wire b against wire b - d * a
        b, c = (d + 1) * a if d > 0 else a, 0
        while b < m:
            ces.append((b - (d if d > 0 else 1) * a, b))
            b += 1
            c += 1
            if c >= a:
                c = 0
                b += a

    a = 1
    while a < m:                            # pairs, pairs of pairs, ...: wire b against wire b - a
        sweep(a, 0)
        a *= 2
    a //= 4
    e = 1
    while a > 0:                            # the merges, finest runs last
        d = e
        while d > 0:
            sweep(a, d)
            d //= 2
        a //= 2
        e = e * 2 + 1
    return [(i, j) for (i, j) in ces if j < n]


def instruction_count(ops):
    return sum(2 if o[0] == "ce" else 1 for o in ops)


def prune(ces, outputs):
    live = set(outputs)
    kept = []
    for (i, j) in reversed(ces):
        if i in live or j in live:
            kept.append((i, j))
            live.add(i)
            live.add(j)
    kept.reverse()
    return kept


def lower(n, ces):
    """Comparators -> instructions. A comparator whose low (high) output is never read again keeps only
    its max (min); two dependent min-only (max-only) steps fuse into one v_min3 (v_max3).
    Returns a list of (op, dst, srcs) with op in {'ce','min','max','min3','max3'}; 'ce' has dst (lo, hi)."""
    live = {n // 2 - 1, n // 2}
    ops = []
    for (i, j) in reversed(ces):
        lo_live, hi_live = i in live, j in live
        if not (lo_live or hi_live):
            continue
        if lo_live and hi_live:
            ops.append(["ce", (i, j), (i, j)])
        elif lo_live:
            ops.append(["min", i, (i, j)])
        else:
            ops.append(["max", j, (i, j)])
        live.add(i)
        live.add(j)
    ops.reverse()
    # fuse  x = min(a,b) ; y = min(x,c)  ->  y = min3(a,b,c)  when x is read by nothing else
    out = []
    skip = set()
    for k, op in enumerate(ops):
        if k in skip:
            continue
        kind, dst, srcs = op
        if kind in ("min", "max"):
            # next op that touches wire dst
            nxt = None
            for m in range(k + 1, len(ops)):
                if m in skip:
                    continue
                if dst in ops[m][2]:
                    nxt = m
                    break
            if nxt is not None and ops[nxt][0] == kind:
                # no op between k and nxt may touch the source wires of op k (they must still hold their values)
                clean = all(not (set(srcs) & set(ops[m][2])) for m in range(k + 1, nxt) if m not in skip)
                if clean:
                    other = [w for w in ops[nxt][2] if w != dst]
                    out_dst = ops[nxt][1]
                    ops[nxt] = [kind + "3", out_dst, (srcs[0], srcs[1], other[0])]
                    continue        # op k disappears; the fused op is emitted at position nxt
        out.append(op)
    return out


def simulate(n, ops, a):
    a = list(a)
    for kind, dst, srcs in ops:
        if kind == "ce":
            lo, hi = min(a[srcs[0]], a[srcs[1]]), max(a[srcs[0]], a[srcs[1]])
            a[dst[0]], a[dst[1]] = lo, hi
        elif kind == "min":
            a[dst] = min(a[srcs[0]], a[srcs[1]])
        elif kind == "max":
            a[dst] = max(a[srcs[0]], a[srcs[1]])
        elif kind == "min3":
            a[dst] = min(a[srcs[0]], a[srcs[1]], a[srcs[2]])
        else:
            a[dst] = max(a[srcs[0]], a[srcs[1]], a[srcs[2]])
    return a


def check_ops(n, ops, trials=400):
    rnd = random.Random(1000 + n)
    for _ in range(trials):
        if rnd.random() < 0.3:
            a = [rnd.randint(0, 3) for _ in range(n)]
        else:
            a = [rnd.random() for _ in range(n)]
        want = sorted(a)
        got = simulate(n, ops, a)
        assert got[n // 2 - 1] == want[n // 2 - 1] and got[n // 2] == want[n // 2], n


def check_sorter(n, ces):
    """0-1 principle, bit-parallel (a comparator on 0-1 values is AND / OR): exhaustive up to 16 wires, otherwise
    every input with one run of ones and 65 536 random ones of all densities."""
    import numpy as np
    if n <= 16:
        codes = np.arange(1 << n, dtype=np.uint64)
        bits = [((codes >> np.uint64(i)) & np.uint64(1)).astype(np.uint8) for i in range(n)]
    else:
        rng = np.random.default_rng(77 + n)
        runs = np.array([[1 if lo <= i < hi else 0 for i in range(n)] for lo in range(n) for hi in range(lo, n + 1)], dtype=np.uint8)
        dens = rng.random((1 << 16, 1))
        rand = (rng.random((1 << 16, n)) < dens).astype(np.uint8)
        allv = np.concatenate([runs, rand])
        bits = [np.ascontiguousarray(allv[:, i]) for i in range(n)]
    for (i, j) in ces:
        bits[i], bits[j] = bits[i] & bits[j], bits[i] | bits[j]
    for i in range(n - 1):
        assert not np.any(bits[i] > bits[i + 1]), n


def check(n, ces, trials=300):
    rnd = random.Random(n)
    for _ in range(trials):
        kind = rnd.random()
        if kind < 0.3:
            a = [rnd.randint(0, 3) for _ in range(n)]      # many ties
        else:
            a = [rnd.random() for _ in range(n)]
        want = sorted(a)
        for (i, j) in ces:
            if a[i] > a[j]:
                a[i], a[j] = a[j], a[i]
        assert a[n // 2 - 1] == want[n // 2 - 1] and a[n // 2] == want[n // 2], n


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "repet-python_amd", "csrc",
                       "median_networks.inc")
    lines = ["// GENERATED by tools/gen_median_network.py -- do not edit.",
             "// REPET_NET<N>::run(a): afterwards a[N/2-1] <= a[N/2] are the two middle order statistics.",
             "// kLoadOrder lists the wires in the order the network first reads them: gathering in that order lets the",
             "// first comparators start while the tail of the gather is still in flight.",
             "// The file is included once per element type, with the includer defining",
             "//   REPET_NET (struct name), REPET_T (register type) and REPET_OP_MIN / MAX / MIN3 / MAX3 (mnemonics):",
             "//   * MedianNet   : float magnitudes compared as signed integers (v_min_i32 ...): every value is a non-negative",
             "//                   float, +inf or the -1.0f pad, for which signed-integer order equals float order;",
             "//   * MedianNetPk : TWO 16-bit rank codes per register (v_pk_min_u16 ... and gfx950's v_pk_minimum3_f16 /",
             "//                   v_pk_maximum3_f16): codes are positive normal f16 bit patterns, 0 or +inf, for which",
             "//                   unsigned-integer order equals f16 order, so both families agree.",
             "// The asm is volatile so the steps issue in network order: the live set stays at N values + 1 temporary (hipcc",
             "// otherwise stretches live ranges to ~2N registers and halves the occupancy), and no NaN",
             "// canonicalisation ops are inserted in front of the min/max. Steps whose other output is dead keep",
             "// one instruction; dependent min-only / max-only pairs are fused into the three-input forms.",
             "#define REPET_CE(i, j) { REPET_T lo_, hi_; asm volatile(REPET_OP_MIN \" %0, %2, %3\\n\\t\" REPET_OP_MAX \" %1, %2, %3\" : \"=&v\"(lo_), \"=v\"(hi_) : \"v\"(a[i]), \"v\"(a[j])); a[i] = lo_; a[j] = hi_; }",
             "#define REPET_MIN(d, i, j) { REPET_T r_; asm volatile(REPET_OP_MIN \" %0, %1, %2\" : \"=v\"(r_) : \"v\"(a[i]), \"v\"(a[j])); a[d] = r_; }",
             "#define REPET_MAX(d, i, j) { REPET_T r_; asm volatile(REPET_OP_MAX \" %0, %1, %2\" : \"=v\"(r_) : \"v\"(a[i]), \"v\"(a[j])); a[d] = r_; }",
             "#define REPET_MIN3(d, i, j, k) { REPET_T r_; asm volatile(REPET_OP_MIN3 \" %0, %1, %2, %3\" : \"=v\"(r_) : \"v\"(a[i]), \"v\"(a[j]), \"v\"(a[k])); a[d] = r_; }",
             "#define REPET_MAX3(d, i, j, k) { REPET_T r_; asm volatile(REPET_OP_MAX3 \" %0, %1, %2, %3\" : \"=v\"(r_) : \"v\"(a[i]), \"v\"(a[j]), \"v\"(a[k])); a[d] = r_; }",
             "template <int N> struct REPET_NET;"]
    for n in SIZES:
        best = None
        for name, full in (("pairwise", pairwise(n)), ("batcher", batcher(n))):
            check_sorter(n, full)
            ces = prune(full, (n // 2 - 1, n // 2))
            check(n, ces)
            ops = lower(n, ces)
            check_ops(n, ops)
            if best is None or instruction_count(ops) < instruction_count(best[3]):
                best = (name, full, ces, ops)
        name, full, ces, ops = best
        n_instr = instruction_count(ops)
        print(f"N={n}: {name}, {len(full)} comparators, {len(ces)} after pruning, {n_instr} instructions "
              f"({2 * len(ces)} before output-level pruning and min3/max3 fusion)", file=sys.stderr)
        lines.append(f"template <> struct REPET_NET<{n}> {{")
        lines.append(f"    static constexpr int kInstructions = {n_instr};")
        first = []
        for kind, dst, srcs in ops:                 # wires in the order the network first reads them
            for w in srcs:
                if w not in first:
                    first.append(w)
        first += [w for w in range(n) if w not in first]
        assert sorted(first) == list(range(n)), n
        lines.append(f"    static constexpr unsigned char kLoadOrder[{n}] = {{{', '.join(str(w) for w in first)}}};")
        lines.append(f"    static __device__ __forceinline__ void run(REPET_T (&a)[{n}]) {{")
        row = []
        for kind, dst, srcs in ops:
            if kind == "ce":
                row.append(f"REPET_CE({srcs[0]},{srcs[1]})")
            elif kind in ("min", "max"):
                row.append(f"REPET_{kind.upper()}({dst},{srcs[0]},{srcs[1]})")
            else:
                row.append(f"REPET_{kind.upper()}({dst},{srcs[0]},{srcs[1]},{srcs[2]})")
            if len(row) == 6:
                lines.append("        " + " ".join(row))
                row = []
        if row:
            lines.append("        " + " ".join(row))
        lines.append("    }")
        lines.append("};")
    for m in ("CE", "MIN", "MAX", "MIN3", "MAX3"):
        lines.append(f"#undef REPET_{m}")
    text = "\n".join(lines) + "\n"
    if "--check" in sys.argv:                 # build(): the committed file must be what this generator emits
        if not os.path.exists(out) or open(out).read() != text:
            raise SystemExit(f"{out} is not what tools/gen_median_network.py generates: regenerate and commit it")
        return
    if not os.path.exists(out) or open(out).read() != text:   # keep the mtime when nothing changed
        with open(out, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main()
