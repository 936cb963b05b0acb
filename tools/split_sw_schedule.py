#!/usr/bin/env python
"""Simulate the ring / register schedule of csrc/gemm_split_sw.hip for one wave and check its invariants for a range of K:

  * every stream item 0 .. 9 nk - 1 is requested exactly once, in order, into slot (item mod 10), never an item that does not exist;
  * a slot is refilled only in a term AFTER the one whose reads emptied it (the term barrier separates them);
  * an item is read only after the counted wait in front of its term guarantees this wave's pieces have landed
    (vmcnt(N): at most the N youngest pieces outstanding; vmcnt retires in order);
  * every MFMA term multiplies the planes it claims to (the operand in the register slot is the right (K tile, plane)), and each of
    the six products of every K tile is formed exactly once.

The tables below mirror the kernel's `unit` / `term` lambdas; run after editing either.
"""
import sys

SLOTS = 10
FORMATS = {
    # bf16x3: nine items per K tile, six terms
    "bf16x3": dict(
        IPT=9, KIND=["A2", "B0a", "B0b", "A1", "B1a", "B1b", "A0", "B2a", "B2b"],
        PROD=[(2, 0), (1, 0), (1, 1), (0, 1), (0, 0), (0, 2)],                  # term -> (plane of A, plane of B)
        # term -> reads: (operand, j relative to the tile's item 0 (B: j of half a; half b = j + 1), register slot as function of PAR)
        READS={0: [("A", 3, lambda par: par ^ 1)], 1: [("B", 4, lambda par: 1)], 2: [("A", 6, lambda par: par)], 3: [],
               4: [("B", 7, lambda par: 1), ("A", 9, lambda par: par ^ 1)], 5: [("B", 10, lambda par: 0)]},
        REQ={0: [11, 12], 1: [13], 2: [14, 15], 3: [16], 4: [], 5: [17, 18, 19]},
        REQ_PENULT_MAX=17,                                                       # the penultimate tile requests items j <= this
        VM={"full": {0: 14, 1: 14, 2: 14, 3: None, 4: 14, 5: 10}, "penult": {0: 14, 1: 14, 2: 14, 3: None, 4: 14, 5: 10},
            "last": {0: 10, 1: 6, 2: 4, 3: None, 4: 0, 5: None}},
        SA=lambda t, par: par if (t == 0 or t >= 3) else par ^ 1, SB=lambda t: 1 if t in (2, 3, 5) else 0,
        PRO_A=(0, 2), PRO_ITEMS=10, PRO_EXTRA=10, PARITY=True),
    # f16x2: six items per K tile, three terms, no parity
    "f16x2": dict(
        IPT=6, KIND=["A1", "B0a", "B0b", "A0", "B1a", "B1b"],
        PROD=[(1, 0), (0, 0), (0, 1)],
        READS={0: [("A", 3, lambda par: 1)], 1: [("B", 4, lambda par: 1), ("A", 6, lambda par: 0)], 2: [("B", 7, lambda par: 0)]},
        REQ={0: [11, 12], 1: [13], 2: [14, 15, 16]},
        REQ_PENULT_MAX=11,
        VM={"full": {0: 14, 1: 12, 2: 10}, "penult": {0: 14, 1: 10, 2: 6}, "last": {0: 4, 1: 0, 2: None}},
        SA=lambda t, par: 0 if t == 0 else 1, SB=lambda t: 1 if t == 2 else 0,
        PRO_A=(0, 1), PRO_ITEMS=10, PRO_EXTRA=10, PARITY=False),
}


def simulate(nk, half=0, fmt="bf16x3"):
    F = FORMATS[fmt]
    IPT, KIND, PROD, READS, REQ = F["IPT"], F["KIND"], F["PROD"], F["READS"], F["REQ"]
    E = IPT * nk
    out = []                      # outstanding pieces of this wave, oldest first (item ids, two per item)
    landed = set()
    issued = []
    slot_item = {}
    read_term = {}                # item -> global term index in which it was read
    gterm = [0]

    def issue(item):
        assert item < E, f"nk={nk}: request of item {item} >= {E}"
        assert item == len(issued), f"nk={nk}: item {item} requested out of order (expected {len(issued)})"
        slot = item % SLOTS
        prev = slot_item.get(slot)
        if prev is not None:
            assert prev == item - SLOTS
            assert prev in read_term and read_term[prev] < gterm[0], f"nk={nk}: slot {slot} refilled with {item} before item {prev} was read + barrier"
        slot_item[slot] = item
        issued.append(item)
        out.extend([item, item])

    def wait(n):
        if n is None:
            return
        while len(out) > n:
            landed.add(out.pop(0))

    def read(item):
        assert item in issued, f"nk={nk}: item {item} read before requested"
        assert item not in [x for x in out], f"nk={nk}: item {item} read while its pieces may be in flight (term {gterm[0]})"
        assert slot_item[item % SLOTS] == item, f"nk={nk}: item {item} overwritten before its read"
        read_term[item] = gterm[0]

    regA, regB = {}, {}
    done = set()
    # prologue
    for j in range(F["PRO_ITEMS"]):
        issue(j)
    wait(18); gterm[0] += 1
    read(0); regA[0] = F["PRO_A"]
    wait(14); gterm[0] += 1
    issue(F["PRO_EXTRA"])
    read(1 + half); read(2 - half)          # (the other half is read by the other wave pair; both count as read for the slot logic)
    regB[0] = (0, 0)
    for kt in range(nk):
        par = (kt & 1) if F["PARITY"] else 0
        mode = "last" if kt == nk - 1 else "penult" if kt == nk - 2 else "full"
        base = IPT * kt
        for t in range(len(PROD)):
            wait(F["VM"][mode][t])
            gterm[0] += 1
            pa, pb = PROD[t]
            sa, sb = F["SA"](t, par), F["SB"](t)
            assert regA.get(sa) == (kt, pa), f"nk={nk} kt={kt} T{t}: A slot {sa} holds {regA.get(sa)}, wanted plane {pa}"
            assert regB.get(sb) == (kt, pb), f"nk={nk} kt={kt} T{t}: B slot {sb} holds {regB.get(sb)}, wanted plane {pb}"
            done.add((kt, pa, pb))
            for op, j, slotf in READS[t]:
                item = base + j
                if item >= E:
                    assert mode == "last"
                    continue
                if op == "A":
                    read(item)
                    assert KIND[item % IPT][0] == "A"
                    regA[slotf(par)] = (item // IPT, int(KIND[item % IPT][1]))
                    assert slotf(par) != sa, "A read into the slot the term multiplies from"
                else:
                    read(item + half); read(item + 1 - half)
                    assert KIND[item % IPT][0] == "B" and KIND[item % IPT][2] == "a"
                    regB[slotf(par)] = (item // IPT, int(KIND[item % IPT][1]))
                    assert slotf(par) != sb, "B read into the slot the term multiplies from"
            for j in REQ[t]:
                item = base + j
                if mode == "last" or (mode == "penult" and j > F["REQ_PENULT_MAX"]):
                    assert item >= E, f"nk={nk}: item {item} exists but is never requested"
                    continue
                issue(item)
    assert len(issued) == E and not [x for x in out if x not in landed and False]
    assert len(done) == len(PROD) * nk
    assert all(i in read_term for i in range(E)), "unread items"
    return True


if __name__ == "__main__":
    for fmt in FORMATS:
        for nk in range(2, 200, 1 if fmt == "f16x2" else 2):
            for half in (0, 1):
                simulate(nk, half, fmt)
        print(f"split_sw schedule ({fmt}): ring, counted waits and register slots consistent for K = 64 .. 6336")
