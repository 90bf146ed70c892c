"""Decode-order structure of a synthetic random-access (hierarchical-B) stream and its DPB slot plan.

This is the part of the host decoder that sits ABOVE the reconstruction stage (DecLib / PicListManager in the reference,
source/Lib/DecoderLib/DecLib.cpp:182-312, CommonLib/PicListManager.cpp:127-290): which picture is reconstructed when,
which DPB slots it references and where it is written.  It is measurement infrastructure for the pre-parsed stream
benchmark and the tests; a real integration keeps the reference's own DPB manager.
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class PicPlan:
    poc: int
    layer: int                      # temporal layer (0 = key pictures)
    slice_type: int                 # 0 B, 1 P, 2 I
    l0: List[int] = field(default_factory=list)     # POCs
    l1: List[int] = field(default_factory=list)
    is_ref: bool = True
    slot: int = -1
    ref_slots: Tuple[List[Tuple[int, int]], List[Tuple[int, int]]] = None   # ([(slot, poc)...], [(slot, poc)...])


def _hier(lo, hi, layer, out):
    """pictures strictly between two already decoded POCs lo < hi, in decode order"""
    if hi - lo < 2:
        return
    mid = (lo + hi) // 2
    out.append((mid, layer, lo, hi))
    _hier(lo, mid, layer + 1, out)
    _hier(mid, hi, layer + 1, out)


def ra_plan(num_frames, gop=16, seed_poc0_is_external=True, pool=0, intra_period=0, irap_lookahead=0):
    """Decode-order list of PicPlan for POC 0..num_frames-1 (num_frames - 1 must be a multiple of gop).

    POC 0 is the IRAP picture; with seed_poc0_is_external it is not part of the plan (the caller uploads it into slot 0).
    Key pictures (multiples of gop) reference the two previous key pictures; B pictures reference the nearest decoded
    picture on each side (L0 = [past, future], L1 = [future, past]), which is the usual RA configuration.

    pool = 0: a freed slot is reused at once (smallest DPB, what a memory-constrained host decoder does).  pool = N > 0: slots are
    taken round-robin from N slots, so that independent pictures of one temporal layer do not serialise on a write-after-read /
    write-after-write hazard of a shared slot when several pictures are in flight (HBM is not the scarce resource here).

    intra_period > 0: every key picture whose POC is a multiple of it is an IRAP picture (I slice, no references).
    irap_lookahead = N: IRAP pictures other than the first are placed N positions earlier in the SUBMISSION order (see below).
    """
    assert (num_frames - 1) % gop == 0
    assert intra_period % gop == 0
    plans = []
    if not seed_poc0_is_external:
        plans.append(PicPlan(0, 0, 2))
    last_irap = 0
    for k in range(gop, num_frames, gop):
        if intra_period and k % intra_period == 0:
            # IRAP (CRA-like: the B pictures before it in output order still reference the previous key picture)
            plans.append(PicPlan(k, 0, 2))
            last_irap = k
        else:
            prev_keys = [k - gop] + ([k - 2 * gop] if k - 2 * gop >= last_irap else [])
            plans.append(PicPlan(k, 0, 0, l0=list(prev_keys), l1=list(prev_keys)))
        inner = []
        _hier(k - gop, k, 1, inner)
        for (poc, layer, lo, hi) in inner:
            plans.append(PicPlan(poc, layer, 0, l0=[lo, hi], l1=[hi, lo]))
    if irap_lookahead:
        # an IRAP picture has no dependencies: a host that parses ahead hands it to the back-end before the pictures that precede
        # it in decoding order, so that its long intra wavefront overlaps with them instead of stalling everything that follows
        for i in range(1, len(plans)):
            if plans[i].slice_type == 2:
                j = max(1, i - irap_lookahead)
                plans.insert(j, plans.pop(i))
    # a picture is a reference if any later picture lists it
    last_use = {}
    for i, p in enumerate(plans):
        for r in p.l0 + p.l1:
            last_use[r] = i
    for p in plans:
        p.is_ref = p.poc in last_use
    # slot plan: a slot is free again after the last picture that references its content has been SUBMITTED; the
    # back-end itself orders the overwrite behind all readers (events), so the plan only needs decode-order lifetimes
    slot_of = {0: 0} if seed_poc0_is_external else {}
    free_at = {0: last_use.get(0, -1)} if seed_poc0_is_external else {}
    used = set(slot_of.values())
    max_slots = len(used)
    nxt = len(used) % pool if pool else 0
    for i, p in enumerate(plans):
        for s, until in list(free_at.items()):
            if until < i:
                used.discard(s)
                del free_at[s]
        if pool:
            s = nxt
            while s in used:
                s = (s + 1) % pool
            nxt = (s + 1) % pool
        else:
            s = 0
            while s in used:
                s += 1
        used.add(s)
        p.slot = s
        slot_of[p.poc] = s
        max_slots = max(max_slots, s + 1)
        free_at[s] = last_use.get(p.poc, i)
        p.ref_slots = ([(slot_of[r], r) for r in p.l0], [(slot_of[r], r) for r in p.l1])
    return plans, max_slots


def submission_order(plans, irap_lookahead):
    """Order in which a host that parses `irap_lookahead` pictures ahead hands the pictures of `plans` (decoding order, slots assigned, i.e. what
    ra_plan(..., irap_lookahead=0) returns) to the back-end: an IRAP picture depends on nothing, so it is submitted that many positions before its
    decoding-order position (its long intra wavefront then overlaps with the pictures it jumped).  -> list of indices into `plans`.

    The slot plan is the decoding-order one, so the two orders of one stream use the same picture descriptions.  That is only sound if the slot
    an IRAP writes is not read by a picture it jumps over (the back-end orders a writer behind the readers submitted BEFORE it): checked here."""
    order = list(range(len(plans)))
    if irap_lookahead <= 0:
        return order
    for i in range(1, len(plans)):
        if plans[i].slice_type == 2:
            pos = order.index(i)
            j = max(1, pos - irap_lookahead)
            jumped = order[j:pos]
            for k in jumped:
                slots = [s for lst in plans[k].ref_slots for (s, _) in lst] + [plans[k].slot]
                assert plans[i].slot not in slots, "IRAP POC %d would overwrite slot %d before POC %d has used it: more DPB slots (pool) needed for this look-ahead" % (plans[i].poc, plans[i].slot, plans[k].poc)
            order.insert(j, order.pop(pos))
    return order
