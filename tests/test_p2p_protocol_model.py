"""Model check of the peer-to-peer transport's protocol (multigridsolver_amd/csrc/comm_p2p.hip) on the CPU: sequence-numbered flag words, two
window slots per peer used alternately, NO acknowledgements.  The claim the kernel's header makes — a rank can never overwrite a slot its
neighbour has not unpacked yet, whatever the relative speeds — is checked here on an executable model of exactly that protocol, under
thousands of random interleavings and adversarial schedules (one rank as fast as possible, another as slow as possible), for the
communication patterns of the cycle: neighbour exchanges on several levels (not every pair takes part in every exchange), all-gathers and
all-reduces (every pair), K-cycle-like repeats of one level.  Also checked: without the second slot the same schedules DO corrupt data
(the model can tell), and asymmetric participation deadlocks (why a pair that communicates in one direction signals in both)."""
import random

import pytest


class Rank:
    """one rank = one stream of exchange kernels; each kernel has three phases the scheduler can interleave with other ranks' phases:
    push (stores into the peers' slots + publish the sequence number), wait (until every participating peer's number arrived), unpack"""

    def __init__(self, r, world, program, slots=2):
        self.r, self.world, self.program, self.slots = r, world, program, slots
        self.pc, self.phase = 0, "push"
        self.seq = [0] * world                                  # exchanges done with each peer (device memory in the product)
        self.flag = [0] * world                                 # flag[p]: sequence number peer p published here (own window)
        self.slot = [[None] * slots for _ in range(world)]      # slot[p][s]: what peer p stored for this rank
        self.unpacked = []

    def done(self):
        return self.pc >= len(self.program)


def step(ranks, r):
    """advance rank r by one phase if it can; returns False if it is blocked (waiting) or finished"""
    me = ranks[r]
    if me.done():
        return False
    op, peers = me.program[me.pc]
    if me.phase == "push":
        for p in peers:
            s = me.seq[p] % me.slots
            dst = ranks[p]
            # the store itself: whatever was in the slot is gone — if the peer has not unpacked it yet, that is the bug the model looks for
            dst.slot[r][s] = (op, r, me.seq[p])
            dst.flag[r] = me.seq[p] + 1
        me.phase = "wait"
        return True
    if me.phase == "wait":
        if all(me.flag[p] >= me.seq[p] + 1 for p in peers):
            me.phase = "unpack"
            return True
        return False
    for p in peers:                                             # unpack: the slot must hold THIS exchange's data of that peer
        s = me.seq[p] % me.slots
        got = me.slot[p][s]
        assert got == (op, p, me.seq[p]), f"rank {r} op {op}: slot of peer {p} holds {got}, expected {(op, p, me.seq[p])}"
        me.unpacked.append(got)
        me.seq[p] += 1
    me.pc += 1
    me.phase = "push"
    return True


def run(world, programs, schedule, slots=2, max_steps=200000):
    ranks = [Rank(r, world, programs[r], slots) for r in range(world)]
    for _ in range(max_steps):
        if all(k.done() for k in ranks):
            return ranks
        order = schedule(ranks)
        if not any(step(ranks, r) for r in order):
            raise RuntimeError("deadlock: " + str([(k.pc, k.phase) for k in ranks]))
    raise RuntimeError("did not finish")


def symmetric(programs):
    """participation must be symmetric: p in peers(r, op) <=> r in peers(p, op)"""
    for r, prog in enumerate(programs):
        for op, peers in prog:
            for p in peers:
                assert any(o == op and r in q for o, q in programs[p]), (r, p, op)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("kcycle", [False, True])
def test_two_slots_without_acknowledgements_never_lose_data(world, kcycle):
    # the exchanges of three cycles as every rank issues them: per sharded level a neighbour exchange on the way down (twice on level 2 with
    # kcycle: a K step visits the level twice), an all-gather for the tail, neighbour exchanges on the way up, an all-reduce per Krylov step;
    # symmetric neighbour sets per level: plane neighbours, with the pair (world-2, world-1) dropping out at level >= 3 (a shard whose coarse level lost its halo)
    def peers_of(r, lvl):
        out = []
        for p in (r - 1, r + 1):
            if 0 <= p < world:
                pair = (min(r, p), max(r, p))
                if lvl >= 3 and pair == (world - 2, world - 1) and world > 2:
                    continue
                out.append(p)
        return out
    programs = []
    for r in range(world):
        prog, k = [], 0
        everyone = [p for p in range(world) if p != r]
        for _ in range(3):
            for lvl in range(4):
                for _rep in range(2 if (kcycle and lvl == 2) else 1):
                    prog.append((("down", lvl, k), peers_of(r, lvl))); k += 1
            prog.append((("allgather", k), everyone)); k += 1
            for lvl in (3, 2, 1):
                prog.append((("up", lvl, k), peers_of(r, lvl))); k += 1
            prog.append((("allreduce", k), everyone)); k += 1
        programs.append([(op, peers) for op, peers in prog if peers])
    symmetric(programs)
    rng = random.Random(1234 + world)
    schedules = [lambda ranks: rng.sample(range(world), world) for _ in range(1)]
    # adversarial: rank f always first (runs ahead as far as the protocol lets it), rank s always last
    for f in range(world):
        for s in range(world):
            if f != s:
                schedules.append(lambda ranks, f=f, s=s: [f] + [q for q in range(world) if q not in (f, s)] + [s])
    for sched in schedules:
        ranks = run(world, programs, sched)
        for r, k in enumerate(ranks):
            assert len(k.unpacked) == sum(len(p) for _, p in programs[r])
    for seed in range(300):                       # random interleavings, phase by phase
        rs = random.Random(seed)
        run(world, programs, lambda ranks: rs.sample(range(world), world))


def test_model_detects_the_bug_a_single_slot_would_be():
    """the same schedules with ONE slot per peer: a rank that runs one exchange ahead overwrites data its neighbour has not unpacked — the model
    must see it, or its green runs above would mean nothing"""
    world = 2
    programs = [[(("x", k), [1 - r]) for k in range(6)] for r in range(world)]
    with pytest.raises(AssertionError):
        run(world, programs, lambda ranks: [0, 1], slots=1)       # rank 0 always first
    run(world, programs, lambda ranks: [0, 1], slots=2)


def test_asymmetric_participation_deadlocks_in_the_model():
    """if only the sender counted an exchange for the pair, the receiver would wait for a sequence number that never comes: the reason a pair that
    communicates in ONE direction still signals in both (comm_p2p.hip: participation is scnt > 0 or rcnt > 0 on both sides)"""
    programs = [[(("x", 0), [1])], [(("y", 0), [])]]
    programs[1] = []                                                # rank 1 does not take part
    with pytest.raises(RuntimeError):
        run(2, programs, lambda ranks: [0, 1])
