"""Model of the hit-stash control flow of rt_trace_kernel (kStash variants, rt_kernels.h): one wave, 64 lanes, the same steps
in the same order — stash step (pop, decide, push or process), fresh paths when every lane is idle, exit check, scan,
transitions — with random scan and hit-processing outcomes.  Checked for every parameter set, adversarial ones included
(every hit ends its path while only part of the scans hit: the case in which the first version left the loop with records
in the stash — with `break` in place of `continue` below, four of these parameter sets fail):

  * the loop ends, and only when every lane is idle, the stash is empty and the queue is drained;
  * every path handed out is finished exactly once (no record lost, none processed twice);
  * a push only ever happens into an EMPTY stash and never exceeds its capacity;
  * hits are processed by more than `cap` lanes while fresh paths remain.

No GPU: this is the exit condition every wave reaches, stated as a test.

NOTE: this file exercises a MODEL of the policy, written from rt_kernels.h; an edit to the kernel's push / pop /
process order does not run through it.  It documents the invariants; the authority for the shipped code are the GPU regression
tests -- test_hits_left_in_the_stash_when_every_lane_finishes_are_not_lost, the stash-capacity knobs of
test_launch_and_layout_knobs_give_the_same_bits / test_hierarchy_scan_knobs_give_the_same_bits, and the full-size digests."""
import random

import pytest

IDLE, NEED_CLOSEST, NEED_SHADOW, HAVE_HIT = 0, 1, 2, 3


def run_wave(cap, n_paths, block, p_hit, p_finish, p_far, seed, max_iters=200000):
    rng = random.Random(seed)
    state = [IDLE] * 64
    path = [None] * 64
    stash = []                      # records = path ids
    blocks = [(b, min(b + block, n_paths)) for b in range(0, n_paths, block)]
    blk_next = blk_end = 0
    queue_empty = False
    finished = {}
    processed_at = []               # lanes holding a hit at each processing pass, with whether fresh paths remained
    iters = 0

    def next_block():
        nonlocal blk_next, blk_end
        if not blocks:
            return False
        blk_next, blk_end = blocks.pop(0)
        return True

    def finish(lane):
        assert path[lane] not in finished, "path finished twice"
        finished[path[lane]] = True
        state[lane], path[lane] = IDLE, None

    def process_hit(lane):
        # Scatter, shadow query, shade: the path continues, ends, or (far hit point) needs a shadow scan first
        u = rng.random()
        if u < p_far:
            state[lane] = NEED_SHADOW
        elif rng.random() < p_finish:
            finish(lane)
        else:
            state[lane] = NEED_CLOSEST

    while True:
        iters += 1
        assert iters < max_iters, "the wave does not terminate"
        # ---- stash step (1): idle lanes take stashed hits
        idle = [l for l in range(64) if state[l] == IDLE]
        n = min(len(stash), len(idle))
        for l in idle[:n]:
            path[l] = stash.pop()
            state[l] = HAVE_HIT
        # ---- (2) process, push, or wait
        hit = [l for l in range(64) if state[l] == HAVE_HIT]
        need_scan = any(s in (NEED_CLOSEST, NEED_SHADOW) for s in state)
        process = len(hit) > cap
        if not process and hit and not need_scan:
            if blk_next == blk_end and not queue_empty and not next_block():
                queue_empty = True
            if blk_next == blk_end:
                process = True
            else:
                assert not stash, "push into a stash that is not empty"
                assert len(hit) <= cap
                for l in hit:
                    stash.append(path[l])
                    state[l], path[l] = IDLE, None
        if process:
            processed_at.append((len(hit), blk_next != blk_end or bool(blocks)))
            for l in hit:
                process_hit(l)
        # ---- (3) fresh paths when every lane is idle
        if all(s == IDLE for s in state):
            if blk_next == blk_end and not queue_empty and not next_block():
                queue_empty = True
            if blk_next != blk_end:
                n_gen = min(64, blk_end - blk_next)
                for l in range(n_gen):
                    state[l], path[l] = NEED_CLOSEST, blk_next + l
                blk_next += n_gen
        # ---- exit check
        if all(s == IDLE for s in state):
            if stash:
                continue            # (the first version broke out here)
            break
        # ---- scan + transitions
        for l in range(64):
            if state[l] == NEED_CLOSEST:
                if rng.random() < p_hit:
                    state[l] = HAVE_HIT
                else:
                    finish(l)       # sky
            elif state[l] == NEED_SHADOW:
                if rng.random() < p_finish:
                    finish(l)
                else:
                    state[l] = NEED_CLOSEST
    assert not stash and queue_empty and not blocks and blk_next == blk_end
    assert len(finished) == n_paths and set(finished) == set(range(n_paths))
    return iters, processed_at


@pytest.mark.parametrize("cap", [16, 31, 44, 63])
@pytest.mark.parametrize("p_hit,p_finish,p_far", [(0.59, 0.06, 0.0), (1.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.5, 0.0), (0.59, 0.06, 0.3),
                                                  (0.95, 0.5, 1.0), (0.3, 0.9, 0.05), (0.7, 1.0, 0.0), (0.85, 1.0, 0.0), (0.5, 1.0, 0.0)])
def test_a_wave_finishes_every_path_once_and_leaves_with_an_empty_stash(cap, p_hit, p_finish, p_far):
    for seed, (n_paths, block) in enumerate([(128, 128), (3 * 256 + 51, 256), (64, 128), (1, 128), (1000, 128)]):
        if p_hit == 1.0 and p_finish == 0.0:
            n_paths = min(n_paths, 200)  # paths that never end by themselves: the model's depth limit is the finish draw
            p_fin = 0.02
        else:
            p_fin = p_finish
        iters, passes = run_wave(cap, n_paths, block, p_hit, p_fin, p_far, seed=17 * cap + seed)
        # while fresh paths remained, hit processing ran with more than `cap` lanes
        assert all(lanes > cap for lanes, fresh_left in passes if fresh_left)
