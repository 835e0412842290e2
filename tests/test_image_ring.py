"""The decision logic of rendering.allocate_image_ring, on the CPU with injected timings: "fast" is absolute
-- a launch that takes at most 0.98 x the fill_ of the same run -- the search never stops while it has not seen `count` fast candidates,
the ring is never aliased unless the caller allows it."""
import torch

from torchdrivesim_amd.rendering import allocate_image_ring

F, M, S, FILL = 7.07, 7.45, 8.3, 7.45          # ms: fast / in-between / slow launches and torch's fill_ (DESIGN_HISTORY.md section 4)


class FakeTimer:
    def __init__(self, launches, fill=FILL):
        self.launches, self.fill_ms, self.i = list(launches), fill, -1
        self.held = []

    def first_touch(self, buf):
        self.i += 1
        self.held.append(buf)
        return 20.0

    def launch(self, buf):
        return self.launches[self.i]

    def fill(self, buf):
        return self.fill_ms


def ring(launches, **kw):
    t = FakeTimer(launches, kw.pop('fill', FILL))
    bufs, rep = allocate_image_ring(None, (2, 3), torch.float32, 'cpu', count=2, candidates=kw.pop('candidates', 5), timer=t,
                                    alloc=lambda shape, dtype, device: torch.empty(shape, dtype=dtype), **kw)
    return bufs, rep, t


def test_two_slow_candidates_first_do_not_end_the_search():
    bufs, rep, t = ring([S, S, F, F, F])
    assert rep['launch_ms'] == [S, S, F, F] and rep['kept'] == [2, 3] and rep['fast'] == [False, False, True, True]
    assert bufs[0] is t.held[2] and bufs[1] is t.held[3] and not rep['aliased'] and rep['fill_ms'] == FILL


def test_all_slow_takes_the_best_distinct_buffers():
    bufs, rep, t = ring([S, S + 0.1, S, S + 0.2, S])
    assert len(rep['launch_ms']) == 5 and rep['fast'] == [False] * 5            # searched to the end
    assert len(set(rep['kept'])) == 2 and bufs[0] is not bufs[1] and not rep['aliased']
    assert sorted(rep['launch_ms'][i] for i in rep['kept']) == [S, S]


def test_an_in_between_buffer_is_not_fast():
    bufs, rep, t = ring([M, S, F], candidates=3)
    assert rep['fast'] == [False, False, True]
    assert rep['kept'] == [0, 2] and not rep['aliased']                          # the fast one and the best of the rest, distinct
    bufs, rep, t = ring([M, S, F], candidates=3, allow_aliasing=True)
    assert rep['kept'] == [2, 2] and rep['aliased'] and bufs[0] is bufs[1]       # only on request


def test_fast_candidates_end_the_search_at_once():
    bufs, rep, t = ring([F, F, S, S])
    assert rep['launch_ms'] == [F, F] and rep['kept'] == [0, 1]


def test_a_launch_that_is_not_write_bound_takes_the_first_buffers():
    bufs, rep, t = ring([5.1, 5.1, 5.1], fill=1.9)                              # the uint8 mode: 5.1 ms against a 1.9 ms fill
    assert rep['write_bound'] is False and rep['kept'] == [0, 1] and len(rep['launch_ms']) == 2 and rep['fast'] is None


def test_an_in_between_buffer_that_passes_the_fill_yardstick_loses_to_a_faster_one():
    """spread-out pages: fill_ takes 7.8 ms, so a 7.44 ms buffer (the 15/16 class) passes `0.98 x fill` -- but not `1.03 x the fastest launch seen`"""
    bufs, rep, t = ring([7.44, F, F, F], fill=7.8)
    assert rep['launch_ms'] == [7.44, F, F] and rep['fast'] == [False, True, True] and rep['kept'] == [1, 2]
    bufs, rep, t = ring([7.44, 7.45, 7.46, 7.44], fill=7.8, candidates=4)           # nothing faster exists: they are what this device offers
    assert rep['launch_ms'] == [7.44, 7.45] and rep['kept'] == [0, 1]


def test_short_launches_stop_after_one_more_candidate_and_say_that_the_yardstick_does_not_apply():
    """B = 256: a 12.9 GB launch (1.85 ms) never beats its fill_ (1.84 ms), whatever the buffer (all four candidates were built
    to then keep the first two).  Three agreeing candidates end the search; the report says why.  At the headline size the same ratios keep searching."""
    bufs, rep, t = ring([1.93, 1.89, 1.91, 1.92, 1.85], fill=1.868)                 # (the figures of a round-5 bench run)
    assert rep['launch_ms'] == [1.93, 1.89, 1.91] and rep['kept'] == [0, 1] and rep['yardstick'] == 'not applicable' and rep['write_bound']
    assert rep['fast'] == [False, False, False] and not rep['aliased']
    # candidates that do NOT agree (a slower placement class among them) are searched on
    bufs, rep, t = ring([1.85, 1.99, 1.85, 1.85, 1.85], fill=1.84)
    assert len(rep['launch_ms']) == 5 and rep['yardstick'] == 'fill' and sorted(rep['kept']) == [0, 2]
    # long launches in the dead band (slow pages at the headline size) are searched to the end, as before
    bufs, rep, t = ring([S, S, S, S, S])
    assert len(rep['launch_ms']) == 5 and rep['yardstick'] == 'fill'
    bufs, rep, t = ring([5.1, 5.1, 5.1], fill=1.9)
    assert rep['yardstick'] == 'not write-bound'
