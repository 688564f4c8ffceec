"""The oracle's jump tables (oracle.c: cst_oracle_{ans,rc}_jump_table / _decode_from = the reference's Pos / Seek,
stack.rs:1107-1139, queue.rs:172-196, 900-926) against the oracle's own single coders and against the two-part messages of the
reference's Python seek tests (tests/python/test_docexamples.py:403-426 `test_ans_seek`, :620-642 `test_range_coding_seek`:
a checkpoint between two parts of a message, decoding resumes at part 2)."""
import numpy as np
import pytest

from oracle import oracle as O

PART1, PART2 = [1, 2, 0, 3, 2, 3, 0], [2, 2, 0, 1, 3]           # the messages of the reference's seek tests
PROBS = [0.2, 0.4, 0.1, 0.3]


def test_reference_seek_scenarios_through_the_jump_tables():
    P = 24
    cdf = O.categorical_fast_cdf(np.array(PROBS), P)
    # ANS: part 2 is encoded first (a stack); the checkpoint sits between the parts.  With chunks of 7 symbols over
    # part1 ++ part2[:7 - ...] the reference's checkpoint is the jump point in front of symbol 7 of part1 ++ part2.
    msg = np.array([PART1 + PART2 + [0, 0]], np.int32)             # two chunks of 7 (two filler symbols behind part 2)
    pos, state = O.ans_jump_table(msg, 0, cdf, P, 7)
    c = O.AnsCoder()
    c.encode_iid_table_reverse(msg[0, 7:], cdf, 0, P)
    assert c.pos() == (int(pos[0, 1]), int(state[0, 1]))           # coder.pos() after part 2 (+ filler), before part 1
    words, n, _ = O.ans_encode_batch(msg, 0, cdf, P)
    assert O.ans_decode_from(words[0], pos[0, 1], state[0, 1], 5, 0, cdf, P).tolist() == PART2
    assert O.ans_decode_from(words[0], pos[0, 0], state[0, 0], 1, 0, cdf, P).tolist() == [1]
    # range coder: a queue, part 1 first
    msg = np.array([PART1 + PART2], np.int32)
    rpos, lower, rng = O.range_jump_table(msg, 0, cdf, P, 7)
    e = O.RangeEncoder()
    e.encode(msg[0, :7], O.TableModel(cdf, 0, P), P)
    assert e.pos() == (int(rpos[0, 1]), (int(lower[0, 1]), int(rng[0, 1])))
    rw, rn, _ = O.rc_encode_batch(msg, 0, cdf, P)
    got, st = O.range_decode_from(rw[0, : rn[0]], rpos[0, 1], lower[0, 1], rng[0, 1], 5, 0, cdf, P)
    assert st == 0 and got.tolist() == PART2
    assert (int(rpos[0, 0]), int(lower[0, 0]), int(rng[0, 0])) == (0, 0, 2**64 - 1)


@pytest.mark.parametrize("W,S,P", [(32, 64, 12), (32, 64, 24), (16, 32, 12)])
@pytest.mark.parametrize("n,interval", [(96, 32), (100, 1), (90, 100), (200, 64)])
def test_jump_tables_against_the_single_coders(W, S, P, n, interval):
    cdf = O.GaussianModel(-30, 30, 1.5, 6.0, P, 32 if W == 32 else 16).cdf_table()
    sym = O.synth_symbols(5, 0, 4, n, -30, cdf, P)
    pos, state = O.ans_jump_table(sym, -30, cdf, P, interval, W, S)
    words, nw, _ = O.ans_encode_batch(sym, -30, cdf, P, W, S)
    rpos, lower, rng = O.range_jump_table(sym, -30, cdf, P, interval, W, S)
    rw, rn, _ = O.rc_encode_batch(sym, -30, cdf, P, W, S)
    for s in range(4):
        for j in range((n + interval - 1) // interval):
            chunk = sym[s, j * interval:(j + 1) * interval]
            c = O.AnsCoder(W=W, S=S)
            c.encode_iid_table_reverse(sym[s, j * interval:], cdf, -30, P)
            assert c.pos() == (int(pos[s, j]), int(state[s, j]))
            assert O.ans_decode_from(words[s], pos[s, j], state[s, j], len(chunk), -30, cdf, P, W, S).tolist() == chunk.tolist()
            e = O.RangeEncoder(W=W, S=S)
            e.encode(sym[s, : j * interval], O.TableModel(cdf, -30, P), P)
            assert e.pos() == (int(rpos[s, j]), (int(lower[s, j]), int(rng[s, j])))
            got, st = O.range_decode_from(rw[s, : rn[s]], rpos[s, j], lower[s, j], rng[s, j], len(chunk), -30, cdf, P, W, S)
            assert st == 0 and got.tolist() == chunk.tolist()
