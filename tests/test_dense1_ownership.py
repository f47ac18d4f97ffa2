"""The ownership rule of the super-block dense CG (csrc/cg_dense1.hip, d1_persist_blk_kernel, "ownership, by (chunk,
column)") restated in Python and checked exhaustively: for every geometry the kernel accepts -- S = ceil(nt / 3) super-rows
from 6 to 22, ragged nt, 1..8 columns (7 and 8 from S = 12 on) -- every (chunk, column) item has exactly ONE owner, the
owner is a workgroup that holds p of that chunk (as one of its three row chunks or its three column chunks), and no
workgroup owns more items than its LDS arrays hold.  A hole here would not give wrong numbers on the GPU -- an unowned
item makes the solve time out and fail over to two launches per iteration -- it would silently cost the fast path."""


def items(S, nt, BT, spread, SI, SJ):
    E = 1 if (not spread and BT <= 6) else min(BT, S // 3)
    dl = SJ - SI
    out = []
    if dl <= 3 * E - 1 and 3 * SI + dl % 3 < nt:
        out += [(3 * SI + dl % 3, e, dl % 3) for e in range(dl // 3, BT, E)]
    dw = S - dl
    if dl > 0 and dw <= 3 * E - 1 and 3 * SJ + dw % 3 < nt:
        out += [(3 * SJ + dw % 3, e, 3 + dw % 3) for e in range(dw // 3, BT, E)]
    return out


def test_every_item_has_one_owner_that_holds_its_p():
    for S in range(6, 23):
        for nt in range(3 * S - 2, 3 * S + 1):
            for BT in range(1, 9):
                if BT > 6 and S < 12:
                    continue  # d1_persist_form: 7, 8 columns from S = 12 on
                NI = 6 if BT >= 6 else (2 if BT < 2 else 2 * ((BT + 1) // 2))
                for spread in (0, 1):
                    owned = {}
                    for SI in range(S):
                        for SJ in range(SI, S):
                            mine = items(S, nt, BT, spread, SI, SJ)
                            assert len(mine) <= NI <= 6, (S, nt, BT, spread, SI, SJ, mine)
                            for c, e, p in mine:
                                assert (p < 3 and c == 3 * SI + p) or (p >= 3 and c == 3 * SJ + p - 3)
                                assert (c, e) not in owned, (S, nt, BT, spread, c, e)
                                owned[(c, e)] = (SI, SJ)
                    assert len(owned) == nt * BT, (S, nt, BT, spread)


def test_every_chunk_collects_its_s_plus_one_vectors_exactly_once():
    """Block (SI, SJ) publishes, per column, one row vector per tile row 3 SI + g into slot SJ of that chunk and one
    column vector per tile column 3 SJ + ly into slot SI (slot S from a diagonal block): the owner of a chunk polls
    slots 0..S, so each must be written by exactly one block."""
    for S in range(6, 23):
        for nt in range(3 * S - 2, 3 * S + 1):
            got = {}
            for SI in range(S):
                for SJ in range(SI, S):
                    for g in range(3):
                        if 3 * SI + g < nt:
                            key = (3 * SI + g, SJ)
                            assert key not in got
                            got[key] = (SI, SJ)
                    for ly in range(3):
                        if 3 * SJ + ly < nt:
                            key = (3 * SJ + ly, S if SI == SJ else SI)
                            assert key not in got
                            got[key] = (SI, SJ)
            assert set(got) == {(c, j) for c in range(nt) for j in range(S + 1)}, (S, nt)
