"""CPU checks of the float64 recogniser's weight packing against the MFMA operand layouts measured on the GPU
(tools/ubench/mfma_f64.hip -> profiles/r04_mfma_f64.txt; tools/ubench/mfma_f64_4x4.hip -> profiles/r05_mfma_f64_4x4.txt):
a numpy model of the two matrix instructions and of the gfx950 row swaps replays one step's pre-activations from the
packed fragments exactly as csrc/ta_lstm_f64.hip feeds them, and must give W . h."""
import numpy as np
import pytest


def _mfma_f64_16x16x4(a, b, d):
    """v_mfma_f64_16x16x4_f64 on fragments: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k,
    D[i][j] in lane j + 16 (i % 4), register i // 4  (d: [64][4])"""
    lane = np.arange(64)
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[lane % 16, lane // 16] = a
    B[lane // 16, lane % 16] = b
    P = A.dot(B)
    out = d.copy()
    for i in range(16):
        out[np.arange(16) + 16 * (i % 4), i // 4] += P[i]
    return out


def _mfma_f64_4x4x4_4b(a, b, d):
    """v_mfma_f64_4x4x4_4b_f64: four blocks; A[i][k] of block q in lane i + 4 q + 16 k, B[k][j] in lane j + 4 q + 16 k,
    D[i][j] in lane j + 4 q + 16 i  (d: [64])"""
    out = d.copy()
    for q in range(4):
        A = np.array([[a[i + 4 * q + 16 * k] for k in range(4)] for i in range(4)])
        B = np.array([[b[j + 4 * q + 16 * k] for j in range(4)] for k in range(4)])
        P = A.dot(B)
        for i in range(4):
            for j in range(4):
                out[j + 4 * q + 16 * i] += P[i, j]
    return out


def _swap16(a, b):
    """v_permlane16_swap a, b: a.row1 <-> b.row0, a.row3 <-> b.row2 (rows of 16 lanes)"""
    a, b = a.copy(), b.copy()
    for hi, lo in ((1, 0), (3, 2)):
        t = a[16 * hi:16 * hi + 16].copy()
        a[16 * hi:16 * hi + 16] = b[16 * lo:16 * lo + 16]
        b[16 * lo:16 * lo + 16] = t
    return a, b


def _swap32(a, b):
    """v_permlane32_swap a, b: a.rows 2, 3 <-> b.rows 0, 1"""
    a, b = a.copy(), b.copy()
    t = a[32:].copy()
    a[32:] = b[:32]
    b[:32] = t
    return a, b


@pytest.fixture(scope="module")
def packed(native):
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = R.synthetic_model(7002, no=64)
    model = ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)
    return om, ocr._pack_lstm_f64(model)


def test_four_line_fragments_and_row_swaps_give_every_cell_its_four_gates(packed):
    """lstm_seq4_f64_kernel: tile = block (unit-in-tile) x row (gate); B = h of the four lines in every block; three
    accumulators of a wave through the (accumulator, row) transpose -> row r of the wave holds gates 0..3 of tile r."""
    om, (wh, wx, peep, wh4) = packed
    rng = np.random.default_rng(1)
    h = rng.uniform(-1, 1, size=(100, 4))                                    # h_{t-1}[unit][line]
    lane = np.arange(64)
    for d, w in enumerate((om.fwd, om.rev)):
        Wg = np.stack([w[n][:, 49:] for n in ("WGI", "WGF", "WGO", "WCI")])  # [gate][unit][100]
        want = np.einsum("gur,rl->gul", Wg, h)                              # [gate][unit][line]
        for tile0 in (0, 9, 21):
            acc = []
            for tile in (tile0, tile0 + 1, tile0 + 2):
                dreg = np.zeros(64)
                for kk in range(25):
                    bfrag = h[4 * kk + lane // 16, lane % 4]                 # k = lane // 16, line = lane % 4, any block
                    dreg = _mfma_f64_4x4x4_4b(wh4[d, tile, kk], bfrag, dreg)
                # before the transpose: lane (line j, unit-in-tile q, gate i) = j + 4 q + 16 i
                for L in (0, 5, 23, 42, 63):
                    assert abs(dreg[L] - want[L // 16, 4 * tile + (L // 4) % 4, L % 4]) < 1e-12
                acc.append(dreg)
            g0, g1, g2, g3 = acc[0], acc[1], acc[2], np.full(64, np.nan)
            g0, g1 = _swap16(g0, g1)
            g2, g3 = _swap16(g2, g3)
            g0, g2 = _swap32(g0, g2)
            g1, g3 = _swap32(g1, g3)
            for L in range(48):                                              # rows 0..2: the cell (tile slot L // 16, unit, line)
                tile, unit, line = tile0 + L // 16, 4 * (tile0 + L // 16) + (L // 4) % 4, L % 4
                got = np.array([g0[L], g1[L], g2[L], g3[L]])
                assert np.abs(got - want[:, unit, line]).max() < 1e-12, (tile, L)


def test_sixteen_line_fragments(packed):
    """lstm_seq_f64_kernel: weights as the A operand, row i = 4 gate + unit-in-tile; lane (line j, q) ends up with the
    four gates of (line j, unit 4 tile + q) in its four accumulator registers.  lstm_xproj_f64_kernel: rows as A,
    weights as B, column i of a tile = position i of the tile in a row of Gx."""
    om, (wh, wx, peep, wh4) = packed
    rng = np.random.default_rng(2)
    h = rng.uniform(-1, 1, size=(100, 16))
    x = rng.uniform(0, 1, size=(16, 48))
    lane = np.arange(64)
    for d, w in enumerate((om.fwd, om.rev)):
        Wg = np.stack([w[n] for n in ("WGI", "WGF", "WGO", "WCI")])          # [gate][unit][149]
        want_h = np.einsum("gur,rl->gul", Wg[:, :, 49:], h)
        for wave, slot in ((0, 0), (2, 5), (1, 6), (3, 6)):
            tile = 6 * wave + slot if slot < 6 else 24
            dreg = np.zeros((64, 4))
            for kk in range(25):
                dreg = _mfma_f64_16x16x4(wh[d, wave, slot, kk], h[4 * kk + lane // 16, lane % 16], dreg)
            for L in (0, 17, 40, 63):
                for g in range(4):
                    assert abs(dreg[L, g] - want_h[g, 4 * tile + L // 16, L % 16]) < 1e-12
        # the projection: Gx[row][gx_index(unit, gate)] with gx_index = 16 (unit // 4) + 8 (gate // 2) + 2 (unit % 4) + gate % 2
        xa = np.concatenate([np.ones((16, 1)), x], axis=1)                        # [row][49]
        want_x = np.einsum("guk,rk->rug", Wg[:, :, :49], xa)                        # [row][unit][gate]
        for tile in (0, 13, 24):
            dreg = np.repeat(wx[d, tile, 12][:, None], 4, axis=1)                  # the accumulators start from the bias
            assert np.array_equal(wx[d, tile, 12][:16], wx[d, tile, 12][48:])      # (the same in every lane of a column)
            for kk in range(12):
                dreg = _mfma_f64_16x16x4(x[lane % 16, 4 * kk + lane // 16], wx[d, tile, kk], dreg)
            for L in (0, 21, 63):
                for r in range(4):
                    row, col = 4 * r + L // 16, L % 16                          # D[i][j]: i = row of x, j = column of the tile
                    unit, gate = 4 * tile + (col % 8) // 2, 2 * (col // 8) + col % 2
                    assert abs(dreg[L, r] - want_x[row, unit, gate]) < 1e-12
    assert np.array_equal(peep[0, 0], om.fwd["WIP"]) and np.array_equal(peep[1, 2], om.rev["WOP"])
