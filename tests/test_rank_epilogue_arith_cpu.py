"""The arithmetic the counting epilogue of the scoring kernel relies on (kge_amd/csrc/score_pairs_bf16_v4.hip,
V3_RANK), restated in numpy float32 and checked against the tie arithmetic it must reproduce
(EntityRankingJob._get_ranks_and_num_ties, eval_entity_ranking.py:571-596 = count_one in csrc/common.hpp):

  for a finite true score t and a finite tolerance allowed = atol + |rtol * t| >= 0,
      x' = max(x, -inf)            (NaN -> -inf, as the reference does)
      e  = x' - t
      greater-and-not-close  <=>  sign bit of (allowed - e)
      close                  <=>  NOT sign bit of (allowed - |e|)
  and the 16 dense result bits of a half tile spread to the tile's column layout (bit 8 q + b for element 4 q + b).

No GPU: host arithmetic only (IEEE float32 in numpy = what v_max_f32 / v_sub_f32 compute)."""
import numpy as np

F = np.float32


def count_one(x, t, atol, rtol):
    """(greater-and-not-close, close) of rank.hip / the reference: NaN -> -inf; torch.isclose semantics in float32."""
    x = np.where(np.isnan(x), F(-np.inf), x).astype(F)
    with np.errstate(invalid="ignore", over="ignore"):
        err = np.abs(x - t).astype(F)
        allowed = F(atol + np.abs(F(rtol * t)))
        close = (x == t) | (np.isfinite(err) & (err <= allowed))
    return (x > t) & ~close, close


def sign_bit_form(x, t, atol, rtol):
    with np.errstate(invalid="ignore", over="ignore"):
        al = F(atol + np.abs(F(rtol * t)))
        xp = np.fmax(x, F(-np.inf)).astype(F)          # v_max_f32: the non-NaN operand
        e = (xp - t).astype(F)
        greater = np.signbit((al - e).astype(F))
        close = ~np.signbit((al - np.abs(e)).astype(F))
    return greater, close


def _candidates(rng, t, al):
    eps = np.finfo(F).eps
    edge = [t + al, t - al, np.nextafter(F(t + al), F(np.inf)), np.nextafter(F(t + al), F(-np.inf)),
            np.nextafter(F(t - al), F(np.inf)), np.nextafter(F(t - al), F(-np.inf)),
            t, np.nextafter(F(t), F(np.inf)), np.nextafter(F(t), F(-np.inf)), t * (1 + eps), t * (1 - eps)]
    special = [np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0, np.finfo(F).max, -np.finfo(F).max, np.finfo(F).tiny,
               1e-45, -1e-45]
    rand = (rng.standard_normal(400) * max(abs(float(t)), 1.0) * rng.choice([1e-6, 1e-3, 1.0, 1e3], 400)).tolist()
    near = (t + rng.standard_normal(400) * max(float(al), 1e-30) * 2).tolist()
    return np.array(edge + special + rand + near, dtype=F)


def test_sign_bit_comparisons_equal_the_reference_tie_arithmetic():
    rng = np.random.default_rng(0)
    cases = 0
    for atol, rtol in ((1e-5, 1e-4), (0.0, 0.0), (0.05, 0.0), (0.0, 1e-2), (1e-30, 0.0), (3.0, 0.5)):
        for t in [0.0, -0.0, 1.0, -1.0, 3.5e-4, -417.25, 1e30, -1e30, 1e-38, 65504.0] + (rng.standard_normal(40) * 50).tolist():
            t = F(t)
            al = F(atol + np.abs(F(rtol * t)))
            assert np.isfinite(t) and np.isfinite(al) and al >= 0       # the kernel's fast-path condition
            x = _candidates(rng, t, al)
            g0, c0 = count_one(x, t, F(atol), F(rtol))
            g1, c1 = sign_bit_form(x, t, F(atol), F(rtol))
            bad = np.nonzero((g0 != g1) | (c0 != c1))[0]
            assert bad.size == 0, (float(t), atol, rtol, x[bad][:5], g0[bad][:5], g1[bad][:5], c0[bad][:5], c1[bad][:5])
            assert not (g1 & c1).any()
            cases += x.size
    assert cases > 40000


def test_dense_bits_spread_to_the_tile_layout():
    def spread(d):
        return (d & 0xF) | ((d & 0xF0) << 4) | ((d & 0xF00) << 8) | ((d & 0xF000) << 12)
    for r in range(16):
        assert spread(1 << r) == 1 << (8 * (r >> 2) + (r & 3))
    assert spread(0xFFFF) == 0x0F0F0F0F
    # with the 4 fh shift the two lanes of a row cover all 32 columns of a half tile exactly once
    assert (spread(0xFFFF) | (spread(0xFFFF) << 4)) == 0xFFFFFFFF and (spread(0xFFFF) & (spread(0xFFFF) << 4)) == 0
    # shifting in through bit 0 with the elements taken 15 .. 0 leaves element r at bit r
    d = 0
    for r in range(15, -1, -1):
        d = ((d << 1) | (1 if r in (0, 5, 15) else 0)) & 0xFFFFFFFF
    assert d & 0xFFFF == (1 << 0) | (1 << 5) | (1 << 15)
