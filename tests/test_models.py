"""Design-validation models (tests/models/): a numpy re-enactment of one wavefront of the AAC long-block kernel with the
kernel's exact lane <-> data mapping and LDS transposes, checked bit-for-bit against the oracle, and the LDS bank model
that certifies the three transposes as conflict-free."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent / "models"))


def test_aac_wave_model_matches_oracle_and_is_conflict_free():
    import aac_wave_model
    ok, totals = aac_wave_model.main()
    assert ok
    assert totals, "the LDS access log is empty"
    for (name, instr), (cycles, ideal) in totals.items():
        assert cycles == ideal, "%s %s: %d LDS cycles, ideal %d" % (name, instr, cycles, ideal)


def short_transform_lds_cycles(skew):
    """LDS-array cycles of the 32 strided per-window accesses of imdct_short_wave (imdct_wave.h): lane = (window w,
    column c); pre-twiddle reads x_w[2i], x_w[127 - 2i] (i = c + 8 s), post-twiddle writes of the half-stored output."""
    import lds_sim
    total = 0
    for s in range(8):
        for rev in (0, 1):
            addrs = []
            for lane in range(64):
                w, c = lane >> 3, lane & 7
                i = c + 8 * s
                addrs.append(4 * (128 * w + skew[w] + (127 - 2 * i if rev else 2 * i)))
            total += lds_sim.cycles("read_b32", addrs)
    for j in range(8):
        for which in (0, 1):
            addrs = []
            for lane in range(64):
                w, c = lane >> 3, lane & 7
                i = 8 * j + c
                if j < 4:
                    off = 2 * i if which == 0 else 64 + 63 - 2 * i
                else:
                    off = 63 - 2 * (i - 32) if which == 0 else 64 + 2 * (i - 32)
                addrs.append(4 * (128 * w + skew[w] + off))
            total += lds_sim.cycles("write_b32", addrs)
    return total


def test_short_window_rows_are_skewed_off_each_others_banks():
    """The skew of imdct_wave.h's short_row(): unskewed rows are 4-way conflicted (256 cycles for 32 instructions whose
    ideal is 64; measured on the MI355X as 65 % of imdct128_wave_kernel's LDS cycles), the chosen multiples of four halve
    that, and no other 16-byte-aligned, non-overlapping choice does better."""
    chosen = [0, 16, 24, 40, 40, 56, 80, 96]
    assert all(b >= a for a, b in zip(chosen, chosen[1:])) and all(v % 4 == 0 for v in chosen)  # rows do not overlap, stay aligned
    assert short_transform_lds_cycles([0] * 8) == 256
    assert short_transform_lds_cycles(chosen) == 128
    # only the skew mod 32 matters to the bank functions of these b32 accesses: a random search over aligned residues
    import random
    rng = random.Random(1)
    for _ in range(3000):
        cand = [0] + [rng.randrange(0, 32, 4) for _ in range(7)]
        assert short_transform_lds_cycles(cand) >= 128
