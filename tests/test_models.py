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
