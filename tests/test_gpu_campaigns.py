"""Fixed-seed slices of the differential campaigns (tools/band_campaign.py, tally_campaign.py, pass1_campaign.py), run by
the driver with the rest of `-m gpu`: every shortcut of the product path against the kernels that evaluate the whole
window / the whole strand / every base, on random and adversarial configurations: five million reads for the band pipeline, two million each
for the tally and pass 1 (round 4: the time went to configs[4] at its full size, tests/test_gpu_config4_full.py).

* band pipeline (plan, values DP, trace DP: csrc/bandx_body.h) against the full-window DP kernels
  (MIA_HIP_NO_DIAG_FILTER=1): flat, ancient.submat.txt and ancient.submat.solexa.pe.txt, references of plain bases and
  references with 3-20 % ambiguity codes (mt311's kind); score, end points and script of every read
  (reference recurrence: /root/reference/src/mia.c:740-981, traceback :612-637,1440-1497);
* tally: the one-read-per-lane / bit-sliced paths against per-base score adds (flat), the LDS-window tally against the
  plain global-atomic tally (ancient matrix): all tally words, ref->gaps, consensus, insert tallies
  (/root/reference/src/map_align.c:229-276, src/mia.c:515-603);
* pass 1 without a k-mer mask: diagonal filter + anchored windows against the whole-strand DP, plain and N-rich
  references, flat and position-specific matrices (/root/reference/src/mia.c:1500-1665).
The long campaigns of profiles/r0*/README.md are the same functions with more rounds."""
import pytest

pytestmark = pytest.mark.gpu


def test_band_pipeline_against_full_window_kernels():
    import band_campaign
    total = placed = 0
    for matrix, seed, nrich in (("flat", 410000, False), ("flat", 411000, True), ("ancient", 412000, False), ("ancient", 413000, True),
                                ("solexa", 414000, False), ("solexa", 415000, True)):
        r, p = band_campaign.run(9, seed, "MIA_HIP_NO_DIAG_FILTER", matrix, nrich, n=100_000, quiet=True)
        total += r
        placed += p
    assert total >= 5_000_000 and placed > 0.2 * total, (total, placed)      # (a third of the configurations are adversarial)


def test_tally_paths_against_plain_tallies():
    import tally_campaign
    total = tally_campaign.run(6, 420000, n=200_000, matrix="flat", switch="MIA_HIP_NO_LINEAR_TALLY", quiet=True)
    total += tally_campaign.run(5, 421000, n=200_000, matrix="ancient", switch="MIA_HIP_NO_BINNED_TALLY", quiet=True)
    assert total >= 2_000_000


def test_pass1_shortcuts_against_whole_strand_dp():
    import pass1_campaign
    total = decided = 0
    for seed, nrich in ((430000, False), (431000, True)):
        r, f, a = pass1_campaign.run(10, seed, nrich, n=100_000, quiet=True)
        total += r
        decided += f + a
    assert total >= 2_000_000 and decided > 0.3 * total, (total, decided)
    # position-specific matrices: the anchored windows in losses (no diagonal filter), damaged reads
    total = decided = 0
    for seed, nrich, matrix in ((432000, False, "ancient"), (433000, True, "ancient"), (434000, False, "solexa")):
        r, f, a = pass1_campaign.run(5, seed, nrich, n=100_000, quiet=True, matrix=matrix)
        total += r
        decided += a
    assert total >= 1_500_000 and decided > 0.15 * total, (total, decided)
