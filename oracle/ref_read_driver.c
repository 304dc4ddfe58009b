/* ref_read_driver.c -- TEST INFRASTRUCTURE (ours, not reference code).
 *
 * The REAL reference's fragment reader -- find_input_type + read_next_seq (read_fasta / read_fastq,
 * /root/reference/src/io.c:11-281), the loop of /root/reference/src/mia_main.c:742-759 -- over one file, linked from the
 * reference's own objects by oracle/Makefile.ref.  It prints every record the loop would hand to the aligner:
 *     <id> US <desc> US <seq> LF          (US = 0x1f)
 * so that tools/make_goldens.py ingest can commit what the reference reads out of deliberately awkward inputs
 * (tests/golden/ingest/) and tests/test_ingest_cpu.py can hold the product's multi-threaded reader against it.
 * usage: ref_read_driver <reads.fa|fq>     (the reference's own messages go to stderr)
 */
#include "mia.h"

int main(int argc, char **argv) {
  FILE *f;
  FragSeqP fs;
  int code;
  if (argc < 2) { fprintf(stderr, "usage: %s reads\n", argv[0]); return 2; }
  f = fopen(argv[1], "r");
  if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fs = (FragSeqP)calloc(1, sizeof(FragSeq));
  code = find_input_type(f);
  while (read_next_seq(f, fs, code)) printf("%s\x1f%s\x1f%s\n", fs->id, fs->desc, fs->seq);
  fclose(f);
  return 0;
}
