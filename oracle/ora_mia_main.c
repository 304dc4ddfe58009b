/* ora_mia_main.c -- TEST INFRASTRUCTURE: command-line front end of the oracle
 * restatement, flag-compatible with the subset of `mia` (src/mia_main.c:477)
 * that is in scope: -r -f -m -s -c -i -n -p -H -S -N -k -M -F.
 * Used only to diff whole .maln outputs against oracle/_ref/mia. */
#include "mia_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int main(int argc, char **argv) {
  ora_opts o;
  ora_pssm anc;
  ora_state *st;
  const char *ref = NULL, *frags = NULL, *root = "assembly.maln.iter";
  int ch, iters;
  ora_opts_default(&o);
  ora_pssm_flat(&anc);
  while ((ch = getopt(argc, argv, "s:r:f:m:p:H:S:N:k:a:FcinMT")) != -1) {
    switch (ch) {
      case 'c': o.circular = 1; break;
      case 'n': o.iterate = 0; break;
      case 'i': o.iterate = 1; break;
      case 'p': o.cons_code = atoi(optarg); break;
      case 'H': o.hard_cut = atoi(optarg); break;
      case 'M': o.soft_mask = 1; break;
      case 's': if (!ora_pssm_read(optarg, &anc)) { fprintf(stderr, "cannot read matrix %s\n", optarg); return 10; } break;
      case 'r': ref = optarg; break;
      case 'f': frags = optarg; break;
      case 'm': root = optarg; break;
      case 'k': o.kmer_len = atoi(optarg); break;
      case 'S': o.slope = atof(optarg); o.score_cut_set = 1; break;
      case 'N': o.intercept = atof(optarg); o.score_cut_set = 1; break;
      case 'F': o.final_only = 1; break;
      case 'T': o.do_trim = 1; break;
      case 'a':   /* src/mia_main.c:558-578 */
        if (strlen(optarg) > 127) strcpy(o.adapter, "CTGAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG");
        else if (strlen(optarg) > 1) strcpy(o.adapter, optarg);
        else if (!(optarg[0] == 'n' || optarg[0] == 'N')) strcpy(o.adapter, "CTGAGACACGCAACAGGGGATAGGCAAGGCACACAGGGGATAGG");
        break;
      default: return 2;
    }
  }
  if (!ref || !frags) { fprintf(stderr, "need -r and -f\n"); return 2; }
  st = ora_new(&o, &anc);
  if (!ora_load_ref_fasta(st, ref)) { fprintf(stderr, "Problem reading reference sequence file %s\n", ref); return 1; }
  ora_prepare_ref(st);
  iters = ora_run(st, frags, root);
  fprintf(stderr, "oracle: %d iteration(s)\n", iters);
  ora_free(st);
  return iters < 0;
}
