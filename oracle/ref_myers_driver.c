/* ref_myers_driver.c -- TEST INFRASTRUCTURE (ours, not reference code).
 * Line driver around the reference's myers_diff (src/myers_align.h:35),
 * linked against /root/reference/src/myers_align.c by oracle/Makefile.ref.
 * stdin:  <mode 0|1|2> <maxd> <seq_a> <seq_b>
 * stdout: <d as unsigned> <bt_a>            (bt_b is not NUL-terminated by the
 *          reference, src/myers_align.c:44-45, so only bt_a is printed) */
#include "myers_align.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ML (1 << 16)
static char a[ML], b[ML], bta[4 * ML], btb[4 * ML];
int main(void) {
  int mode, maxd;
  while (scanf("%d %d %65535s %65535s", &mode, &maxd, a, b) == 4) {
    memset(bta, 0, sizeof bta); memset(btb, 0, sizeof btb);
    unsigned d = myers_diff(a, (enum myers_align_mode)mode, b, maxd, bta, btb);
    printf("%u %s\n", d, d == (unsigned)-1 ? "-" : bta);
  }
  return 0;
}
