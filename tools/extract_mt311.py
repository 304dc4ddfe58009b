#!/usr/bin/env python3
"""Regenerate misc/mt311.fa (stripped from the reference checkout, see
/root/reference/.MISSING_LARGE_BLOBS) from the string literal embedded in
src/mt311.c:8-284.  Output is a data file (a 16 619 bp sequence), 60-column
FASTA, header as SURVEY.md section 8(d) prescribes."""
import re, sys

def extract(c_path):
    txt = open(c_path).read()
    body = txt[txt.index("mt311_sequence[]"):]
    body = body[:body.index(";")]
    return "".join(re.findall(r'"([^"]*)"', body))

def main():
    seq = extract(sys.argv[1])
    with open(sys.argv[2], "w") as f:
        f.write(">mt311 consensus of 311 human mitochondria\n")
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + "\n")
    print(f"mt311: {len(seq)} bp", file=sys.stderr)

if __name__ == "__main__":
    main()
