// ingest_dump.cpp -- TEST HARNESS around host/ingest.h (the product's FASTA/FASTQ reader): prints the records it reads in
// the format of oracle/ref_read_driver.c, so that tests/test_ingest_cpu.py can compare them with what the reference's own
// reader made of the same file.  usage: ingest_dump <file> <threads> [bytes per thread at least]
#include <stdio.h>
#include <stdlib.h>

#include <chrono>

#include "../../mapping-iterative-assembler_amd/host/ingest.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::vector<ingest::Read> reads;
  std::string log;
  bool fastq = false;
  if (argc > 3) ingest::min_stretch_bytes() = (size_t)atoll(argv[3]);
  if (getenv("INGEST_SLOW")) ingest::use_fast_path() = false;
  const auto t0 = std::chrono::steady_clock::now();
  if (!ingest::read_all(argv[1], atoi(argv[2]), &reads, &fastq, &log)) return 1;
  if (getenv("INGEST_TIME")) {      // throughput only: no dump
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"reads\": %zu, \"seconds\": %.6f, \"reads_per_s\": %.0f, \"threads\": %d}\n", reads.size(), s, reads.size() / s, atoi(argv[2]));
    return 0;
  }
  for (const auto& r : reads) printf("%s\x1f%s\x1f%s\n", r.id.c_str(), r.desc.c_str(), r.seq.c_str());
  fputs(log.c_str(), stderr);
  return 0;
}
