// mia_comm.h -- what mia_hip.hip needs from mia_comm.hip (the transports of a sharded iteration)
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/mia_hip.h"

// ncclCommInitRank on the current device; fills `out` with the RCCL table.  Every rank calls it at the same time.
int mia_comm_rccl_table(const void* id128, int32_t n_ranks, int32_t rank, mia_hip_collectives* out, std::string* err);
