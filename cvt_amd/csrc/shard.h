// shard.h -- internal interface between the communicator (shard.hip) and the search entry points (api.hip).
#pragma once
#include "host_util.h"

namespace cvtmi {

int comm_world(cvtmi_comm_t c);
int comm_device(cvtmi_comm_t c);
bool comm_has_transport(cvtmi_comm_t c);
HandleSync *comm_sync(cvtmi_comm_t c);
size_t comm_slot_bytes(int64_t nq, int k);
// this rank's slot of the gather buffer for an [nq][k] result: the local search writes its lists there
int comm_local_slot(cvtmi_comm_t c, int64_t nq, int k, float **dist, int64_t **ids);
// ONE all-gather of the slots (RCCL, or the caller's transport) + merge of the per-rank lists into dist / ids
int comm_exchange_merge(cvtmi_comm_t c, int64_t nq, int k, float *dist, int64_t *ids, hipStream_t st);
void comm_set_force_rccl(int v);

}  // namespace cvtmi
