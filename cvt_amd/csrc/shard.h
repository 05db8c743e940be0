// shard.h -- internal interface between the communicator (shard.hip) and the search entry points (api.hip).
#pragma once
#include "host_util.h"

namespace cvtmi {

int comm_validate(cvtmi_comm_t c);   // CVTMI_EINVAL for a null / stale / destroyed handle
int comm_world(cvtmi_comm_t c);
int comm_rank(cvtmi_comm_t c);
int comm_device(cvtmi_comm_t c);
bool comm_has_transport(cvtmi_comm_t c);
HandleSync *comm_sync(cvtmi_comm_t c);
size_t comm_slot_bytes(int64_t nq, int k);
// this rank's slot of the gather buffer for an [nq][k] result: the local search writes its lists there
int comm_local_slot(cvtmi_comm_t c, int64_t nq, int k, float **dist, int64_t **ids);
// ONE all-gather of the slots (RCCL, or the caller's transport) + merge of the per-rank lists into dist / ids
// status: the return code of this rank's local search (it travels with the slot: a failure anywhere fails every rank)
// the failure a deferred status check (comm_check_status = 2) has recorded since the last report: CVTMI_ECOMM once, else CVTMI_OK
int comm_take_deferred(cvtmi_comm_t c);
int comm_exchange_merge(cvtmi_comm_t c, int64_t nq, int k, int status, float *dist, int64_t *ids, hipStream_t st);
// the communicators of one process (cvtmi_comm_create_all): grouped all-gathers, merge on comms[0]'s device
int comm_exchange_merge_all(cvtmi_comm_t *comms, int ndev, int64_t nq, int k, const int *status, float *dist, int64_t *ids);
void comm_set_force_rccl(int v);
void comm_set_check_status(int v);

}  // namespace cvtmi
