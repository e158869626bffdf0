// Layout of the shared-prefix (cascade) decode plan, an int32 array produced on the device by
// sgl_amd_cascade_plan and consumed by the cascade attention kernels (no host involvement, so
// the whole thing is recorded into the decode hipGraph).
//
//   header[8]            [0] n_items  [1] n_groups  [2] n_member_rows
//   req_shared[B]        kv tokens of request b that are covered by its group's shared part (0 = none)
//   member_rows[B]       batch indices sorted by group (the "query tokens" of a group)
//   group_qo[B + 1]      member_rows range of each group
//   group_pool_row[B]    req_to_token row of the group's leader
//   group_kvlen[B]       shared kv length of the group (multiple of the kv tile)
//   items[8 * max_items] self-contained records: (group, kv chunk, row tile, members, first member_rows
//                        index, kv tokens, req_to_token row of the leader, 0); members == 0: unused entry
//   batch_order[B]       permutation of the batch: grouped requests (group by group) first, then the rest
//   compare[2 B]         scratch between the plan's two launches: leader of request b, its common prefix with the leader
#pragma once
#include <stdint.h>

namespace sgl_amd {

struct CascadePlanView {
  int32_t* header;
  int32_t* req_shared;
  int32_t* member_rows;
  int32_t* group_qo;
  int32_t* group_pool_row;
  int32_t* group_kvlen;
  int32_t* items;
  int32_t* batch_order;
  int32_t* compare;
};

__host__ __device__ inline int64_t cascade_plan_ints(int64_t batch, int64_t max_items) {
  return 8 + batch + batch + (batch + 1) + batch + batch + 8 * max_items + batch + 2 * batch;
}

__host__ __device__ inline CascadePlanView cascade_plan_view(const int32_t* plan, int64_t batch, int64_t max_items) {
  CascadePlanView v;
  int32_t* p = const_cast<int32_t*>(plan);
  v.header = p; p += 8;
  v.req_shared = p; p += batch;
  v.member_rows = p; p += batch;
  v.group_qo = p; p += batch + 1;
  v.group_pool_row = p; p += batch;
  v.group_kvlen = p; p += batch;
  v.items = p; p += 8 * max_items;
  v.batch_order = p; p += batch;
  v.compare = p;
  return v;
}

}  // namespace sgl_amd
