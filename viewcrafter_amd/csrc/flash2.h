// Interface between attention.hip (C ABI entry, dispatch) and attention_v2.hip (software-pipelined flash kernel).
#pragma once
#include "vcx_common.h"

struct Flash2Args {
    const half_t* q;
    const half_t* k;
    const half_t* vt;
    half_t* o;
    int heads, nq, nk, kv_rows, kv_div;
    int64_t ldq, ldk, ldvt, ldo;
    int nqb, nprob;   // 256-row query blocks per problem, problems (group x head): 1-D grid of nqb * roundup(nprob, 8)
};

// nk % 64 == 0, base-2 logits (VCX_ATTN_LOG2_LOGITS), no accumulate: checked by the caller.
int vcx_flash2_launch(const Flash2Args& a, hipStream_t s);
